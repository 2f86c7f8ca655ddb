// nvx_api.cpp -- host runtime behind the C ABI of include/navtex_amd.h:
// handle (device buffers, carried state, result ring), block API, pinned
// staging for host input, reference-compatible push shim, SDRplay-shaped
// callback adapter, WAV harness, device launcher of the synthetic source.
//
// There is no CPU implementation of the signal path in this library: every
// entry point that needs the GPU returns NVX_ERR_NODEV / NVX_ERR_HIP without it
// (and the void reference-shaped entry points print and abort()).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "navtex_amd.h"
#include "nvx_internal.h"
#include "nvx_kernels.h"
#include "nvx_fsm.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
extern "C" void nvx_set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *nvx_last_error(void) { return g_err; }
extern "C" const char *nvx_version(void) { return "navtex_amd 0.1 (gfx950)"; }

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            nvx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? NVX_ERR_NODEV : NVX_ERR_HIP; \
        }                                                                                  \
    } while (0)

static int select_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        nvx_set_error("no HIP device available (%s); libnavtex_amd has no CPU path",
                      e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return NVX_ERR_NODEV;
    }
    if (device < 0 || device >= n) { nvx_set_error("device %d out of range (0..%d)", device, n - 1); return NVX_ERR_ARG; }
    HIP_TRY(hipSetDevice(device));
    return NVX_OK;
}

// ------------------------------------------------------------------ handle
static const int RESULT_SLOTS = 4;

struct Message { std::string bbbb, text; int freq; };

struct Slot {                          // one (stream, chain)
    bool active = false;
    int label = 0;
    std::string bits;                  // the most recent decoded bits (at most 2*NVX_BIT_HISTORY of them)
    size_t base = 0;                   // absolute index (since create/reset) of bits[0]
    size_t polled = 0;                 // nvx_poll_bits cursor, absolute
    nvx_sitor *sitor = nullptr;
    std::vector<Message> outbox;       // messages completed during a (possibly threaded) collect
};

struct Result {                        // one in-flight launch's bit output
    uint8_t *d_bits = nullptr; int *d_nbits = nullptr;
    uint8_t *h_bits = nullptr; int *h_nbits = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t ev[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };   // begin/end of cascade, demod front, demod FSM
    bool timed = false;
    bool pending = false;
};

struct nvx_handle {
    nvx_config cfg{};
    int n_streams = 0, n_slots = 0, nch = 1;   // n_streams: 252 kS/s-path streams (8 per input in wideband mode)
    int n_in = 0;                      // input streams the caller addresses (= n_streams unless wideband)
    size_t bit_history = NVX_BIT_HISTORY;
    bool cascade_raw = false;          // the cascade kernel's RAW switch (never set in wideband mode)
    size_t frame_in = 0;               // complex input samples per frame at the input rate
    // wideband mode: channeliser on stream3 into sub[b], overlapping the cascade of the previous launch
    hipStream_t stream3 = nullptr;
    uint32_t *d_sub[2] = { nullptr, nullptr };
    uint32_t *d_whist[2] = { nullptr, nullptr };
    hipEvent_t chan_done[2] = { nullptr, nullptr }, sub_free[2] = { nullptr, nullptr }, in_ready[2] = { nullptr, nullptr };
    bool sub_busy[2] = { false, false };
    uint64_t wide_launches = 0;
    int y3_cap = 0, bits_cap = 0;
    hipStream_t stream = nullptr;      // FIR cascade (or the caller's stream) and H2D staging
    hipStream_t stream2 = nullptr;     // demodulator FSM + D2H of the bits: overlaps the next cascade launch
    hipEvent_t casc_done[2] = { nullptr, nullptr };   // y3[b] written
    hipEvent_t demod_done[2] = { nullptr, nullptr };  // y3[b] consumed
    hipEvent_t fsm_done = nullptr; bool fsm_pending = false;   // word buffer consumed
    bool demod_pending[2] = { false, false };
    // device
    uint8_t *d_masks = nullptr, *d_active = nullptr, *d_cstate = nullptr;
    double2 *d_y3[2] = { nullptr, nullptr };   // double buffer between the two streams
    double *d_dd = nullptr, *d_dphi = nullptr; int *d_di = nullptr;
    uint32_t *d_fsm_tab = nullptr;     // bit-period transition table of the demodulator FSM (nvx_fsm.h)
    unsigned short *d_words = nullptr;
    int *d_ctrl = nullptr;             // cascade work queue: counter, status, done[n_streams]
    int *h_status = nullptr;           // pinned copy of the status word of the last launch
    unsigned long long g0 = 0;         // 900 S/s samples per chain since reset
    Result res[RESULT_SLOTS];
    uint64_t launched = 0, collected = 0;
    int last_n3 = 0;
    // timing
    bool timing = false;
    float ms[2] = { 0.f, 0.f };          // last collected launch
    double ms_sum[2] = { 0.0, 0.0 };     // over all collected launches since the last stats reset
    uint64_t ms_count = 0;
    // host
    std::vector<uint8_t> masks;
    std::vector<Slot> slots;
    std::vector<struct SinkCtx *> sinks;   // user pointers handed to the per-slot character layers
    std::mutex mu;
    // push mode staging: two pinned sets [n_streams][stage_cap] of packed IQ words
    uint32_t *h_stage[2] = { nullptr, nullptr };
    hipEvent_t stage_free[2] = { nullptr, nullptr };
    bool stage_busy[2] = { false, false };
    int cur = 0;
    size_t stage_cap = 0;
    std::vector<size_t> fill;
    uint32_t *d_in = nullptr;
};

struct SinkCtx { nvx_handle *h; int stream; int slot; };

// The character layers of different chains run on worker threads; their messages are
// parked per slot and handed to the user's sink afterwards, in slot order, by the
// collecting thread (the reference calls add_message from its single DSP thread).
static void sitor_sink(void *user, const char *bbbb, const char *message, int freq);

static void deliver_outbox(nvx_handle *h, int stream, Slot &s);

// Default sink when nothing else in the program defines add_message (the reference's
// message_store.c does): the database named by NAVTEX_AMD_DB, else stdout.
extern "C" __attribute__((weak, visibility("default"))) int add_message(char *bbbb, char *message, int freq)
{
    static std::once_flag once;
    static nvx_store *store = nullptr;
    std::call_once(once, [] {
        const char *path = getenv("NAVTEX_AMD_DB");
        if (path && *path && nvx_store_open(path, 1, &store) != NVX_OK) {
            fprintf(stderr, "navtex_amd: NAVTEX_AMD_DB=%s: %s\n", path, nvx_last_error());
            abort();                             // a configured sink that cannot be opened must not lose messages quietly
        }
    });
    if (store) return nvx_store_add_message(store, bbbb, message, freq);
    printf("[navtex_amd] message freq=%d bbbb=%s\n%s", freq, bbbb, message);
    fflush(stdout);
    return 0;
}

static void sitor_sink_impl(nvx_handle *h, int slot, const char *bbbb, const char *message, int freq);

static void sitor_sink(void *user, const char *bbbb, const char *message, int freq)
{
    SinkCtx *c = (SinkCtx *)user;
    sitor_sink_impl(c->h, c->slot, bbbb, message, freq);
}

extern "C" void nvx_config_default(nvx_config *c)
{
    memset(c, 0, sizeof *c);
    c->device = 0; c->n_streams = 1; c->raw_rate = 0;
    c->chain_mask = NVX_CHAIN_518 | NVX_CHAIN_490;
    c->max_frames = 1; c->char_layer = 1; c->push_mode = 0;
}

// Bit-period transition table of the demodulator FSM, generated once from the per-sample rule (nvx_fsm.h).
static const uint32_t *fsm_table_host()
{
    static const std::vector<uint32_t> table = [] {
        std::vector<uint32_t> t(NVX_FSM_TABLE_ALLOC, 0u);
        for (int p1 = 0; p1 < 9; p1++)
            for (int so = 0; so < 10; so++)
                for (int a = 0; a < 9; a++)
                    for (int b = 0; b < 9; b++) t[NVX_FSM_KEY(p1, so, a, b)] = nvx_fsm_table_entry(p1, so, a, b);
        for (int prev1 = 0; prev1 < 10; prev1++)
            for (int rawc = 0; rawc < 10; rawc++) t[NVX_FSM_TIMING_BASE + prev1 * 10 + rawc] = nvx_fsm_timing_entry(prev1, rawc);
        return t;
    }();
    return table.data();
}

// Replays pseudo-random front-kernel words through the per-sample rule and through the table, and counts
// differences in the decided bits and in the carried registers.  No device needed (tests/test_host_layer.py).
extern "C" int nvx_fsm_selftest(uint32_t seed, int periods)
{
    const uint32_t *tab = fsm_table_host();
    uint32_t x = seed ? seed : 1u;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; };
    // per-sample registers (the reference's variables) and per-period registers
    int synced = 0, sync_off = 0, next_sync_off = 0, phase = -1, prev_a = -1;
    nvx_fsm_regs r = { 0, NVX_FSM_UNSYNCED, 0, -1 };
    const int lead = (int)(rnd() % 70u);                     // periods before the class sums are primed
    int raw = (int)(rnd() % 9u), bad = 0;
    for (int m = 0; m < periods; m++) {
        const uint32_t u = rnd();
        if (u % 7u == 0) raw = (int)((u >> 8) % 9u);           // timing jumps; otherwise it drifts or holds
        else if (u % 7u == 1) raw = (raw + 1) % 9;
        else if (u % 7u == 2) raw = (raw + 8) % 9;
        const unsigned w = ((u >> 16) & 0x1ffu) | ((unsigned)(m < lead ? 15 : raw) << 12);
        unsigned want = 0; int n_want = 0;
        for (int k = 0; k < 9; k++) {
            if (k == NVX_FSM_TIMING_SAMPLE) {
                int offset;
                const int have = nvx_fsm_timing((int)(w >> 12), &prev_a, &offset);
                sync_off = (have && !synced) ? offset : sync_off;      // decoder.C:62-70
                next_sync_off = have ? offset : next_sync_off;
                synced = have ? 1 : synced;
            }
            if (nvx_fsm_bit_step(k, synced, &phase, &sync_off, next_sync_off)) { want |= ((w >> k) & 1u) << n_want; n_want++; }
        }
        int n_got;
        const unsigned got = nvx_fsm_period(tab, w, &r, &n_got) & ((1u << n_got) - 1u);
        if (n_got != n_want || got != want) bad++;
        if (r.phase1 != phase + 1 || r.nso != next_sync_off || r.prev_offset != prev_a ||
            (r.so != NVX_FSM_UNSYNCED) != (synced != 0) || (synced && r.so != sync_off)) bad++;
    }
    return bad;
}

static void free_handle(nvx_handle *h)
{
    if (!h) return;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->stream2) hipStreamSynchronize(h->stream2);
    if (h->stream3) { hipStreamSynchronize(h->stream3); hipStreamDestroy(h->stream3); }
    for (int i = 0; i < 2; i++) {
        hipFree(h->d_sub[i]); hipFree(h->d_whist[i]);
        if (h->chan_done[i]) hipEventDestroy(h->chan_done[i]);
        if (h->sub_free[i]) hipEventDestroy(h->sub_free[i]);
        if (h->in_ready[i]) hipEventDestroy(h->in_ready[i]);
    }
    hipFree(h->d_masks); hipFree(h->d_active); hipFree(h->d_cstate); hipFree(h->d_y3[0]); hipFree(h->d_y3[1]);
    for (int i = 0; i < 2; i++) { if (h->casc_done[i]) hipEventDestroy(h->casc_done[i]); if (h->demod_done[i]) hipEventDestroy(h->demod_done[i]); }
    if (h->fsm_done) hipEventDestroy(h->fsm_done);
    hipFree(h->d_dd); hipFree(h->d_di); hipFree(h->d_fsm_tab); hipFree(h->d_dphi); hipFree(h->d_in); hipFree(h->d_words); hipFree(h->d_ctrl);
    if (h->h_status) hipHostFree(h->h_status);
    for (auto &r : h->res) {
        hipFree(r.d_bits); hipFree(r.d_nbits);
        if (r.h_bits) hipHostFree(r.h_bits);
        if (r.h_nbits) hipHostFree(r.h_nbits);
        if (r.done) hipEventDestroy(r.done);
        for (int i = 0; i < 6; i++) if (r.ev[i]) hipEventDestroy(r.ev[i]);
    }
    for (int i = 0; i < 2; i++) {
        if (h->h_stage[i]) hipHostFree(h->h_stage[i]);
        if (h->stage_free[i]) hipEventDestroy(h->stage_free[i]);
    }
    for (auto &s : h->slots) {
        if (s.sitor) nvx_sitor_free(s.sitor);
    }
    for (auto *c : h->sinks) delete c;
    if (h->stream) hipStreamDestroy(h->stream);
    if (h->stream2) hipStreamDestroy(h->stream2);
    delete h;
}

extern "C" void nvx_destroy(nvx_handle *h) { free_handle(h); }

extern "C" int nvx_create(const nvx_config *cfg, nvx_handle **out)
{
    if (!cfg || !out || cfg->n_streams < 1 || cfg->max_frames < 1) { nvx_set_error("nvx_create: bad config"); return NVX_ERR_ARG; }
    *out = nullptr;
    int rc = select_device(cfg->device);
    if (rc != NVX_OK) return rc;

    nvx_handle *h = new nvx_handle();
    h->cfg = *cfg;
    h->n_in = cfg->n_streams;
    h->n_streams = cfg->wideband ? NVX_WB_SUBBANDS * cfg->n_streams : cfg->n_streams;
    h->n_slots = 2 * h->n_streams;
    h->cascade_raw = cfg->raw_rate && !cfg->wideband;
    h->frame_in = (cfg->raw_rate || cfg->wideband) ? (size_t)NVX_FRAME_RAW : (size_t)NVX_FRAME_IN;
    h->y3_cap = cfg->max_frames * NVX_FRAME_Y3;
    if (cfg->bit_history < 0) { nvx_set_error("nvx_create: negative bit_history"); delete h; return NVX_ERR_ARG; }
    if (cfg->bit_history > 0) h->bit_history = (size_t)cfg->bit_history;
    // a bit needs >= 8 samples (the offset slews by at most 1 per bit); packed 8 bits per byte, whole words
    h->bits_cap = (((h->y3_cap / 8 + 8) + 31) / 32) * 4;
    h->masks.resize(h->n_streams);
    h->slots.resize(h->n_slots);
    bool any_two = false;
    for (int s = 0; s < h->n_streams; s++) {
        uint8_t m = cfg->chain_masks ? cfg->chain_masks[s] : (uint8_t)cfg->chain_mask;
        m &= 3;
        if (!m) { nvx_set_error("nvx_create: stream %d has an empty chain mask", s); free_handle(h); return NVX_ERR_ARG; }
        h->masks[s] = m;
        if (m == 3) any_two = true;
        for (int c = 0; c < 2; c++) {
            Slot &sl = h->slots[2 * s + c];
            sl.active = (m >> c) & 1;
            sl.label = cfg->labels ? cfg->labels[2 * s + c] : (c == 0 ? 518 : 490);
            if (sl.active && cfg->char_layer) {
                SinkCtx *ctx = new SinkCtx{ h, s, 2 * s + c };
                h->sinks.push_back(ctx);
                sl.sitor = nvx_sitor_new(sl.label, sitor_sink, ctx);
            }
        }
    }
    h->nch = any_two ? 2 : 1;

#define CR_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
        nvx_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); free_handle(h); \
        return e_ == hipErrorOutOfMemory ? NVX_ERR_NOMEM : NVX_ERR_HIP; } } while (0)

    CR_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CR_TRY(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        CR_TRY(hipEventCreateWithFlags(&h->casc_done[i], hipEventDisableTiming));
        CR_TRY(hipEventCreateWithFlags(&h->demod_done[i], hipEventDisableTiming));
    }
    CR_TRY(hipEventCreateWithFlags(&h->fsm_done, hipEventDisableTiming));
    std::vector<uint8_t> active(h->n_slots);
    for (int i = 0; i < h->n_slots; i++) active[i] = h->slots[i].active;
    CR_TRY(hipMalloc(&h->d_masks, h->n_streams));
    CR_TRY(hipMalloc(&h->d_active, h->n_slots));
    CR_TRY(hipMemcpy(h->d_masks, h->masks.data(), h->n_streams, hipMemcpyHostToDevice));
    CR_TRY(hipMemcpy(h->d_active, active.data(), h->n_slots, hipMemcpyHostToDevice));
    CR_TRY(hipMalloc(&h->d_cstate, (size_t)h->n_streams * NVX_CASCADE_STATE_BYTES));
    for (int i = 0; i < 2; i++) CR_TRY(hipMalloc(&h->d_y3[i], (size_t)h->n_slots * h->y3_cap * sizeof(double2)));
    CR_TRY(hipMalloc(&h->d_dd, (size_t)NVX_DEMOD_DOUBLES * h->n_slots * sizeof(double)));
    CR_TRY(hipMalloc(&h->d_di, (size_t)NVX_DEMOD_INTS * h->n_slots * sizeof(int)));
    CR_TRY(hipMalloc(&h->d_fsm_tab, NVX_FSM_TABLE_ALLOC * sizeof(uint32_t)));
    CR_TRY(hipMemcpy(h->d_fsm_tab, fsm_table_host(), NVX_FSM_TABLE_ALLOC * sizeof(uint32_t), hipMemcpyHostToDevice));
    CR_TRY(hipMalloc(&h->d_words, (size_t)(h->y3_cap / 9) * h->n_slots * sizeof(unsigned short)));
    CR_TRY(hipMalloc(&h->d_ctrl, (size_t)(NVX_CASCADE_CTRL_INTS + h->n_streams) * sizeof(int)));
    CR_TRY(hipHostMalloc((void **)&h->h_status, RESULT_SLOTS * sizeof(int), hipHostMallocDefault));
    memset(h->h_status, 0, RESULT_SLOTS * sizeof(int));
    for (auto &r : h->res) {
        CR_TRY(hipMalloc(&r.d_bits, (size_t)h->n_slots * h->bits_cap));
        CR_TRY(hipMalloc(&r.d_nbits, (size_t)h->n_slots * sizeof(int)));
        CR_TRY(hipHostMalloc((void **)&r.h_bits, (size_t)h->n_slots * h->bits_cap, hipHostMallocDefault));
        CR_TRY(hipHostMalloc((void **)&r.h_nbits, (size_t)h->n_slots * sizeof(int), hipHostMallocDefault));
        CR_TRY(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
        for (int i = 0; i < 6; i++) CR_TRY(hipEventCreate(&r.ev[i]));
    }
    if (cfg->wideband) {
        CR_TRY(hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            CR_TRY(hipMalloc(&h->d_sub[i], (size_t)h->n_streams * cfg->max_frames * NVX_FRAME_IN * 4));
            CR_TRY(hipMalloc(&h->d_whist[i], (size_t)h->n_in * 40 * 4));
            CR_TRY(hipEventCreateWithFlags(&h->chan_done[i], hipEventDisableTiming));
            CR_TRY(hipEventCreateWithFlags(&h->sub_free[i], hipEventDisableTiming));
            CR_TRY(hipEventCreateWithFlags(&h->in_ready[i], hipEventDisableTiming));
        }
    }
    if (cfg->push_mode) {
        h->stage_cap = (size_t)(cfg->max_frames + 1) * h->frame_in;
        for (int i = 0; i < 2; i++) {
            CR_TRY(hipHostMalloc((void **)&h->h_stage[i], (size_t)h->n_in * h->stage_cap * 4, hipHostMallocDefault));
            CR_TRY(hipEventCreateWithFlags(&h->stage_free[i], hipEventDisableTiming));
        }
        CR_TRY(hipMalloc(&h->d_in, (size_t)h->n_in * cfg->max_frames * h->frame_in * 4));
        h->fill.assign(h->n_in, 0);
    }
#undef CR_TRY
    rc = nvx_reset(h);
    if (rc != NVX_OK) { free_handle(h); return rc; }
    *out = h;
    return NVX_OK;
}

static int collect_locked(nvx_handle *h);

extern "C" int nvx_reset(nvx_handle *h)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream2));
    for (auto &r : h->res) r.pending = false;
    h->collected = h->launched;
    h->g0 = 0;
    h->demod_pending[0] = h->demod_pending[1] = false;
    h->fsm_pending = false;
    if (h->stream3) {
        HIP_TRY(hipStreamSynchronize(h->stream3));
        for (int i = 0; i < 2; i++) HIP_TRY(hipMemsetAsync(h->d_whist[i], 0, (size_t)h->n_in * 40 * 4, h->stream));
        h->sub_busy[0] = h->sub_busy[1] = false;
        h->wide_launches = 0;
    }
    HIP_TRY(hipMemsetAsync(h->d_cstate, 0, (size_t)h->n_streams * NVX_CASCADE_STATE_BYTES, h->stream));
    HIP_TRY(hipMemsetAsync(h->d_dd, 0, (size_t)NVX_DEMOD_DOUBLES * h->n_slots * sizeof(double), h->stream));
    // ints: all zero except prev_offset = -1 (decoder.C:30) and the bit-FSM phase = -1 (waiting)
    std::vector<int> ints((size_t)NVX_DEMOD_INTS * h->n_slots, 0);
    for (int i = 0; i < h->n_slots; i++) {
        ints[(size_t)NVX_DI_PREV_OFFSET * h->n_slots + i] = -1;
        ints[(size_t)NVX_DI_PHASE * h->n_slots + i] = -1;
    }
    HIP_TRY(hipMemcpyAsync(h->d_di, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto &s : h->slots) { s.bits.clear(); s.base = 0; s.polled = 0; if (s.sitor) nvx_sitor_reset(s.sitor); }
    if (!h->fill.empty()) std::fill(h->fill.begin(), h->fill.end(), (size_t)0);
    return NVX_OK;
}

// launch cascade + demod over n_frames frames of [n_streams][pitch] packed IQ
static int launch_locked(nvx_handle *h, const void *d_iq, size_t pitch, size_t first_sample, int n_frames, hipStream_t st,
                         bool input_on_stream3 = false)
{
    if (n_frames < 1 || n_frames > h->cfg.max_frames) { nvx_set_error("n_frames %d outside 1..max_frames %d", n_frames, h->cfg.max_frames); return NVX_ERR_ARG; }
    if ((pitch & 3) || (first_sample & 3)) { nvx_set_error("pitch and first sample must be multiples of 4 samples"); return NVX_ERR_ARG; }
    Result &r = h->res[h->launched % RESULT_SLOTS];
    if (r.pending) { int rc = collect_locked(h); if (rc != NVX_OK) return rc; }

    const int wb = (int)(h->wide_launches & 1);
    // Measured (profiles/r01, DESIGN.md tuning log): letting the channeliser of launch k+1 run beside the
    // cascade of launch k (own stream, cascade grid capped at 8 waves per CU to leave LDS) LOSES: 22.2 vs
    // 21.2 ms per step -- the capped cascade and the demodulator slow down by more than the 6 ms hidden.
    // Default: channeliser in front of the cascade on the same stream.  NVX_WB_OVERLAP=1 re-enables it.
    static const bool wb_overlap = getenv("NVX_WB_OVERLAP") && atoi(getenv("NVX_WB_OVERLAP")) == 1;
    if (h->cfg.wideband) {
        // channeliser into sub[wb]; sub[wb] was last read by the cascade two launches ago
        hipStream_t s3 = (wb_overlap || input_on_stream3) ? h->stream3 : st;
        // Input ordering.  Push path: the H2D copy was issued on stream3 itself.  Resident path on the
        // handle's own stream: the caller made the data ready before the call (an event recorded on that
        // stream would also capture the previous cascade and serialise the overlap away).  A caller-supplied
        // stream may have produced the input, so the channeliser waits for it.
        if (!input_on_stream3 && st != h->stream) {
            HIP_TRY(hipEventRecord(h->in_ready[wb], st));
            HIP_TRY(hipStreamWaitEvent(s3, h->in_ready[wb], 0));
        }
        if (h->sub_busy[wb]) HIP_TRY(hipStreamWaitEvent(s3, h->sub_free[wb], 0));
        int rc = nvx_channelise_resident(h->cfg.device, d_iq, pitch, first_sample, h->n_in, (size_t)n_frames * NVX_FRAME_IN,
                                         h->d_whist[wb], h->d_whist[wb ^ 1], h->d_sub[wb], (size_t)h->cfg.max_frames * NVX_FRAME_IN, 0, s3);
        if (rc != NVX_OK) return rc;
        HIP_TRY(hipEventRecord(h->chan_done[wb], s3));
        HIP_TRY(hipStreamWaitEvent(st, h->chan_done[wb], 0));
        d_iq = h->d_sub[wb];
        pitch = (size_t)h->cfg.max_frames * NVX_FRAME_IN;
        first_sample = 0;
    }

    const int yb = (int)(h->launched & 1);              // y3 buffer of this launch
    // Where the demodulator runs (measured, DESIGN.md tuning log).  Its time-parallel front needs LDS, and the
    // persistent cascade grid owns every CU's LDS: beside the NEXT cascade launch it only becomes resident as
    // that drains (loses), so it stays on the cascade's stream.  The sequential FSM kernel is 64 waves without
    // LDS and does run beside the next cascade (NVX_FSM_OVERLAP=1, second stream), but what it hides (0.34 ms)
    // the cascade loses again (20.1 vs 19.8 ms): step time equal, so the default is one stream.
    static const bool fsm_overlap = getenv("NVX_FSM_OVERLAP") && atoi(getenv("NVX_FSM_OVERLAP")) == 1;
    hipStream_t s2 = fsm_overlap ? h->stream2 : st;
    nvx_cascade_args ca{};
    ca.iq = (const uint32_t *)d_iq; ca.pitch = pitch; ca.first_sample = first_sample;
    ca.n_frames = n_frames; ca.n_streams = h->n_streams; ca.chain_masks = h->d_masks;
    ca.state = h->d_cstate; ca.y3 = h->d_y3[yb]; ca.y3_cap = (size_t)h->y3_cap; ca.y3_base = 0;
    ca.queue = h->d_ctrl; ca.status = h->d_ctrl + 1; ca.done = h->d_ctrl + NVX_CASCADE_CTRL_INTS;
    // wideband: leave LDS room beside the persistent cascade grid for the next launch's channeliser workgroups
    ca.max_waves_per_cu = (h->cfg.wideband && wb_overlap) ? 8 : 0;
    nvx_demod_args da{};
    da.y3 = h->d_y3[yb]; da.y3_cap = (size_t)h->y3_cap; da.y3_base = 0; da.n3 = n_frames * NVX_FRAME_Y3;
    da.n_slots = h->n_slots; da.slot_active = h->d_active;
    da.g0 = h->g0; da.dstate = h->d_dd; da.state_i = h->d_di; da.fsm_table = h->d_fsm_tab; da.words = h->d_words;
    da.bits = r.d_bits; da.bits_cap = h->bits_cap; da.nbits = r.d_nbits; da.dphi = h->d_dphi;

    // cascade on `st`: it may not overwrite y3[yb] before the demodulator of two launches ago has read it
    if (h->demod_pending[yb]) HIP_TRY(hipStreamWaitEvent(st, h->demod_done[yb], 0));
    r.timed = h->timing;
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[0], st));
    HIP_TRY(nvx_launch_cascade(&ca, h->cascade_raw, h->nch, st));
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[1], st));
    if (h->cfg.wideband) { HIP_TRY(hipEventRecord(h->sub_free[wb], st)); h->sub_busy[wb] = true; h->wide_launches++; }
    HIP_TRY(hipMemcpyAsync(h->h_status + (h->launched % RESULT_SLOTS), h->d_ctrl + 1, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(h->casc_done[yb], st));
    // demodulator front behind the cascade; it reuses the word buffer the previous launch's FSM reads
    if (h->fsm_pending && s2 != st) HIP_TRY(hipStreamWaitEvent(st, h->fsm_done, 0));
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[2], st));
    HIP_TRY(nvx_launch_demod_front(&da, st));
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[3], st));
    HIP_TRY(hipEventRecord(h->demod_done[yb], st));      // y3[yb] consumed
    h->demod_pending[yb] = true;
    // FSM + bit download behind the front
    if (s2 != st) HIP_TRY(hipStreamWaitEvent(s2, h->demod_done[yb], 0));
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[4], s2));
    HIP_TRY(nvx_launch_demod_fsm(&da, s2));
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[5], s2));
    HIP_TRY(hipEventRecord(h->fsm_done, s2));
    h->fsm_pending = true;
    HIP_TRY(hipMemcpyAsync(r.h_nbits, r.d_nbits, (size_t)h->n_slots * sizeof(int), hipMemcpyDeviceToHost, s2));
    HIP_TRY(hipMemcpyAsync(r.h_bits, r.d_bits, (size_t)h->n_slots * h->bits_cap, hipMemcpyDeviceToHost, s2));
    HIP_TRY(hipEventRecord(r.done, s2));
    r.pending = true;
    h->launched++;
    h->last_n3 = da.n3;
    h->g0 += (unsigned long long)da.n3;
    return NVX_OK;
}

// wait for every launched block, append bits, run the character layer
static int collect_locked(nvx_handle *h)
{
    while (h->collected < h->launched) {
        Result &r = h->res[h->collected % RESULT_SLOTS];
        if (r.pending) {
            HIP_TRY(hipEventSynchronize(r.done));
            if (h->h_status[h->collected % RESULT_SLOTS] != 0) {
                nvx_set_error("FIR cascade work queue: a wait on the previous frame of a stream timed out");
                return NVX_ERR_HIP;
            }
            if (r.timed) {
                HIP_TRY(hipEventElapsedTime(&h->ms[0], r.ev[0], r.ev[1]));
                float fsm_ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&h->ms[1], r.ev[2], r.ev[3]));
                HIP_TRY(hipEventElapsedTime(&fsm_ms, r.ev[4], r.ev[5]));
                h->ms[1] += fsm_ms;                       // "demodulator" = front + FSM
                h->ms_sum[0] += h->ms[0]; h->ms_sum[1] += h->ms[1]; h->ms_count++;
            }
            std::atomic<int> bad_slot{ -1 };
            auto work = [&](int lo, int hi) {
                for (int i = lo; i < hi; i++) {
                    Slot &s = h->slots[i];
                    if (!s.active) continue;
                    int n = r.h_nbits[i];
                    if (n > h->bits_cap * 8) { bad_slot = i; continue; }
                    const uint32_t *pw = (const uint32_t *)(r.h_bits + (size_t)i * h->bits_cap);
                    const size_t at = s.bits.size();
                    s.bits.resize(at + (size_t)n);
                    for (int k = 0; k < n; k++) s.bits[at + k] = ((pw[k >> 5] >> (k & 31)) & 1u) ? 'B' : 'Y';
                    if (s.sitor) nvx_sitor_receive_bits(s.sitor, s.bits.data() + at, (size_t)n);
                    if (s.bits.size() > 2 * h->bit_history) {                // a receiver runs for weeks: bound the poll history
                        const size_t drop = s.bits.size() - h->bit_history;
                        s.bits.erase(0, drop);
                        s.base += drop;
                    }
                }
            };
            static const int host_threads = [] {
                const char *e = getenv("NVX_HOST_THREADS");
                int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
                return n < 1 ? 1 : (n > 16 ? 16 : n);
            }();
            const int nt = (h->n_slots >= 256 && h->cfg.char_layer) ? host_threads : 1;
            if (nt > 1) {
                std::vector<std::thread> pool;
                const int per = (h->n_slots + nt - 1) / nt;
                for (int t = 0; t < nt; t++) pool.emplace_back(work, t * per, std::min(h->n_slots, (t + 1) * per));
                for (auto &t : pool) t.join();
            } else {
                work(0, h->n_slots);
            }
            if (bad_slot >= 0) { nvx_set_error("bit buffer overflow on slot %d", bad_slot.load()); return NVX_ERR_STATE; }
            for (int i = 0; i < h->n_slots; i++) if (!h->slots[i].outbox.empty()) deliver_outbox(h, i / 2, h->slots[i]);
            r.pending = false;
        }
        h->collected++;
    }
    return NVX_OK;
}

static void sitor_sink_impl(nvx_handle *h, int slot, const char *bbbb, const char *message, int freq)
{
    h->slots[slot].outbox.push_back(Message{ bbbb, message, freq });
}

static void deliver_outbox(nvx_handle *h, int stream, Slot &s)
{
    for (auto &m : s.outbox) {
        if (h->cfg.on_message) h->cfg.on_message(h->cfg.user, stream, m.bbbb.c_str(), m.text.c_str(), m.freq);
        else add_message((char *)m.bbbb.c_str(), (char *)m.text.c_str(), m.freq);      // receiver/message_store.h:7
    }
    s.outbox.clear();
}

extern "C" int nvx_process_resident(nvx_handle *h, const void *d_iq, size_t pitch, size_t first_frame, int n_frames, void *hip_stream)
{
    if (!h || !d_iq) { nvx_set_error("nvx_process_resident: null argument"); return NVX_ERR_ARG; }
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
    return launch_locked(h, d_iq, pitch, first_frame * h->frame_in, n_frames, st);
}

extern "C" int nvx_fetch_bits(nvx_handle *h)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    return collect_locked(h);
}

extern "C" size_t nvx_bit_count(nvx_handle *h, int stream, int chain)
{
    if (!h || stream < 0 || stream >= h->n_streams || chain < 0 || chain > 1) return 0;
    std::lock_guard<std::mutex> lk(h->mu);
    const Slot &s = h->slots[2 * stream + chain];
    return s.base + s.bits.size();
}

extern "C" size_t nvx_poll_bits(nvx_handle *h, int stream, int chain, char *out, size_t cap)
{
    if (!h || !out || stream < 0 || stream >= h->n_streams || chain < 0 || chain > 1) return 0;
    std::lock_guard<std::mutex> lk(h->mu);
    Slot &s = h->slots[2 * stream + chain];
    if (s.polled < s.base) s.polled = s.base;                 // the reader fell more than the history behind
    size_t n = std::min(cap, s.base + s.bits.size() - s.polled);
    memcpy(out, s.bits.data() + (s.polled - s.base), n);
    s.polled += n;
    return n;
}

extern "C" void nvx_enable_timing(nvx_handle *h, int enabled) { if (h) h->timing = enabled != 0; }
extern "C" float nvx_last_kernel_ms(nvx_handle *h, int which) { return (h && which >= 0 && which < 2) ? h->ms[which] : -1.f; }
extern "C" int nvx_kernel_time_stats(nvx_handle *h, int which, double *sum_ms, uint64_t *launches, int reset)
{
    if (!h || which < 0 || which > 1) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (sum_ms) *sum_ms = h->ms_sum[which];
    if (launches) *launches = h->ms_count;
    if (reset) { h->ms_sum[0] = h->ms_sum[1] = 0.0; h->ms_count = 0; }
    return NVX_OK;
}

extern "C" int nvx_enable_debug(nvx_handle *h, int enabled)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (enabled && !h->d_dphi) HIP_TRY(hipMalloc(&h->d_dphi, (size_t)h->n_slots * h->y3_cap * sizeof(double)));
    if (!enabled && h->d_dphi) { hipFree(h->d_dphi); h->d_dphi = nullptr; }
    return NVX_OK;
}

extern "C" size_t nvx_debug_y3(nvx_handle *h, int stream, int chain, double *out, size_t cap_pairs)
{
    if (!h || !out || stream < 0 || stream >= h->n_streams || chain < 0 || chain > 1) return 0;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->cfg.device);
    hipDeviceSynchronize();
    size_t n = std::min(cap_pairs, (size_t)h->last_n3);
    if (hipMemcpy(out, h->d_y3[(h->launched + 1) & 1] + (size_t)(2 * stream + chain) * h->y3_cap, n * sizeof(double2), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

extern "C" size_t nvx_debug_dphi(nvx_handle *h, int stream, int chain, double *out, size_t cap)
{
    if (!h || !out || !h->d_dphi || stream < 0 || stream >= h->n_streams || chain < 0 || chain > 1) return 0;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->cfg.device);
    hipDeviceSynchronize();
    size_t n = std::min(cap, (size_t)h->last_n3);
    if (hipMemcpy(out, h->d_dphi + (size_t)(2 * stream + chain) * h->y3_cap, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------- host-input path
// Submit the largest common whole-frame prefix of the current staging set.
static int submit_locked(nvx_handle *h)
{
    size_t minfill = *std::min_element(h->fill.begin(), h->fill.end());
    int frames = (int)std::min<size_t>(minfill / h->frame_in, (size_t)h->cfg.max_frames);
    if (frames < 1) return NVX_OK;
    const int cur = h->cur, nxt = cur ^ 1;
    const size_t take = (size_t)frames * h->frame_in;
    const size_t dpitch = (size_t)h->cfg.max_frames * h->frame_in;
    // the other staging set must have left the copy engine before it is refilled
    if (h->stage_busy[nxt]) { HIP_TRY(hipEventSynchronize(h->stage_free[nxt])); h->stage_busy[nxt] = false; }
    // d_in is reused by every launch: stream order makes its previous reader finish first (the cascade on
    // h->stream, or in wideband mode the channeliser on stream3, which is why the copy goes there)
    hipStream_t cs = h->cfg.wideband ? h->stream3 : h->stream;
    HIP_TRY(hipMemcpy2DAsync(h->d_in, dpitch * 4, h->h_stage[cur], h->stage_cap * 4, take * 4, (size_t)h->n_in,
                             hipMemcpyHostToDevice, cs));
    HIP_TRY(hipEventRecord(h->stage_free[cur], cs));
    h->stage_busy[cur] = true;
    int rc = launch_locked(h, h->d_in, dpitch, 0, frames, h->stream, h->cfg.wideband != 0);
    if (rc != NVX_OK) return rc;
    // carry what was not submitted over to the other set
    for (int s = 0; s < h->n_in; s++) {
        size_t rest = h->fill[s] - take;
        if (rest) memcpy(h->h_stage[nxt] + (size_t)s * h->stage_cap, h->h_stage[cur] + (size_t)s * h->stage_cap + take, rest * 4);
        h->fill[s] = rest;
    }
    h->cur = nxt;
    return NVX_OK;
}

template <typename F>
static int push_common(nvx_handle *h, int stream, size_t n, F copy_in)
{
    if (!h || stream < 0 || stream >= h->n_in) { nvx_set_error("nvx_push: bad stream"); return NVX_ERR_ARG; }
    if (!h->cfg.push_mode) { nvx_set_error("nvx_push: handle was not created with push_mode"); return NVX_ERR_STATE; }
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    size_t done = 0;
    while (done < n) {
        size_t room = h->stage_cap - h->fill[stream];
        if (room == 0) {
            int rc = submit_locked(h);
            if (rc != NVX_OK) return rc;
            room = h->stage_cap - h->fill[stream];
            if (room == 0) { nvx_set_error("stream %d is a whole staging buffer ahead of the slowest stream", stream); return NVX_ERR_FULL; }
        }
        size_t m = std::min(room, n - done);
        copy_in(h->h_stage[h->cur] + (size_t)stream * h->stage_cap + h->fill[stream], done, m);
        h->fill[stream] += m;
        done += m;
        size_t minfill = *std::min_element(h->fill.begin(), h->fill.end());
        if (minfill >= h->frame_in) { int rc = submit_locked(h); if (rc != NVX_OK) return rc; }
    }
    return NVX_OK;
}

extern "C" int nvx_push_iq(nvx_handle *h, int stream, const int16_t *iq, size_t n)
{
    return push_common(h, stream, n, [&](uint32_t *dst, size_t off, size_t m) { memcpy(dst, iq + 2 * off, m * 4); });
}

extern "C" int nvx_push_planar(nvx_handle *h, int stream, const int16_t *xi, const int16_t *xq, size_t n)
{
    return push_common(h, stream, n, [&](uint32_t *dst, size_t off, size_t m) {
        for (size_t k = 0; k < m; k++)                     // interleave as capt_sched.c:120-129 does
            dst[k] = (uint32_t)(uint16_t)xi[off + k] | ((uint32_t)(uint16_t)xq[off + k] << 16);
    });
}

extern "C" int nvx_flush(nvx_handle *h)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->cfg.push_mode) {
        for (;;) {
            size_t minfill = *std::min_element(h->fill.begin(), h->fill.end());
            if (minfill < h->frame_in) break;
            int rc = submit_locked(h);
            if (rc != NVX_OK) return rc;
        }
    }
    return collect_locked(h);
}

// ------------------------------------------------------------ device helpers
extern "C" int nvx_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
extern "C" void *nvx_device_alloc(int device, size_t bytes)
{
    if (select_device(device) != NVX_OK) return nullptr;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { nvx_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
extern "C" void nvx_device_free(int device, void *p) { if (p && select_device(device) == NVX_OK) hipFree(p); }
extern "C" int nvx_memcpy_h2d(int device, void *d, const void *s, size_t n)
{
    int rc = select_device(device); if (rc != NVX_OK) return rc;
    HIP_TRY(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); return NVX_OK;
}
extern "C" int nvx_memcpy_d2h(int device, void *d, const void *s, size_t n)
{
    int rc = select_device(device); if (rc != NVX_OK) return rc;
    HIP_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToHost)); return NVX_OK;
}
extern "C" int nvx_device_sync(int device)
{
    int rc = select_device(device); if (rc != NVX_OK) return rc;
    HIP_TRY(hipDeviceSynchronize()); return NVX_OK;
}

// ------------------------------------------------------------ synthetic source
extern "C" int nvx_synth_device(int device, const nvx_synth_stream *streams, int n_streams,
                                uint32_t sample_rate, size_t n, void *d_out, size_t pitch)
{
    if (!streams || n_streams < 1 || !d_out || (sample_rate != NVX_RATE_RAW && sample_rate != NVX_RATE_IN) || pitch < n || (pitch & 3)) {
        nvx_set_error("nvx_synth_device: bad argument"); return NVX_ERR_ARG;
    }
    int rc = select_device(device); if (rc != NVX_OK) return rc;
    const uint32_t spb = sample_rate / 100;
    std::vector<nvx_synth_desc> desc(n_streams);
    std::vector<nvx_period> pool;
    for (int s = 0; s < n_streams; s++) {
        const nvx_synth_stream &st = streams[s];
        if (st.n_carriers < 0 || st.n_carriers > NVX_SYNTH_MAX_CARRIERS) { nvx_set_error("nvx_synth_device: stream %d: bad carrier count", s); return NVX_ERR_ARG; }
        nvx_synth_desc &d = desc[s];
        memset(&d, 0, sizeof d);
        d.seed = st.seed; d.noise_amp = st.noise_amp; d.n_carriers = st.n_carriers;
        for (int c = 0; c < st.n_carriers; c++) {
            if (st.carrier[c].bit_offset >= spb) { nvx_set_error("nvx_synth_device: bit_offset >= samples per bit"); return NVX_ERR_ARG; }
            size_t periods = (n + st.carrier[c].bit_offset) / spb + 2;
            d.amp[c] = st.carrier[c].amplitude; d.bit_offset[c] = st.carrier[c].bit_offset;
            d.pool_off[c] = (uint32_t)pool.size();
            pool.resize(pool.size() + periods);
            nvx_synth_periods(&st.carrier[c], sample_rate, 0, periods, pool.data() + d.pool_off[c]);
        }
    }
    nvx_synth_desc *d_desc = nullptr; nvx_period *d_pool = nullptr;
    HIP_TRY(hipMalloc(&d_desc, desc.size() * sizeof(nvx_synth_desc)));
    hipError_t e = hipMalloc(&d_pool, std::max<size_t>(pool.size(), 1) * sizeof(nvx_period));
    if (e != hipSuccess) { hipFree(d_desc); nvx_set_error("hipMalloc pool failed: %s", hipGetErrorString(e)); return NVX_ERR_NOMEM; }
    hipMemcpy(d_desc, desc.data(), desc.size() * sizeof(nvx_synth_desc), hipMemcpyHostToDevice);
    if (!pool.empty()) hipMemcpy(d_pool, pool.data(), pool.size() * sizeof(nvx_period), hipMemcpyHostToDevice);
    nvx_synth_args a{};
    a.desc = d_desc; a.pool = d_pool; a.out = (uint32_t *)d_out; a.pitch = pitch; a.n = n; a.spb = spb;
    e = nvx_launch_synth(&a, n_streams, nullptr);
    hipError_t e2 = hipDeviceSynchronize();
    hipFree(d_desc); hipFree(d_pool);
    if (e != hipSuccess || e2 != hipSuccess) {
        nvx_set_error("synth kernel failed: %s", hipGetErrorString(e != hipSuccess ? e : e2)); return NVX_ERR_HIP;
    }
    return NVX_OK;
}

// ------------------------------------------------------------ wideband front-end
extern "C" void *nvx_handle_stream(nvx_handle *h) { return h ? (void *)h->stream : nullptr; }

static struct ChanTiming {
    std::mutex mu;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool, pending;
    double sum_ms = 0.0; uint64_t n = 0;
} g_ct;

extern "C" void nvx_channelise_timing(int enable) { std::lock_guard<std::mutex> lk(g_ct.mu); g_ct.on = enable != 0; }

extern "C" int nvx_channelise_time_stats(double *sum_ms, uint64_t *launches, int reset)
{
    std::lock_guard<std::mutex> lk(g_ct.mu);
    for (auto &p : g_ct.pending) {
        HIP_TRY(hipEventSynchronize(p.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        g_ct.sum_ms += ms; g_ct.n++;
        g_ct.pool.push_back(p);
    }
    g_ct.pending.clear();
    if (sum_ms) *sum_ms = g_ct.sum_ms;
    if (launches) *launches = g_ct.n;
    if (reset) { g_ct.sum_ms = 0.0; g_ct.n = 0; }
    return NVX_OK;
}

extern "C" int nvx_channelise_resident(int device, const void *d_raw, size_t pitch_raw, size_t first_sample, int n_wide,
                                       size_t n_out, const void *d_hist_in, void *d_hist_out, void *d_sub, size_t pitch_sub,
                                       size_t sub_first, void *hip_stream)
{
    if (!d_raw || !d_sub || n_wide < 1 || n_out == 0 || (n_out % 64) || (pitch_raw & 3) || (first_sample & 3) ||
        (d_hist_in && d_hist_in == d_hist_out)) {
        nvx_set_error("nvx_channelise_resident: bad argument (n_out must be a multiple of 64, pitches/offsets of 4)");
        return NVX_ERR_ARG;
    }
    int rc = select_device(device); if (rc != NVX_OK) return rc;
    nvx_channelise_args a{};
    a.raw = (const uint32_t *)d_raw; a.pitch_raw = pitch_raw; a.first_sample = first_sample; a.n_wide = n_wide; a.n_out = n_out;
    a.hist_in = (const uint32_t *)d_hist_in; a.hist_out = (uint32_t *)d_hist_out;
    a.sub = (uint32_t *)d_sub; a.pitch_sub = pitch_sub; a.sub_first = sub_first;
    // enough blocks to fill the chip several times over, long enough spans to amortise the 40-sample halo
    const size_t n_chunks = n_out / 64;
    size_t cpb = (n_chunks * (size_t)n_wide + 16383) / 16384;
    if (cpb < 8) cpb = 8;
    a.chunks_per_block = (int)std::min<size_t>(cpb, n_chunks);
    std::pair<hipEvent_t, hipEvent_t> ev{ nullptr, nullptr };
    bool timed = false;
    {
        std::lock_guard<std::mutex> lk(g_ct.mu);
        if (g_ct.on) {
            if (g_ct.pool.empty()) { HIP_TRY(hipEventCreate(&ev.first)); HIP_TRY(hipEventCreate(&ev.second)); }
            else { ev = g_ct.pool.back(); g_ct.pool.pop_back(); }
            timed = true;
        }
    }
    if (timed) HIP_TRY(hipEventRecord(ev.first, (hipStream_t)hip_stream));
    HIP_TRY(nvx_launch_channelise(&a, (hipStream_t)hip_stream));
    if (timed) {
        HIP_TRY(hipEventRecord(ev.second, (hipStream_t)hip_stream));
        std::lock_guard<std::mutex> lk(g_ct.mu);
        g_ct.pending.push_back(ev);
    }
    return NVX_OK;
}

// ------------------------------------------------------------------ WAV path
extern "C" int nvx_decode_wav(nvx_handle *h, int stream, const char *filename)
{
    if (!h || !filename) { nvx_set_error("nvx_decode_wav: null argument"); return NVX_ERR_ARG; }
    if (!h->cfg.push_mode) { nvx_set_error("nvx_decode_wav: handle needs push_mode"); return NVX_ERR_STATE; }
    nvx_wav *w = nvx_wav_open(filename, NVX_WAV_OPEN_READ);
    if (!w) { nvx_set_error("nvx_decode_wav: %s", nvx_wav_err()); return NVX_ERR_IO; }
    const uint32_t want = (h->cfg.raw_rate || h->cfg.wideband) ? NVX_RATE_RAW : NVX_RATE_IN;
    if (nvx_wav_get_num_channels(w) != 2 || nvx_wav_get_sample_size(w) != 2 || nvx_wav_get_format(w) != 1 ||
        nvx_wav_get_sample_rate(w) != want) {
        nvx_set_error("nvx_decode_wav: need 2-channel 16-bit PCM at %u Hz (capt_sched.c:91-95)", want);
        nvx_wav_close(w); return NVX_ERR_IO;
    }
    std::vector<int16_t> buf(2 * 65536);
    size_t total = 0, got;
    int rc = NVX_OK;
    while ((got = nvx_wav_read(w, buf.data(), 65536)) > 0) {
        rc = nvx_push_iq(h, stream, buf.data(), got);
        if (rc != NVX_OK) break;
        total += got;
    }
    nvx_wav_close(w);
    if (rc != NVX_OK) return rc;
    size_t pad = (h->frame_in - total % h->frame_in) % h->frame_in;       // silence up to a whole frame
    std::fill(buf.begin(), buf.end(), (int16_t)0);
    while (pad) {
        size_t m = std::min<size_t>(pad, 65536);
        rc = nvx_push_iq(h, stream, buf.data(), m);
        if (rc != NVX_OK) return rc;
        pad -= m;
    }
    rc = nvx_flush(h);
    if (rc != NVX_OK) return rc;
    return (int)((total + h->frame_in - 1) / h->frame_in);
}

// ===========================================================================
// reference-compatible push surface + stream callback (sections A, B)
// ===========================================================================
static nvx_handle *g_shim = nullptr;
static std::mutex g_shim_mu;                     // callback re-entrancy (capt_sched.c:111)
static int16_t g_shim_buf[2 * 4096];
static size_t g_shim_n = 0;

static void shim_fatal(const char *what)
{
    fprintf(stderr, "navtex_amd: %s: %s\n", what, nvx_last_error());
    abort();                                     // void reference entry points cannot report errors
}

static void shim_require(void)
{
    if (g_shim) return;
    nvx_config c; nvx_config_default(&c);
    c.n_streams = 1; c.raw_rate = 0; c.chain_mask = NVX_CHAIN_518 | NVX_CHAIN_490;   // nav_sched.C:10-17
    c.max_frames = 4; c.char_layer = 1; c.push_mode = 1;
    if (const char *d = getenv("NAVTEX_AMD_DEVICE")) c.device = atoi(d);
    if (nvx_create(&c, &g_shim) != NVX_OK) shim_fatal("cannot create the GPU pipeline");
}

static void shim_drain(void)
{
    if (g_shim_n && nvx_push_iq(g_shim, 0, g_shim_buf, g_shim_n) != NVX_OK) shim_fatal("push failed");
    g_shim_n = 0;
}

extern "C" void init_fir_filter1(void)           // receiver/fir1cpp.C:65-77
{
    std::lock_guard<std::mutex> lk(g_shim_mu);
    shim_require();
    g_shim_n = 0;
    if (nvx_reset(g_shim) != NVX_OK) shim_fatal("reset failed");
}

extern "C" void init_fir2_wrapper(void)          // receiver/nav_sched.C:19-22
{
    std::lock_guard<std::mutex> lk(g_shim_mu);
    shim_require();                              // the object graph already exists; nothing else to wire
}

extern "C" void sample_in_1(double sample_I, double sample_Q)   // receiver/fir1cpp.C:80
{
    // capt_sched.c:511 passes (double) of int16 values; the cast back is exact
    if (!g_shim) { std::lock_guard<std::mutex> lk(g_shim_mu); shim_require(); }
    g_shim_buf[2 * g_shim_n] = (int16_t)sample_I;
    g_shim_buf[2 * g_shim_n + 1] = (int16_t)sample_Q;
    if (++g_shim_n == 4096) shim_drain();
}

extern "C" int nvx_shim_flush(void)
{
    std::lock_guard<std::mutex> lk(g_shim_mu);
    if (!g_shim) { nvx_set_error("shim not initialised"); return NVX_ERR_STATE; }
    shim_drain();
    return nvx_flush(g_shim);
}

extern "C" size_t nvx_shim_bits(int chain, char *out, size_t cap)
{
    if (!g_shim) return 0;
    return nvx_poll_bits(g_shim, 0, chain, out, cap);
}

extern "C" void nvx_StreamACallback(short *xi, short *xq, void *params, unsigned int numSamples,
                                    unsigned int reset, void *cbContext)
{
    (void)params; (void)reset;                   // ignored by the reference too (capt_sched.c:105-148)
    std::lock_guard<std::mutex> lk(g_shim_mu);
    nvx_handle *h = (nvx_handle *)cbContext;
    if (!h) { shim_require(); shim_drain(); h = g_shim; }
    if (nvx_push_planar(h, 0, xi, xq, numSamples) != NVX_OK) shim_fatal("stream callback push failed");
}

// ===========================================================================
// live-capture ring (section B'): capt_sched.c's producer / ring / consumer
// ===========================================================================
struct nvx_capture {
    nvx_handle *h = nullptr;
    int stream = 0;
    std::vector<int16_t> ring;                  // interleaved I,Q (capt_sched.c:443: shorts)
    size_t cap = 0;                             // complex samples
    std::atomic<uint64_t> head{ 0 }, tail{ 0 }; // samples ever written / ever read
    std::atomic<uint64_t> received{ 0 }, dropped{ 0 }, consumed{ 0 };
    std::mutex prod_mu;                         // callback re-entrancy (capt_sched.c:111)
    std::mutex cv_mu; std::condition_variable cv;
    std::atomic<bool> stop{ false }, paused{ false };
    std::atomic<int> error{ NVX_OK };
    std::mutex rec_mu; nvx_wav *rec = nullptr;  // debug recording of what the consumer hands on (capt_sched.c:87-101, 516)
    std::thread worker;
};

static void capture_consumer(nvx_capture *c)
{
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(c->cv_mu);
            c->cv.wait_for(lk, std::chrono::milliseconds(50), [&] {     // the reference polls every 50 ms (capt_sched.c:486)
                return c->stop.load() || (!c->paused.load() && c->head.load() != c->tail.load());
            });
        }
        if (c->paused.load() && !c->stop.load()) continue;
        uint64_t t = c->tail.load(), hd = c->head.load();
        while (t != hd) {                                                // contiguous spans, wrap split as capt_sched.c:494-503
            size_t at = (size_t)(t % c->cap);
            size_t n = (size_t)std::min<uint64_t>(hd - t, c->cap - at);
            int rc = nvx_push_iq(c->h, c->stream, c->ring.data() + 2 * at, n);
            if (rc != NVX_OK) { c->error.store(rc); c->stop.store(true); return; }
            {
                std::lock_guard<std::mutex> lk(c->rec_mu);
                if (c->rec && nvx_wav_write(c->rec, c->ring.data() + 2 * at, n) != n) {     // disk full etc.: stop recording, keep decoding
                    nvx_wav_close(c->rec); c->rec = nullptr;
                }
            }
            t += n;
            c->tail.store(t);
            c->consumed.fetch_add(n);
        }
        if (c->stop.load() && c->head.load() == c->tail.load()) return;
    }
}

extern "C" int nvx_capture_start(nvx_handle *h, int stream, double ring_seconds, nvx_capture **out)
{
    if (!h || !out || stream < 0 || stream >= h->n_in || !(ring_seconds > 0)) { nvx_set_error("nvx_capture_start: bad argument"); return NVX_ERR_ARG; }
    if (!h->cfg.push_mode) { nvx_set_error("nvx_capture_start: handle needs push_mode"); return NVX_ERR_STATE; }
    nvx_capture *c = new nvx_capture();
    c->h = h; c->stream = stream;
    const double rate = (h->cfg.raw_rate || h->cfg.wideband) ? (double)NVX_RATE_RAW : (double)NVX_RATE_IN;
    c->cap = (size_t)(ring_seconds * rate);                              // capt_sched.c:443: rate * seconds
    if (c->cap < 16) c->cap = 16;
    c->ring.assign(2 * c->cap, 0);
    c->worker = std::thread(capture_consumer, c);
    *out = c;
    return NVX_OK;
}

extern "C" void nvx_capture_callback(short *xi, short *xq, void *params, unsigned int numSamples, unsigned int reset, void *cbContext)
{
    (void)params; (void)reset;
    nvx_capture *c = (nvx_capture *)cbContext;
    if (!c || !xi || !xq) return;
    std::lock_guard<std::mutex> lk(c->prod_mu);
    c->received.fetch_add(numSamples);
    const uint64_t hd = c->head.load();
    const uint64_t room = c->cap - (hd - c->tail.load());
    const size_t n = (size_t)std::min<uint64_t>(numSamples, room);
    if (n < numSamples) c->dropped.fetch_add(numSamples - n);            // overrun: newest samples are dropped
    for (size_t k = 0; k < n; k++) {                                     // interleave, capt_sched.c:120-129
        const size_t at = (size_t)((hd + k) % c->cap);
        c->ring[2 * at] = xi[k];
        c->ring[2 * at + 1] = xq[k];
    }
    c->head.store(hd + n);
    c->cv.notify_one();
}

extern "C" int nvx_capture_record(nvx_capture *c, const char *filename)
{
    if (!c) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(c->rec_mu);
    if (c->rec) { nvx_wav_close(c->rec); c->rec = nullptr; }
    if (!filename) return NVX_OK;
    nvx_wav *w = nvx_wav_open(filename, NVX_WAV_OPEN_WRITE);
    if (!w) { nvx_set_error("nvx_capture_record: %s", nvx_wav_err()); return NVX_ERR_IO; }
    nvx_wav_set_format(w, 1);                                            // PrepWav, capt_sched.c:87-96
    nvx_wav_set_num_channels(w, 2);
    nvx_wav_set_sample_rate(w, (c->h->cfg.raw_rate || c->h->cfg.wideband) ? NVX_RATE_RAW : NVX_RATE_IN);
    nvx_wav_set_sample_size(w, sizeof(short));
    c->rec = w;
    return NVX_OK;
}

extern "C" void nvx_capture_pause(nvx_capture *c, int paused)
{
    if (!c) return;
    c->paused.store(paused != 0);
    c->cv.notify_one();
}

extern "C" void nvx_capture_stats(nvx_capture *c, uint64_t *received, uint64_t *dropped, uint64_t *consumed)
{
    if (!c) return;
    if (received) *received = c->received.load();
    if (dropped) *dropped = c->dropped.load();
    if (consumed) *consumed = c->consumed.load();
}

extern "C" int nvx_capture_stop(nvx_capture *c)
{
    if (!c) return NVX_ERR_ARG;
    c->paused.store(false);
    c->stop.store(true);
    c->cv.notify_one();
    if (c->worker.joinable()) c->worker.join();
    int rc = c->error.load();
    if (rc == NVX_OK) rc = nvx_flush(c->h);
    if (c->rec) nvx_wav_close(c->rec);                                   // EndWav, capt_sched.c:98-101
    delete c;
    return rc;
}
