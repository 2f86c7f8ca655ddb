// nvx_api.cpp -- core of the host runtime behind the C ABI of include/navtex_amd.h: errors,
// device selection, the handle (device buffers, carried state, result ring), launch / collect,
// the device-resident block API, instrumentation and the small device helpers.
// Host input: nvx_push.cpp.  Reference-shaped surface: nvx_shim.cpp.  Capture ring: nvx_capture.cpp.
//
// There is no CPU implementation of the signal path in this library: every
// entry point that needs the GPU returns NVX_ERR_NODEV / NVX_ERR_HIP without it
// (and the void reference-shaped entry points print and abort()).
#include "nvx_handle.h"
#include <chrono>
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
int64_t nvx_now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define NVX_STR2(x) #x
#define NVX_STR(x) NVX_STR2(x)
extern "C" const char *nvx_version(void) { return "navtex_amd " NVX_STR(NVX_ABI_VERSION) ".0 (gfx950)"; }
extern "C" int nvx_abi_version(void) { return NVX_ABI_VERSION; }

int nvx_select_device(int device)
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


int nvx_check_device_span(const void *p, size_t bytes, const char *what)
{
    hipDeviceptr_t base = nullptr; size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) { (void)hipGetLastError(); return NVX_OK; }
    const size_t off = (size_t)((const char *)p - (const char *)base);
    if (off > size || bytes > size - off) {
        nvx_set_error("%s: %zu bytes from %p leave the allocation they lie in (%zu bytes from %p): the launch would fault", what, bytes, p, size, (void *)base);
        return NVX_ERR_ARG;
    }
    return NVX_OK;
}

// The character layers of different chains run on worker threads; their messages are
// parked per slot and handed to the user's sink afterwards, in slot order, by the
// collecting thread (the reference calls add_message from its single DSP thread).
static void sitor_sink(void *user, const char *bbbb, const char *message, int freq);

static void deliver_outbox(nvx_handle *h, int stream, Slot &s);

static void sitor_sink_impl(nvx_handle *h, int slot, const char *bbbb, const char *message, int freq);

static void sitor_sink(void *user, const char *bbbb, const char *message, int freq)
{
    SinkCtx *c = (SinkCtx *)user;
    sitor_sink_impl(c->h, c->slot, bbbb, message, freq);
}

extern "C" void nvx_config_default(nvx_config *c)
{
    memset(c, 0, sizeof *c);
    c->struct_size = (uint32_t)sizeof *c;
    c->device = 0; c->n_streams = 1; c->raw_rate = 0;
    c->chain_mask = NVX_CHAIN_518 | NVX_CHAIN_490;
    c->max_frames = 1; c->char_layer = 1; c->push_mode = 0;
}

static void free_handle(nvx_handle *h)
{
    if (!h) return;
    hipSetDevice(h->cfg.device);
    if (h->launch_done_valid) hipEventSynchronize(h->launch_done);      // the last launch may sit on a caller's stream
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->stream2) hipStreamSynchronize(h->stream2);
    hipFree(h->d_whist[0]); hipFree(h->d_whist[1]);
    hipFree(h->d_y2[0]); hipFree(h->d_y2[1]); hipFree(h->d_y2row);
    hipFree(h->d_masks); hipFree(h->d_active); hipFree(h->d_cstate[0]); hipFree(h->d_cstate[1]); hipFree(h->d_y3[0]); hipFree(h->d_y3[1]);
    for (int i = 0; i < 2; i++) { if (h->casc_done[i]) hipEventDestroy(h->casc_done[i]); if (h->demod_done[i]) hipEventDestroy(h->demod_done[i]); }
    if (h->launch_done) hipEventDestroy(h->launch_done);
    hipFree(h->d_ties); if (h->h_ties) hipHostFree(h->h_ties);
    hipFree(h->d_dd[0]); hipFree(h->d_dd[1]); hipFree(h->d_di); hipFree(h->d_fsm_tab); hipFree(h->d_dphi); hipFree(h->d_in); hipFree(h->d_words); hipFree(h->d_ctrl);
    if (h->h_status) hipHostFree(h->h_status);
    for (auto &r : h->res) {
        hipFree(r.d_bits); hipFree(r.d_nbits);
        if (r.h_bits) hipHostFree(r.h_bits);
        if (r.h_nbits) hipHostFree(r.h_nbits);
        hipFree(r.d_part); if (r.h_part) hipHostFree(r.h_part);
        if (r.copied) hipEventDestroy(r.copied);
        if (r.done) hipEventDestroy(r.done);
        for (int i = 0; i < 8; i++) if (r.ev[i]) hipEventDestroy(r.ev[i]);
    }
    for (int i = 0; i < 2; i++) {
        if (h->h_stage[i]) hipHostFree(h->h_stage[i]);
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
    if (!cfg || !out) { nvx_set_error("nvx_create: null argument"); return NVX_ERR_ARG; }
    *out = nullptr;
    // (the first member of every layout: nothing else of the caller's struct is read before this has passed)
    if (cfg->struct_size != sizeof(nvx_config)) {
        nvx_set_error("nvx_create: nvx_config.struct_size is %u, this library's nvx_config has %zu bytes (ABI %d): the caller was built against "
                      "another navtex_amd.h, or did not start from nvx_config_default", cfg->struct_size, sizeof(nvx_config), NVX_ABI_VERSION);
        return NVX_ERR_ARG;
    }
    if (cfg->n_streams < 1 || cfg->max_frames < 1) { nvx_set_error("nvx_create: bad config"); return NVX_ERR_ARG; }
    if (cfg->stage0_order != 0 && cfg->stage0_order != 1 && cfg->stage0_order != 3) { nvx_set_error("nvx_create: stage0_order %d (1 or 3)", cfg->stage0_order); return NVX_ERR_ARG; }
    if (cfg->stage0_order == 3 && !(cfg->raw_rate && !cfg->wideband)) { nvx_set_error("nvx_create: stage0_order 3 needs raw_rate input (a wideband handle has its channeliser, 252 kS/s input no stage 0)"); return NVX_ERR_ARG; }
    if (!cfg->push_mode && (cfg->eager_launch || cfg->stall_timeout_ms)) { nvx_set_error("nvx_create: eager_launch / stall_timeout_ms belong to push_mode handles"); return NVX_ERR_ARG; }
    int rc = nvx_select_device(cfg->device);
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
    h->parity.assign(h->n_in, 0);
    h->g0s.assign(h->n_in, 0);
    h->ended.assign(h->n_in, 0);
    h->arrival.assign(h->n_in, nullptr);
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
    CR_TRY(hipEventCreateWithFlags(&h->launch_done, hipEventDisableTiming));
    std::vector<uint8_t> active(h->n_slots);
    for (int i = 0; i < h->n_slots; i++) active[i] = h->slots[i].active;
    CR_TRY(hipMalloc(&h->d_masks, h->n_streams));
    CR_TRY(hipMalloc(&h->d_active, h->n_slots));
    CR_TRY(hipMemcpy(h->d_masks, h->masks.data(), h->n_streams, hipMemcpyHostToDevice));
    CR_TRY(hipMemcpy(h->d_active, active.data(), h->n_slots, hipMemcpyHostToDevice));
    for (int i = 0; i < 2; i++) CR_TRY(hipMalloc(&h->d_cstate[i], (size_t)h->n_streams * NVX_CASCADE_STATE_BYTES));
    for (int i = 0; i < 2; i++) CR_TRY(hipMalloc(&h->d_y3[i], (size_t)h->n_slots * h->y3_cap * sizeof(double2)));
    if (cfg->wideband) {
        // the fused wideband kernel's waves end at FIR2: a row of 9 kS/s fp64 pairs per ACTIVE chain, two buffers (nvx_kernels.h)
        std::vector<int> rows(h->n_slots, -1);
        for (int i = 0; i < h->n_slots; i++) if (h->slots[i].active) rows[i] = h->y2_rows++;
        h->y2_pitch = (size_t)NVX_Y2_PREFIX + (size_t)cfg->max_frames * NVX_Y2_PER_FRAME;
        CR_TRY(hipMalloc(&h->d_y2row, (size_t)h->n_slots * sizeof(int)));
        CR_TRY(hipMemcpy(h->d_y2row, rows.data(), (size_t)h->n_slots * sizeof(int), hipMemcpyHostToDevice));
        h->y2row = rows;
        for (int i = 0; i < 2; i++) CR_TRY(hipMalloc(&h->d_y2[i], (size_t)h->y2_rows * h->y2_pitch * sizeof(double2)));
    }
    for (int i = 0; i < 2; i++) CR_TRY(hipMalloc(&h->d_dd[i], (size_t)NVX_DEMOD_DOUBLES * h->n_slots * sizeof(double)));
    CR_TRY(hipMalloc(&h->d_di, (size_t)NVX_DEMOD_INTS * h->n_slots * sizeof(int)));
    CR_TRY(hipMalloc(&h->d_fsm_tab, NVX_FSM_TABLE_ALLOC * sizeof(uint32_t)));
    CR_TRY(hipMemcpy(h->d_fsm_tab, nvx_fsm_table_host(), NVX_FSM_TABLE_ALLOC * sizeof(uint32_t), hipMemcpyHostToDevice));
    CR_TRY(hipMalloc(&h->d_words, (size_t)(h->y3_cap / 9) * h->n_slots * sizeof(unsigned short)));
    CR_TRY(hipMalloc(&h->d_ctrl, (size_t)(NVX_CASCADE_CTRL_INTS + h->n_streams) * sizeof(int)));
    CR_TRY(hipMalloc(&h->d_ties, sizeof(nvx_tie_stats)));
    CR_TRY(hipHostMalloc((void **)&h->h_ties, sizeof(nvx_tie_stats), hipHostMallocDefault));
    CR_TRY(hipHostMalloc((void **)&h->h_status, RESULT_SLOTS * NVX_STATUS_INTS * sizeof(int), hipHostMallocDefault));
    memset(h->h_status, 0, RESULT_SLOTS * NVX_STATUS_INTS * sizeof(int));
    for (auto &r : h->res) {
        CR_TRY(hipMalloc(&r.d_bits, (size_t)h->n_slots * h->bits_cap));
        CR_TRY(hipMalloc(&r.d_nbits, (size_t)h->n_slots * sizeof(int)));
        CR_TRY(hipHostMalloc((void **)&r.h_bits, (size_t)h->n_slots * h->bits_cap, hipHostMallocDefault));
        CR_TRY(hipHostMalloc((void **)&r.h_nbits, (size_t)h->n_slots * sizeof(int), hipHostMallocDefault));
        CR_TRY(hipMalloc(&r.d_part, (size_t)h->n_in * sizeof(nvx_part)));
        CR_TRY(hipHostMalloc((void **)&r.h_part, (size_t)h->n_in * sizeof(nvx_part), hipHostMallocDefault));
        CR_TRY(hipEventCreateWithFlags(&r.copied, hipEventDisableTiming));
        CR_TRY(hipEventCreateWithFlags(&r.done, hipEventDisableTiming));
        for (int i = 0; i < 8; i++) CR_TRY(hipEventCreate(&r.ev[i]));
    }
    if (cfg->wideband)         // the channeliser's 40-sample halo in front of a launch, two blocks by stream parity (nvx_kernels.h)
        for (int i = 0; i < 2; i++) CR_TRY(hipMalloc(&h->d_whist[i], (size_t)h->n_in * 40 * 4));
    if (cfg->push_mode) {
        h->stage_cap = (size_t)(cfg->max_frames + 1) * h->frame_in;
        for (int i = 0; i < 2; i++) {
            CR_TRY(hipHostMalloc((void **)&h->h_stage[i], (size_t)h->n_in * h->stage_cap * 4, hipHostMallocDefault));
            h->set_launch[i].assign(h->n_in, 0);
        }
        CR_TRY(hipMalloc(&h->d_in, (size_t)h->n_in * cfg->max_frames * h->frame_in * 4));
        h->fill.assign(h->n_in, 0);
        h->cur.assign(h->n_in, 0);
        h->active.assign(h->n_in, 1);
        h->writing.assign(h->n_in, 0);
        h->pushing.assign(h->n_in, 0);
        h->closing.assign(h->n_in, 0);
        h->last_push_ns.assign(h->n_in, nvx_now_ns());
        h->stall_ns.assign(h->n_in, cfg->stall_timeout_ms < 0 ? 0 : (int64_t)(cfg->stall_timeout_ms ? cfg->stall_timeout_ms : 2000) * 1000000);
    }
#undef CR_TRY
    rc = nvx_reset(h);
    if (rc != NVX_OK) { free_handle(h); return rc; }
    *out = h;
    return NVX_OK;
}


extern "C" int nvx_reset(nvx_handle *h)
{
    if (!h) return NVX_ERR_ARG;
    std::unique_lock<std::mutex> lk(h->mu);
    StreamClose closing(h, lk, 0, h->n_in);              // push calls in progress end first; later ones start on the fresh streams
    StagingQuiesce quiet(h, lk);                         // no push is copying into the staging sets while they are emptied
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->launch_done_valid) HIP_TRY(hipEventSynchronize(h->launch_done));   // launches on a caller's stream included
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream2));
    h->launch_done_valid = false;
    for (auto &r : h->res) r.pending = false;
    h->collected = h->launched;
    std::fill(h->parity.begin(), h->parity.end(), (uint8_t)0);
    std::fill(h->g0s.begin(), h->g0s.end(), 0ull);
    h->diverged = false;
    // a reset drops whatever was staged: the frames of an attached capture ring and the handle's no longer line up, so its
    // latency bookkeeping ends here (nvx_capture_latency keeps what it has; a new nvx_capture_start books again)
    for (ArrivalClock *ac : h->arrival)
        if (ac) { std::lock_guard<std::mutex> al(ac->mu); ac->base = UINT64_MAX; }
    h->demod_pending[0] = h->demod_pending[1] = false;
    h->poisoned = false; h->poison_why.clear();
    std::fill(h->ended.begin(), h->ended.end(), (uint8_t)0);
    for (int i = 0; i < 2 && h->d_whist[i]; i++) HIP_TRY(hipMemsetAsync(h->d_whist[i], 0, (size_t)h->n_in * 40 * 4, h->stream));
    for (int i = 0; i < 2; i++) HIP_TRY(hipMemsetAsync(h->d_cstate[i], 0, (size_t)h->n_streams * NVX_CASCADE_STATE_BYTES, h->stream));
    // FIR3's history: silence in front of every chain's first launch (the prefix of every y2 row, both buffers)
    for (int i = 0; i < 2 && h->d_y2[i]; i++)
        HIP_TRY(hipMemset2DAsync(h->d_y2[i], h->y2_pitch * sizeof(double2), 0, (size_t)NVX_Y2_PREFIX * sizeof(double2), (size_t)h->y2_rows, h->stream));
    for (int i = 0; i < 2; i++) HIP_TRY(hipMemsetAsync(h->d_dd[i], 0, (size_t)NVX_DEMOD_DOUBLES * h->n_slots * sizeof(double), h->stream));
    // ints: all zero except prev_offset = -1 (decoder.C:30) and the bit-FSM phase = -1 (waiting)
    std::vector<int> ints((size_t)NVX_DEMOD_INTS * h->n_slots, 0);
    for (int i = 0; i < h->n_slots; i++) {
        ints[(size_t)NVX_DI_PREV_OFFSET * h->n_slots + i] = -1;
        ints[(size_t)NVX_DI_PHASE * h->n_slots + i] = -1;
    }
    HIP_TRY(hipMemcpyAsync(h->d_di, ints.data(), ints.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    const nvx_tie_stats ties0 = { 0, 0, 0x7f800000u, 0 };
    *h->h_ties = ties0;
    HIP_TRY(hipMemcpyAsync(h->d_ties, &ties0, sizeof ties0, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto &s : h->slots) { s.bits.clear(); s.base = 0; s.polled = 0; if (s.sitor) nvx_sitor_reset(s.sitor); }
    if (!h->fill.empty()) {
        std::fill(h->fill.begin(), h->fill.end(), (size_t)0);
        std::fill(h->active.begin(), h->active.end(), (uint8_t)1);
        std::fill(h->last_push_ns.begin(), h->last_push_ns.end(), nvx_now_ns());
        // (every copy out of the staging sets has finished: the streams were synchronised above)
        for (int i = 0; i < 2; i++) std::fill(h->set_launch[i].begin(), h->set_launch[i].end(), (uint64_t)0);
    }
    return NVX_OK;
}

// One stream starts anew (header: nvx_stream_reset): what nvx_reset does, for the rows of ONE input stream -- the cascade
// state blocks, the demodulator state, FIR3's history and the channeliser halo of a wideband handle, the character layers,
// the bit history, the staging -- while the other streams keep everything they carry.  The handle's work is taken in
// first (the stream's rows must be out of every kernel's reach); afterwards the streams are on different clocks, so
// launches carry participant lists until they meet again.
extern "C" int nvx_stream_reset(nvx_handle *h, int stream)
{
    if (!h || stream < 0 || stream >= h->n_in) { nvx_set_error("nvx_stream_reset: bad stream"); return NVX_ERR_ARG; }
    std::unique_lock<std::mutex> lk(h->mu);
    if (h->poisoned) return nvx_poisoned_error(h);                      // a failed launch taints every stream: nvx_reset
    StreamClose closing(h, lk, stream, stream + 1);                     // a push call in progress on this stream ends first (nvx_handle.h)
    if (h->poisoned) return nvx_poisoned_error(h);                      // (the wait released the lock)
    StagingQuiesce quiet(h, lk);
    HIP_TRY(hipSetDevice(h->cfg.device));
    { int rc = nvx_collect_locked(h); if (rc != NVX_OK) return rc; }   // bits and messages of everything launched so far are delivered
    if (h->launch_done_valid) HIP_TRY(hipEventSynchronize(h->launch_done));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream2));
    const int per = h->cfg.wideband ? NVX_WB_SUBBANDS : 1;             // decoded streams of this input stream
    const int d0 = per * stream, slot0 = 2 * d0, n_sl = 2 * per;
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipMemsetAsync(h->d_cstate[i] + (size_t)d0 * NVX_CASCADE_STATE_BYTES, 0, (size_t)per * NVX_CASCADE_STATE_BYTES, h->stream));
        HIP_TRY(hipMemsetAsync(h->d_dd[i] + (size_t)slot0 * NVX_DEMOD_DOUBLES, 0, (size_t)n_sl * NVX_DEMOD_DOUBLES * sizeof(double), h->stream));
        if (h->d_whist[i]) HIP_TRY(hipMemsetAsync(h->d_whist[i] + (size_t)stream * 40, 0, 40 * 4, h->stream));
        for (int k = 0; k < n_sl && h->d_y2[i]; k++)
            if (h->y2row[slot0 + k] >= 0)
                HIP_TRY(hipMemsetAsync(h->d_y2[i] + (size_t)h->y2row[slot0 + k] * h->y2_pitch, 0, (size_t)NVX_Y2_PREFIX * sizeof(double2), h->stream));
    }
    // ints: all zero except prev_offset = -1 (decoder.C:30) and the bit-FSM phase = -1 (waiting), as nvx_reset
    std::vector<int> zero(n_sl, 0), minus(n_sl, -1);
    for (int f = 0; f < NVX_DEMOD_INTS; f++) {
        const int *src = (f == NVX_DI_PREV_OFFSET || f == NVX_DI_PHASE) ? minus.data() : zero.data();
        HIP_TRY(hipMemcpyAsync(h->d_di + (size_t)f * h->n_slots + slot0, src, (size_t)n_sl * sizeof(int), hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int k = 0; k < n_sl; k++) {
        Slot &s = h->slots[slot0 + k];
        s.bits.clear(); s.base = 0; s.polled = 0;
        if (s.sitor) nvx_sitor_reset(s.sitor);
    }
    h->g0s[stream] = 0; h->ended[stream] = 0;
    if (!h->fill.empty()) { h->fill[stream] = 0; h->active[stream] = 1; h->last_push_ns[stream] = nvx_now_ns(); }
    if (ArrivalClock *ac = h->arrival[stream]) { std::lock_guard<std::mutex> al(ac->mu); ac->base = UINT64_MAX; }
    bool together = true;
    for (int s = 1; s < h->n_in && together; s++) together = h->parity[s] == h->parity[0] && h->g0s[s] == h->g0s[0];
    h->diverged = !together;
    return NVX_OK;
}

// launch cascade + demod over n_frames frames of [n_streams][pitch] packed IQ
// tail_n3 (end of a stream's input, nvx_finish): per participant, how many of the launch's 900 S/s samples the stream's
// REAL input produces -- the demodulator stops there (the reference's loop consumes exactly the samples it is given and
// stops: receiver/capt_sched.c:509-513); such a stream is ended afterwards.
int nvx_launch_locked(nvx_handle *h, const void *d_iq, size_t pitch, size_t first_sample, int n_frames, hipStream_t st,
                      const int *part, int n_part, const int *tail_n3)
{
    if (h->poisoned) return nvx_poisoned_error(h);
    if (n_frames < 1 || n_frames > h->cfg.max_frames) { nvx_set_error("n_frames %d outside 1..max_frames %d", n_frames, h->cfg.max_frames); return NVX_ERR_ARG; }
    if ((pitch & 3) || (first_sample & 3)) { nvx_set_error("pitch and first sample must be multiples of 4 samples"); return NVX_ERR_ARG; }
    if (part && (n_part < 1 || n_part > h->n_in)) { nvx_set_error("launch with %d of %d streams", n_part, h->n_in); return NVX_ERR_ARG; }
    if (part && n_part == h->n_in && !tail_n3) part = nullptr;      // ascending and distinct: that is every stream
    // (validated before anything is enqueued or committed; `diverged` and the statistics move only behind the last enqueue)
    for (int i = 0; part && i < n_part; i++)
        if (part[i] < 0 || part[i] >= h->n_in || (i > 0 && part[i] <= part[i - 1])) { nvx_set_error("launch list: stream %d out of order or range", part[i]); return NVX_ERR_ARG; }
    const int n_named = part ? n_part : h->n_in;
    for (int i = 0; i < n_named; i++) {
        const int s = part ? part[i] : i;
        if (h->ended[s]) { nvx_set_error("stream %d has ended (nvx_finish): nvx_stream_reset or nvx_reset starts a new one", s); return NVX_ERR_STATE; }
        if (tail_n3 && (tail_n3[i] < 1 || tail_n3[i] >= n_frames * NVX_FRAME_Y3)) { nvx_set_error("launch list: %d samples at 900 S/s in the tail of stream %d", tail_n3[i], s); return NVX_ERR_ARG; }
    }
    Result &r = h->res[h->launched % RESULT_SLOTS];
    // Ring full: take in the OLDEST result only (launched RESULT_SLOTS launches ago, long finished), so that the launches
    // behind it keep the GPU busy while the host appends its bits.  (Collecting everything here drained the pipeline every
    // RESULT_SLOTS launches: a demodulator running alone plus the host's character layer, 0.5-1 ms per step.)
    if (r.pending) { int rc = nvx_collect_locked(h, h->launched - RESULT_SLOTS + 1); if (rc != NVX_OK) return rc; }
    // ... and whatever else has finished meanwhile (no waiting): its bits and messages reach the user now, behind the
    // launches that are still queued on the GPU, instead of with the next fetch
    { int rc = nvx_collect_ready_locked(h); if (rc != NVX_OK) return rc; }

    // a launch on another stream than its predecessor: order it behind the predecessor's last operation
    if (h->launch_done_valid && st != h->last_launch_stream) HIP_TRY(hipStreamWaitEvent(st, h->launch_done, 0));

    const bool fused = h->cfg.wideband != 0;           // wideband handles run nvx_wideband_fused + nvx_fir3 (nvx_wideband_fused.hip)
    const int yb = (int)(h->launched & 1);              // y3 buffer of this launch
    // Where the demodulator runs (measured, profiles/TUNING.md): the whole demodulator of launch k -- nvx_fir3 of a
    // wideband handle, front, FSM, bit download -- runs on the second stream, beside the cascade of launch k + 1, whose
    // persistent grid is one wave per CU short of what fits so that a workgroup of the front (14 KB of LDS) finds room on
    // every CU: step 20.9 -> 20.5 ms on the headline workload.
    hipStream_t sd = h->stream2;
    // Who takes part.  While every launch has covered every stream, all streams share one state-block parity and one
    // sample count and the kernels need no list.  From the first partial launch on (a stream of a push-mode handle had
    // no frame) the streams are on their own clocks: the launch carries a list with each participant's parity and g0.
    // A launch that ends streams carries one too (their true sample counts).
    const int per_part = h->cfg.wideband ? NVX_WB_SUBBANDS : 1;          // decoded streams per input stream
    const int n3_full = n_frames * NVX_FRAME_Y3;
    const bool with_list = h->diverged || part != nullptr || tail_n3 != nullptr;
    r.n_part = 0;
    const nvx_part *d_list = nullptr;
    if (with_list) {
        r.n_part = n_named;
        for (int i = 0; i < r.n_part; i++) {
            const int s = part ? part[i] : i;
            r.h_part[i] = nvx_part{ s, (int)h->parity[s], h->g0s[s], tail_n3 ? tail_n3[i] : n3_full, 0 };
        }
        // (the slot's previous launch has been collected above, so neither copy of its list is still in use)
        HIP_TRY(hipMemcpyAsync(r.d_part, r.h_part, (size_t)r.n_part * sizeof(nvx_part), hipMemcpyHostToDevice, st));
        d_list = r.d_part;
    }
    const int n_here = with_list ? r.n_part : h->n_in;                   // input streams in this launch
    // without a list every stream has the same parity: the kernels then read state[0] and write state[1]
    const int p0 = d_list ? 0 : (int)h->parity[0];
    const unsigned third0 = (unsigned)(h->g0s[0] / (NVX_FRAME_Y3 / 3));  // the state blocks' tag (nvx_kernels.h): position in thirds of a frame
    // FIR2 output buffers (fused wideband kernel): with a list [parity of the stream]; without one the buffer to write as [0]
    double2 *const y2_bufs[2] = { h->d_y2[d_list ? 0 : p0], h->d_y2[d_list ? 1 : p0 ^ 1] };
    nvx_demod_args da{};
    da.y3 = h->d_y3[yb]; da.y3_cap = (size_t)h->y3_cap; da.y3_base = 0; da.n3 = n3_full;
    da.n_slots = h->n_slots; da.slot_active = h->d_active;
    da.g0 = h->g0s[0]; da.part = d_list; da.n_part = r.n_part; da.per_part = per_part;
    da.dstate[0] = h->d_dd[p0]; da.dstate[1] = h->d_dd[p0 ^ 1];     // (read, write) without a list; [0], [1] with one
    da.state_i = h->d_di; da.fsm_table = h->d_fsm_tab; da.words = h->d_words;
    da.bits = r.d_bits; da.bits_cap = h->bits_cap; da.nbits = r.d_nbits; da.dphi = h->d_dphi; da.ties = h->d_ties;

    // cascade on `st`: it may not overwrite y3[yb] before the demodulator of two launches ago has read it
    if (h->demod_pending[yb]) HIP_TRY(hipStreamWaitEvent(st, h->demod_done[yb], 0));
    r.timed = h->timing;
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[0], st));
    if (fused) {
        nvx_wideband_args wa{};
        wa.raw = (const uint32_t *)d_iq; wa.pitch = pitch; wa.first_sample = first_sample;
        wa.n_wide = n_here; wa.n_frames = n_frames; wa.chain_masks = h->d_masks;
        wa.part = d_list;
        wa.state[0] = h->d_cstate[p0]; wa.state[1] = h->d_cstate[p0 ^ 1];
        wa.hist[0] = h->d_whist[p0]; wa.hist[1] = h->d_whist[p0 ^ 1];       // a stream reads [its parity], writes the other
        wa.y3 = h->d_y3[yb]; wa.y3_cap = (size_t)h->y3_cap; wa.y3_base = 0;
        wa.queue = h->d_ctrl; wa.status = h->d_ctrl + 1; wa.done = h->d_ctrl + NVX_CASCADE_CTRL_INTS; wa.third0 = third0;
        wa.y2[0] = y2_bufs[0]; wa.y2[1] = y2_bufs[1]; wa.y2_pitch = h->y2_pitch; wa.y2_row = h->d_y2row;
        HIP_TRY(nvx_launch_wideband_fused(&wa, st));
    } else {
        nvx_cascade_args ca{};
        ca.iq = (const uint32_t *)d_iq; ca.pitch = pitch; ca.first_sample = first_sample;
        ca.n_frames = n_frames; ca.n_streams = n_here; ca.chain_masks = h->d_masks;
        ca.part = d_list;
        ca.state[0] = h->d_cstate[p0]; ca.state[1] = h->d_cstate[p0 ^ 1]; ca.y3 = h->d_y3[yb]; ca.y3_cap = (size_t)h->y3_cap; ca.y3_base = 0;
        ca.queue = h->d_ctrl; ca.status = h->d_ctrl + 1; ca.done = h->d_ctrl + NVX_CASCADE_CTRL_INTS;
        ca.stage0_order = h->cfg.stage0_order;
        ca.third0 = third0;
        ca.max_waves_per_cu = -1;                        // one fewer than fit: room for the previous launch's demodulator (above)
        HIP_TRY(nvx_launch_cascade(&ca, h->cascade_raw, h->nch, st));
    }
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[1], st));
    HIP_TRY(hipMemcpyAsync(h->h_status + NVX_STATUS_INTS * (h->launched % RESULT_SLOTS), h->d_ctrl + 1, NVX_STATUS_INTS * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(h->casc_done[yb], st));
    // the demodulator behind the cascade, on its own stream (which also orders it behind the previous launch's FSM, whose
    // word buffer the front reuses)
    HIP_TRY(hipStreamWaitEvent(sd, h->casc_done[yb], 0));
    if (fused) {
        // FIR3 (the fused wideband kernel's waves end at FIR2): y2 rows -> y3[yb], in front of the demodulator on its stream
        nvx_fir3_args fa{};
        fa.y2[0] = y2_bufs[0]; fa.y2[1] = y2_bufs[1]; fa.y2_pitch = h->y2_pitch; fa.y2_row = h->d_y2row;
        fa.y3 = h->d_y3[yb]; fa.y3_cap = (size_t)h->y3_cap; fa.y3_base = 0; fa.n_frames = n_frames; fa.n_slots = h->n_slots;
        fa.part = d_list; fa.n_part = r.n_part; fa.per_part = per_part;
        if (r.timed) HIP_TRY(hipEventRecord(r.ev[6], sd));
        HIP_TRY(nvx_launch_fir3(&fa, sd));
        if (r.timed) HIP_TRY(hipEventRecord(r.ev[7], sd));
    }
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[2], sd));
    HIP_TRY(nvx_launch_demod_front(&da, sd));
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[3], sd));
    HIP_TRY(hipEventRecord(h->demod_done[yb], sd));      // y3[yb] consumed
    h->demod_pending[yb] = true;
    // FSM + bit download behind the front
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[4], sd));
    HIP_TRY(nvx_launch_demod_fsm(&da, sd));
    if (r.timed) HIP_TRY(hipEventRecord(r.ev[5], sd));
    HIP_TRY(hipMemcpyAsync(h->h_ties, h->d_ties, sizeof(nvx_tie_stats), hipMemcpyDeviceToHost, sd));
    HIP_TRY(hipMemcpyAsync(r.h_nbits, r.d_nbits, (size_t)h->n_slots * sizeof(int), hipMemcpyDeviceToHost, sd));
    HIP_TRY(hipMemcpyAsync(r.h_bits, r.d_bits, (size_t)h->n_slots * h->bits_cap, hipMemcpyDeviceToHost, sd));
    HIP_TRY(hipEventRecord(r.done, sd));
    HIP_TRY(hipEventRecord(h->launch_done, sd));                    // sd has waited for st's last operation
    h->launch_done_valid = true; h->last_launch_stream = st;
    r.pending = true;
    r.n3 = n3_full;
    r.g0_all = h->g0s[0];
    h->launched++;
    h->last_n3 = n3_full;
    for (int i = 0; i < n_here; i++) {                   // the participants have moved on: other block, n3 more samples
        const int s = with_list ? r.h_part[i].stream : i;
        h->parity[s] ^= 1; h->g0s[s] += (unsigned long long)(tail_n3 ? tail_n3[i] : n3_full);
        if (tail_n3) h->ended[s] = 1;                    // its filters have run into whatever lay behind its last sample
    }
    if (part) h->partial_launches++;
    // from the first partial launch on the streams are on their own clocks (every launch carries a list) -- until they
    // have come together again: same parity, same position
    h->diverged = with_list;
    if (h->diverged) {
        bool together = true;
        for (int s = 1; s < h->n_in && together; s++) together = h->parity[s] == h->parity[0] && h->g0s[s] == h->g0s[0];
        if (together) h->diverged = false;
    }
    return NVX_OK;
}

// take in every launched block that has already finished (hipEventQuery, never waits), oldest first
int nvx_collect_ready_locked(nvx_handle *h)
{
    while (h->collected < h->launched) {
        Result &o = h->res[h->collected % RESULT_SLOTS];
        if (o.pending) {
            const hipError_t q = hipEventQuery(o.done);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); break; }
            if (q != hipSuccess) { nvx_set_error("hipEventQuery: %s", hipGetErrorString(q)); return NVX_ERR_HIP; }
        }
        int rc = nvx_collect_locked(h, h->collected + 1);
        if (rc != NVX_OK) return rc;
    }
    return NVX_OK;
}

int nvx_poisoned_error(nvx_handle *h)
{
    nvx_set_error("a launch of this handle failed (%s): nvx_reset it", h->poison_why.c_str());
    return NVX_ERR_STATE;
}

int nvx_launches_in_flight(nvx_handle *h)
{
    std::lock_guard<std::mutex> lk(h->mu);
    return (int)(h->launched - h->collected);
}

uint64_t nvx_launch_count(nvx_handle *h)
{
    std::lock_guard<std::mutex> lk(h->mu);
    return h->launched;
}

extern "C" int nvx_poll(nvx_handle *h)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->poisoned) return nvx_poisoned_error(h);
    if (h->collected == h->launched) return NVX_OK;          // nothing in flight: no HIP call at all
    HIP_TRY(hipSetDevice(h->cfg.device));
    return nvx_collect_ready_locked(h);
}

// wait for the launched blocks in front of `upto` (default: all of them), append bits, run the character layer
int nvx_collect_locked(nvx_handle *h, uint64_t upto)
{
    if (upto > h->launched) upto = h->launched;
    while (h->collected < upto) {
        Result &r = h->res[h->collected % RESULT_SLOTS];
        if (r.pending) {
            HIP_TRY(hipEventSynchronize(r.done));
            const int *stat = h->h_status + NVX_STATUS_INTS * (h->collected % RESULT_SLOTS);
            h->wait_polls += (uint64_t)(unsigned)stat[1]; h->wait_units += (uint64_t)(unsigned)stat[2]; h->wait_launches++;
            h->stale_repaired += (uint64_t)(unsigned)stat[3];
            if (stat[0] != 0) {
                // The launch's bits are not taken in: whatever it produced rests on a state nobody vouches for.  Nor can
                // anything behind it be trusted: the unit that found the bad block ran on and sealed what it computed from
                // it, the demodulator state moved on too, and the launches already queued inherit both under valid seals.
                // So the failure sticks (nvx_handle.h, poisoned): every pending result is dropped here, and the handle
                // answers NVX_ERR_STATE until nvx_reset.
                if (stat[0] == NVX_STATUS_INTEGRITY) {
                    h->integrity_failures++;
                    h->poison_why = "the filter state a launch inherited from its predecessor failed its integrity word";
                } else {
                    h->poison_why = "a wait on the previous frame of a stream timed out in the FIR cascade's work queue";
                }
                h->poisoned = true;
                // (launch_done covers every queued launch: nothing of them is in flight when their slots are released)
                if (h->launch_done_valid) (void)hipEventSynchronize(h->launch_done);
                for (auto &q : h->res) q.pending = false;
                h->collected = h->launched;
                nvx_set_error("FIR cascade: %s; the results of this launch and of the launches queued behind it are discarded (nvx_reset the handle)", h->poison_why.c_str());
                return NVX_ERR_HIP;
            }
            if (r.timed) {
                HIP_TRY(hipEventElapsedTime(&h->ms[0], r.ev[0], r.ev[1]));
                float fsm_ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&h->ms[1], r.ev[2], r.ev[3]));
                HIP_TRY(hipEventElapsedTime(&fsm_ms, r.ev[4], r.ev[5]));
                h->ms[1] += fsm_ms;                       // "demodulator" = front + FSM
                h->ms[2] = 0.f;
                if (h->d_y2[0]) HIP_TRY(hipEventElapsedTime(&h->ms[2], r.ev[6], r.ev[7]));
                h->ms_sum[0] += h->ms[0]; h->ms_sum[1] += h->ms[1]; h->ms_sum[2] += h->ms[2]; h->ms_count++;
            }
            std::atomic<int> bad_slot{ -1 };
            // the chains of this launch: every slot, or (a launch with a participant list) the 2 * per_part slots of each
            // participating input stream -- the other slots' rows of this result hold nothing from this launch
            const int per_slots = 2 * (h->cfg.wideband ? NVX_WB_SUBBANDS : 1);
            const int n_chains = r.n_part ? r.n_part * per_slots : h->n_slots;
            auto work = [&](int lo, int hi) {
                for (int k = lo; k < hi; k++) {
                    const int i = r.n_part ? r.h_part[k / per_slots].stream * per_slots + k % per_slots : k;
                    Slot &s = h->slots[i];
                    if (!s.active) continue;
                    int n = r.h_nbits[i];
                    if (n < 0 || n > h->bits_cap * 8) { bad_slot = i; continue; }
                    const uint32_t *pw = (const uint32_t *)(r.h_bits + (size_t)i * h->bits_cap);
                    const size_t at = s.bits.size();
                    s.bits.resize(at + (size_t)n);
                    for (int b = 0; b < n; b++) s.bits[at + b] = ((pw[b >> 5] >> (b & 31)) & 1u) ? 'B' : 'Y';
                    if (s.sitor) nvx_sitor_receive_bits(s.sitor, s.bits.data() + at, (size_t)n);
                    if (s.bits.size() > 2 * h->bit_history) {                // a receiver runs for weeks: bound the poll history
                        const size_t drop = s.bits.size() - h->bit_history;
                        s.bits.erase(0, drop);
                        s.base += drop;
                    }
                }
            };
            // cfg.host_threads, else NVX_HOST_THREADS, else the hardware's count; never more than 16 (7 ns per bit: more
            // threads only add start-up cost).  A group (nvx_group.cpp) gives every member its share of the cores.
            static const int env_threads = [] {
                const char *e = getenv("NVX_HOST_THREADS");
                int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
                return n < 1 ? 1 : (n > 16 ? 16 : n);
            }();
            const int host_threads = h->cfg.host_threads > 0 ? std::min(h->cfg.host_threads, 16) : env_threads;
            // worth spreading out when the launch carried enough bit periods: 7 ns a bit adds up to a millisecond of the
            // launching thread's time at 64 streams x 62 frames (measured: 3.36 ms per step against 2.10 without the layer)
            const long long periods = (long long)n_chains * (r.n3 / 9);
            const int nt = (h->cfg.char_layer && n_chains >= 2 && periods >= 16384) ? std::min(host_threads, n_chains) : 1;
            if (nt > 1) {
                const int per = (n_chains + nt - 1) / nt;
                h->pool.run(nt, [&](int t) { work(std::min(n_chains, t * per), std::min(n_chains, (t + 1) * per)); });
            } else {
                work(0, n_chains);
            }
            if (bad_slot >= 0) {
                // A bit count outside what a launch can produce (it never has been: a bit takes at least eight samples): the
                // result is not to be trusted and the other slots' bits of this launch are already appended -- taking the
                // launch in AGAIN would append them twice.  So this sticks like a failed launch does.
                h->poison_why = "a chain's bit count lay outside what a launch can produce (slot " + std::to_string(bad_slot.load()) + ")";
                h->poisoned = true;
                if (h->launch_done_valid) (void)hipEventSynchronize(h->launch_done);
                for (auto &q : h->res) q.pending = false;
                h->collected = h->launched;
                nvx_set_error("%s; the results of this launch and of the launches queued behind it are discarded (nvx_reset the handle)", h->poison_why.c_str());
                return NVX_ERR_HIP;
            }
            for (int i = 0; i < h->n_slots; i++) if (!h->slots[i].outbox.empty()) deliver_outbox(h, i / 2, h->slots[i]);
            if (h->n_arrival) {
                // live path: the frames of this launch are now decoded, pollable and their messages delivered
                const int64_t now = nvx_now_ns();
                const int n_frames = r.n3 / NVX_FRAME_Y3, n_str = r.n_part ? r.n_part : h->n_in;
                for (int k = 0; k < n_str; k++) {
                    const int s = r.n_part ? r.h_part[k].stream : k;
                    ArrivalClock *ac = h->arrival[s];
                    if (!ac) continue;
                    const uint64_t f0 = (r.n_part ? r.h_part[k].g0 : r.g0_all) / NVX_FRAME_Y3;
                    for (int f = 0; f < n_frames; f++) ac->book(f0 + (uint64_t)f, now);
                }
            }
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

extern "C" int nvx_set_trace(nvx_handle *h, nvx_sitor_trace_fn fn, void *user)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    for (auto &s : h->slots) if (s.sitor) nvx_sitor_set_trace(s.sitor, fn, user);
    return NVX_OK;
}

extern "C" int nvx_process_resident(nvx_handle *h, const void *d_iq, size_t pitch, size_t first_frame, int n_frames, void *hip_stream)
{
    if (!h || !d_iq) { nvx_set_error("nvx_process_resident: null argument"); return NVX_ERR_ARG; }
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (n_frames >= 1 && n_frames <= h->cfg.max_frames) {            // (nvx_launch_locked names the other argument errors)
        // the last stream's last frame must still lie inside the caller's buffer
        const size_t span = ((size_t)(h->n_in - 1) * pitch + (first_frame + (size_t)n_frames) * h->frame_in) * 4;
        int rc = nvx_check_device_span(d_iq, span, "nvx_process_resident: [n_streams][pitch] input");
        if (rc != NVX_OK) return rc;
    }
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
    return nvx_launch_locked(h, d_iq, pitch, first_frame * h->frame_in, n_frames, st);
}

extern "C" int nvx_fetch_bits(nvx_handle *h)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->poisoned) return nvx_poisoned_error(h);
    HIP_TRY(hipSetDevice(h->cfg.device));
    return nvx_collect_locked(h);
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
extern "C" float nvx_last_kernel_ms(nvx_handle *h, int which) { return (h && which >= 0 && which < 3) ? h->ms[which] : -1.f; }
extern "C" int nvx_kernel_time_stats(nvx_handle *h, int which, double *sum_ms, uint64_t *launches, int reset)
{
    if (!h || which < 0 || which > 2) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (sum_ms) *sum_ms = h->ms_sum[which];
    if (launches) *launches = h->ms_count;
    if (reset) { h->ms_sum[0] = h->ms_sum[1] = h->ms_sum[2] = 0.0; h->ms_count = 0; }
    return NVX_OK;
}

extern "C" int nvx_cascade_wait_stats(nvx_handle *h, uint64_t *polls, uint64_t *units_waited, uint64_t *launches, int reset)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (polls) *polls = h->wait_polls;
    if (units_waited) *units_waited = h->wait_units;
    if (launches) *launches = h->wait_launches;
    if (reset) h->wait_polls = h->wait_units = h->wait_launches = 0;
    return NVX_OK;
}

extern "C" int nvx_cascade_integrity_stats(nvx_handle *h, uint64_t *stale_repaired, uint64_t *launch_failures, uint64_t *launches, int reset)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (stale_repaired) *stale_repaired = h->stale_repaired;
    if (launch_failures) *launch_failures = h->integrity_failures;
    if (launches) *launches = h->wait_launches;
    if (reset) h->stale_repaired = h->integrity_failures = 0;
    return NVX_OK;
}

extern "C" int nvx_demod_tie_stats(nvx_handle *h, uint64_t *near_ties, uint64_t *evaluations, double *min_margin)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->launch_done_valid) HIP_TRY(hipEventSynchronize(h->launch_done));     // the pinned copy is refreshed behind every launch
    if (near_ties) *near_ties = h->h_ties->near_ties;
    if (evaluations) *evaluations = h->h_ties->evaluations;
    if (min_margin) {
        float f; memcpy(&f, &h->h_ties->min_margin_bits, sizeof f);
        *min_margin = h->h_ties->evaluations ? (double)f : -1.0;
    }
    return NVX_OK;
}

extern "C" int nvx_enable_debug(nvx_handle *h, int enabled)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->launch_done_valid) HIP_TRY(hipEventSynchronize(h->launch_done));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (enabled && !h->d_dphi) HIP_TRY(hipMalloc(&h->d_dphi, (size_t)h->n_slots * h->y3_cap * sizeof(double)));
    if (!enabled && h->d_dphi) { hipFree(h->d_dphi); h->d_dphi = nullptr; }
    return NVX_OK;
}

static_assert(NVX_STATE_BLOCK_BYTES == NVX_CASCADE_STATE_BYTES, "public size of a state block");
extern "C" int nvx_debug_cascade_state(nvx_handle *h, int stream, void *buf, size_t bytes, int write)
{
    if (!h || !buf || stream < 0 || stream >= h->n_streams || bytes != (size_t)NVX_CASCADE_STATE_BYTES) { nvx_set_error("nvx_debug_cascade_state: bad argument (a block is %d bytes)", NVX_CASCADE_STATE_BYTES); return NVX_ERR_ARG; }
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->launch_done_valid) HIP_TRY(hipEventSynchronize(h->launch_done));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int in = h->cfg.wideband ? stream / NVX_WB_SUBBANDS : stream;      // the input stream whose parity the block follows
    uint8_t *blk = h->d_cstate[h->parity[in]] + (size_t)stream * NVX_CASCADE_STATE_BYTES;
    if (write) HIP_TRY(hipMemcpy(blk, buf, bytes, hipMemcpyHostToDevice));
    else HIP_TRY(hipMemcpy(buf, blk, bytes, hipMemcpyDeviceToHost));
    return NVX_OK;
}

// host copy of seal_tag (nvx_cascade_wave.h): the tag a state block's seal is mixed with.  The kernels judge what this
// writes (tests/test_gpu_deviations.py): a disagreement is a failed launch, not a silent one.
static unsigned long long seal_tag_host(int stream, unsigned third)
{
    unsigned long long t = ((unsigned long long)(unsigned)stream << 32) | third;
    t *= 0x9E3779B97F4A7C15ull; t ^= t >> 29; t *= 0xBF58476D1CE4E5B9ull; t ^= t >> 32;
    return t;
}

// header: nvx_debug_advance_clock.  Everything the kernels derive from a stream's sample clock g is periodic in
// NVX_CLOCK_PERIOD = lcm(288, 9 * 567): the frame phase (g mod 288), the bit-timing filter's ring position and class
// (g mod 5103; nvx_demod.hip), the priming thresholds (g >= 8 / 574 / 582: once passed, passed) -- except the position tag
// in the seal of the carried FIR state (g / 96), which is re-written here for the block the stream's next launch reads.
static_assert(NVX_CLOCK_PERIOD % NVX_FRAME_Y3 == 0 && NVX_CLOCK_PERIOD % (9 * 567) == 0, "the clock's period");
extern "C" int nvx_debug_advance_clock(nvx_handle *h, int stream, uint64_t periods)
{
    if (!h || stream < 0 || stream >= h->n_in) { nvx_set_error("nvx_debug_advance_clock: bad stream"); return NVX_ERR_ARG; }
    if (h->cfg.wideband) { nvx_set_error("nvx_debug_advance_clock: not for wideband handles"); return NVX_ERR_STATE; }
    std::unique_lock<std::mutex> lk(h->mu);
    if (h->poisoned) return nvx_poisoned_error(h);
    if (h->ended[stream]) { nvx_set_error("nvx_debug_advance_clock: stream %d has ended", stream); return NVX_ERR_STATE; }
    HIP_TRY(hipSetDevice(h->cfg.device));
    { int rc = nvx_collect_locked(h); if (rc != NVX_OK) return rc; }
    if (h->launch_done_valid) HIP_TRY(hipEventSynchronize(h->launch_done));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream2));
    const unsigned long long g_old = h->g0s[stream];
    // (the priming thresholds are the one thing that is NOT periodic: a stream that has not passed them would skip its priming)
    if (g_old < 3 * NVX_FRAME_Y3) { nvx_set_error("nvx_debug_advance_clock: stream %d is still priming (%llu samples at 900 S/s; the timing filter is primed at 582)", stream, g_old); return NVX_ERR_STATE; }
    if (periods > (~0ull - g_old) / (unsigned long long)NVX_CLOCK_PERIOD) { nvx_set_error("nvx_debug_advance_clock: the clock would pass 2^64"); return NVX_ERR_ARG; }
    const unsigned long long g_new = g_old + periods * (unsigned long long)NVX_CLOCK_PERIOD;
    const unsigned third_old = (unsigned)(g_old / (NVX_FRAME_Y3 / 3)), third_new = (unsigned)(g_new / (NVX_FRAME_Y3 / 3));
    unsigned long long seal[2];
    uint8_t *entry = h->d_cstate[h->parity[stream]] + (size_t)stream * NVX_CASCADE_STATE_BYTES + (size_t)NVX_STATE_SEAL * 16;
    HIP_TRY(hipMemcpy(seal, entry, sizeof seal, hipMemcpyDeviceToHost));
    seal[0] ^= seal_tag_host(stream, third_old) ^ seal_tag_host(stream, third_new);
    seal[1] = ((unsigned long long)(unsigned)stream << 32) | third_new;
    HIP_TRY(hipMemcpy(entry, seal, sizeof seal, hipMemcpyHostToDevice));
    h->g0s[stream] = g_new;
    bool together = true;
    for (int s = 1; s < h->n_in && together; s++) together = h->parity[s] == h->parity[0] && h->g0s[s] == h->g0s[0];
    h->diverged = !together;
    return NVX_OK;
}

extern "C" size_t nvx_debug_y3(nvx_handle *h, int stream, int chain, double *out, size_t cap_pairs)
{
    if (!h || !out || stream < 0 || stream >= h->n_streams || chain < 0 || chain > 1) return 0;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->cfg.device);
    hipDeviceSynchronize();
    size_t n = std::min(cap_pairs, (size_t)h->last_n3);
    // a stream whose input has ended (nvx_finish): only the samples its real input produced (the rest of the frame was
    // computed from the zeros behind its last sample and is not part of anything)
    const int in = h->cfg.wideband ? stream / NVX_WB_SUBBANDS : stream;
    if (h->ended[in]) n = std::min(n, (size_t)(h->g0s[in] % NVX_FRAME_Y3));
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

// ------------------------------------------------------------ device helpers
extern "C" int nvx_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
extern "C" void *nvx_device_alloc(int device, size_t bytes)
{
    if (nvx_select_device(device) != NVX_OK) return nullptr;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { nvx_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
extern "C" void nvx_device_free(int device, void *p) { if (p && nvx_select_device(device) == NVX_OK) hipFree(p); }
extern "C" int nvx_memcpy_h2d(int device, void *d, const void *s, size_t n)
{
    int rc = nvx_select_device(device); if (rc != NVX_OK) return rc;
    HIP_TRY(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); return NVX_OK;
}
extern "C" int nvx_memcpy_d2h(int device, void *d, const void *s, size_t n)
{
    int rc = nvx_select_device(device); if (rc != NVX_OK) return rc;
    HIP_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToHost)); return NVX_OK;
}
extern "C" int nvx_device_sync(int device)
{
    int rc = nvx_select_device(device); if (rc != NVX_OK) return rc;
    HIP_TRY(hipDeviceSynchronize()); return NVX_OK;
}
extern "C" void *nvx_stream_create(int device)
{
    if (nvx_select_device(device) != NVX_OK) return nullptr;
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { nvx_set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); return nullptr; }
    return (void *)s;
}
extern "C" void nvx_stream_destroy(int device, void *hip_stream)
{
    if (hip_stream && nvx_select_device(device) == NVX_OK) { hipStreamSynchronize((hipStream_t)hip_stream); hipStreamDestroy((hipStream_t)hip_stream); }
}
