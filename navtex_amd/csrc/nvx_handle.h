// nvx_handle.h -- what the translation units of the host runtime share: the handle and its
// helpers.  Internal; the public ABI is include/navtex_amd.h.
#ifndef NVX_HANDLE_H
#define NVX_HANDLE_H

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
#include "nvx_pool.h"

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            nvx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? NVX_ERR_NODEV : NVX_ERR_HIP; \
        }                                                                                  \
    } while (0)

// NVX_OK after hipSetDevice(device); NVX_ERR_NODEV ("no CPU path") without a HIP device
int nvx_select_device(int device);
// A kernel that reads or writes outside its operands faults, and a GPU fault can take the whole node with it: before a
// launch over CALLER memory, [p, p + bytes) is held against the allocation the runtime knows p to lie in
// (hipMemGetAddressRange; the current device must be the pointer's).  NVX_ERR_ARG when the span leaves the allocation;
// NVX_OK when it fits -- or when the runtime does not know the pointer (no verdict possible: a suballocator's arena
// gives a weaker check, memory of another kind none).
int nvx_check_device_span(const void *p, size_t bytes, const char *what);

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
    // the input streams that took part in the launch (nvx_kernels.h, nvx_part): pinned host copy, device copy, and
    // how many -- 0 = every stream (no list); the collect reads bits of the participants' chains only
    nvx_part *h_part = nullptr, *d_part = nullptr;
    int n_part = 0;
    int n3 = 0;                        // 900 S/s samples per chain in the launch
    unsigned long long g0_all = 0;     // no list: the 900 S/s sample count every stream had when the launch went out
    hipEvent_t copied = nullptr;       // push mode: the launch's host-to-device copies have left the staging sets
    hipEvent_t done = nullptr;
    hipEvent_t ev[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };   // begin/end of cascade, demod front, demod FSM, nvx_fir3
    bool timed = false;
    bool pending = false;
};

// Decode latency of the live path (r4).  The reference decodes synchronously, sample by sample, and calls add_message
// inline (receiver/capt_sched.c:484-528, receiver/nav_b_sm.C:87); here a frame waits for its last sample, a launch and a
// collect.  A capture ring stamps the moment the vendor callback that carried a frame's LAST sample was entered; when
// the launch that covers the frame has been collected -- bits pollable, messages delivered -- the handle books
// (collect time - arrival) for the frame.  Owned by the capture ring, registered with the handle per input stream.
struct ArrivalClock {
    static const int RING = 512;             // frames between callback and collect (a ring of seconds holds a few dozen)
    static const int KEEP = 8192;            // latencies kept for the percentiles (a receiver runs for weeks)
    std::mutex mu;
    int64_t t_ns[RING];
    uint64_t stamped = 0;                    // capture frames stamped so far: frame f of the capture, f < stamped
    uint64_t base = 0;                       // the handle's frame count of the stream when the capture started
    std::vector<float> lat_ms;               // the last KEEP booked latencies (ring once full)
    uint64_t booked = 0;
    float max_ms = 0.f, last_ms = 0.f;
    void stamp(uint64_t frame, int64_t ns)
    {
        std::lock_guard<std::mutex> lk(mu);
        t_ns[frame % RING] = ns;
        if (frame + 1 > stamped) stamped = frame + 1;
    }
    void book(uint64_t handle_frame, int64_t now_ns)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (handle_frame < base) return;
        const uint64_t f = handle_frame - base;
        if (f >= stamped || stamped - f > RING) return;       // not one of this capture's frames, or stamped too long ago
        const float ms = (float)((double)(now_ns - t_ns[f % RING]) * 1e-6);
        if (lat_ms.size() < (size_t)KEEP) lat_ms.push_back(ms); else lat_ms[booked % KEEP] = ms;
        booked++; last_ms = ms; if (ms > max_ms) max_ms = ms;
    }
};
int64_t nvx_now_ns();
// what nvx_capture_latency / nvx_shim_latency report from a clock (any out pointer may be null)
void nvx_clock_report(ArrivalClock &c, uint64_t *frames, double *p50_ms, double *p99_ms, double *max_ms, double *last_ms, int reset);

struct nvx_handle {
    nvx_config cfg{};
    int n_streams = 0, n_slots = 0, nch = 1;   // n_streams: 252 kS/s-path streams (8 per input in wideband mode)
    int n_in = 0;                      // input streams the caller addresses (= n_streams unless wideband)
    size_t bit_history = NVX_BIT_HISTORY;
    bool cascade_raw = false;          // the cascade kernel's RAW switch (never set in wideband mode)
    size_t frame_in = 0;               // complex input samples per frame at the input rate
    uint32_t *d_whist[2] = { nullptr, nullptr };   // wideband handles: the channeliser's 40-sample halo in front of a launch, by stream parity
    int y3_cap = 0, bits_cap = 0;
    hipStream_t stream = nullptr;      // FIR cascade (or the caller's stream) and H2D staging
    hipStream_t stream2 = nullptr;     // the demodulator (nvx_fir3 of a wideband handle, front, FSM) + D2H of the bits: beside the next cascade launch
    hipEvent_t casc_done[2] = { nullptr, nullptr };   // y3[b] written
    hipEvent_t demod_done[2] = { nullptr, nullptr };  // y3[b] consumed
    // Everything a launch carries (work queue, cascade state, demodulator state, word buffer) is ordered by stream
    // order on the launch stream.  launch_done is recorded behind the last operation of every launch; a launch on a
    // DIFFERENT stream than its predecessor waits for it, and reset / enable_debug / destroy wait for it on the host,
    // so a caller may pass any stream to nvx_process_resident at any time.
    hipEvent_t launch_done = nullptr; bool launch_done_valid = false;
    hipStream_t last_launch_stream = nullptr;
    bool demod_pending[2] = { false, false };
    // device
    uint8_t *d_masks = nullptr, *d_active = nullptr;
    uint8_t *d_cstate[2] = { nullptr, nullptr };   // cascade state blocks: launch k reads [k & 1], writes [(k + 1) & 1]
    double2 *d_y3[2] = { nullptr, nullptr };   // double buffer between the two streams
    // wideband handles only (the fused wideband kernel's waves end at FIR2, nvx_kernels.h): FIR2 output rows, one per
    // active chain, two buffers by stream parity; the row table
    double2 *d_y2[2] = { nullptr, nullptr };
    int *d_y2row = nullptr;
    std::vector<int> y2row;                    // host copy of the row table: row of slot i, or -1
    size_t y2_pitch = 0; int y2_rows = 0;
    double *d_dd[2] = { nullptr, nullptr };   // demodulator state blocks: a chain reads [its stream's parity], writes the other
    double *d_dphi = nullptr; int *d_di = nullptr;
    uint32_t *d_fsm_tab = nullptr;     // bit-period transition table of the demodulator FSM (nvx_fsm.h)
    unsigned short *d_words = nullptr;
    nvx_tie_stats *d_ties = nullptr;   // arg-max margin statistics, cumulative since create / reset
    nvx_tie_stats *h_ties = nullptr;   // pinned copy, refreshed behind every launch
    int *d_ctrl = nullptr;             // cascade work queue: counter, status, done[n_streams]
    int *h_status = nullptr;           // pinned copies of {status, wait polls, units that waited, stale hand-overs repaired} per result slot
    uint64_t wait_polls = 0, wait_units = 0, wait_launches = 0;   // accumulated at collect
    uint64_t stale_repaired = 0, integrity_failures = 0;          // state-block seals (nvx_kernels.h): hand-overs repaired by a pre-roll, launches failed
    // Per INPUT stream, advanced by every launch the stream takes part in: which cascade state block it reads next
    // (it writes the other) and how many 900 S/s samples its chains have been through since reset.  As long as every
    // launch covered every stream (`diverged` false) all entries are equal and launches need no participant list.
    std::vector<uint8_t> parity;
    std::vector<unsigned long long> g0s;
    bool diverged = false;
    uint64_t partial_launches = 0;     // launches that covered only some of the streams, since create
    // ended[s]: the stream's input has ended (nvx_finish ran its last, partial frame at its true length): its carried state
    // is no continuation of anything -- pushes and launches are refused until nvx_reset
    std::vector<uint8_t> ended;
    // A launch failed (the state it inherited broke its seal, or a hand-over timed out): what the device carries from
    // there on -- filter and demodulator state computed from the bad block, under fresh valid seals -- is garbage that
    // every later launch would pass on.  Sticky: results of the launches queued behind it are dropped, launches, pushes,
    // polls and fetches return NVX_ERR_STATE, nvx_reset clears it.
    bool poisoned = false;
    std::string poison_why;
    Result res[RESULT_SLOTS];
    std::vector<ArrivalClock *> arrival;   // per input stream: the capture ring's clock, or nullptr (guarded by mu)
    int n_arrival = 0;
    uint64_t launched = 0, collected = 0;
    int last_n3 = 0;
    // timing
    bool timing = false;
    float ms[3] = { 0.f, 0.f, 0.f };     // last collected launch: cascade, demodulator (front + FSM), nvx_fir3
    double ms_sum[3] = { 0.0, 0.0, 0.0 };   // over all collected launches since the last stats reset
    uint64_t ms_count = 0;
    // host
    std::vector<uint8_t> masks;
    std::vector<Slot> slots;
    std::vector<struct SinkCtx *> sinks;   // user pointers handed to the per-slot character layers
    HostPool pool;                         // character-layer workers, started on first use
    std::mutex mu;
    // push mode staging: two pinned sets [n_streams][stage_cap] of packed IQ words.  Every stream fills ITS current set
    // (cur[s]) and flips to the other one when a launch takes frames from it; set_launch[s][k] = 1 + the number of the
    // launch whose host-to-device copy last read set k of stream s (0 = none), copies_synced = the highest launch
    // number + 1 whose copy is known to have finished.  active[s] = 0: the stream has gone silent (capture ring's
    // stall timeout, nvx_stream_set_active) and the lock-step trigger does not wait for it.
    uint32_t *h_stage[2] = { nullptr, nullptr };
    std::vector<uint8_t> cur;
    std::vector<uint64_t> set_launch[2];
    uint64_t copies_synced = 0;
    std::vector<uint8_t> active;
    size_t stage_cap = 0;
    std::vector<size_t> fill;
    uint32_t *d_in = nullptr;
    // Large pushes copy into the pinned staging WITHOUT the handle's lock, so that the capture threads of several radios
    // (or replay threads) fill their streams' staging side by side: one thread's copy rate is ~30-40 GB/s, the PCIe link
    // takes 50.  writing[s]: stream s has such a copy in flight (its fill / cur must not move: one writer per stream);
    // writers: how many in all.  Whoever wants to move fill / cur (a launch out of the staging sets, flush, reset) raises
    // quiesce, waits on wr_cv for writers == 0 (no new copy starts meanwhile) and lowers it again.
    std::vector<uint8_t> writing;
    std::vector<uint8_t> pushing;              // stream s has a push call in progress (one pusher per stream, for the whole call)
    std::vector<uint8_t> closing;              // how many finish / reset calls are ending or restarting stream s right now (StreamClose below)
    std::vector<int64_t> last_push_ns;         // when stream s last delivered samples (nvx_now_ns; create / reset count as a delivery)
    std::vector<int64_t> stall_ns;             // how long the others' launches wait for stream s after that (0 = for ever): cfg.stall_timeout_ms,
                                               // or the timeout of the capture ring attached to the stream (nvx_capture_set_stall_timeout)
    int writers = 0, quiesce = 0;
    std::condition_variable wr_cv;
};

struct SinkCtx { nvx_handle *h; int stream; int slot; };

// Scope in which the staging sets may be rearranged: raised with the handle locked, waits (lock released meanwhile)
// until the unlocked copies in flight have been committed; pushes start no new unlocked copy while one is up.
struct StagingQuiesce {
    nvx_handle *h; std::unique_lock<std::mutex> &lk;
    StagingQuiesce(nvx_handle *h_, std::unique_lock<std::mutex> &lk_) : h(h_), lk(lk_)
    {
        h->quiesce++;
        h->wr_cv.wait(lk, [&] { return h->writers == 0; });
    }
    ~StagingQuiesce() { h->quiesce--; h->wr_cv.notify_all(); }
};

// Scope in which streams [first, last) of a push-mode handle are ENDED or RESTARTED (nvx_finish, nvx_stream_finish,
// nvx_stream_reset, nvx_reset).  A push call is atomic against these: the calls already in progress on the streams run
// to their end first (each is finite: a pusher that needs a launch makes it itself), and a call that arrives meanwhile
// waits at its entry until the scope is left -- then it finds the stream ended (NVX_ERR_STATE, nothing staged) or
// fresh.  Without it a pusher that had released the lock inside its loop went on staging samples into a stream that
// nvx_finish had ended under it, and that stream then vetoed every launch of the handle.
// Raised with the handle locked and BEFORE a StagingQuiesce (a pusher in its loop waits for quiesce == 0).
struct StreamClose {
    nvx_handle *h; int first, last;
    StreamClose(nvx_handle *h_, std::unique_lock<std::mutex> &lk, int first_, int last_) : h(h_), first(first_), last(last_)
    {
        if (h->pushing.empty()) { first = last = 0; return; }           // not a push-mode handle: nobody pushes
        for (int s = first; s < last; s++) h->closing[s]++;
        h->wr_cv.wait(lk, [&] { for (int s = first; s < last; s++) if (h->pushing[s]) return false; return true; });
    }
    ~StreamClose() { for (int s = first; s < last; s++) h->closing[s]--; if (first != last) h->wr_cv.notify_all(); }
};

// launch cascade + demodulator over n_frames frames of [n_streams][pitch] packed IQ (handle locked)
// part / n_part: the input streams that take part, ascending (nullptr = every stream)
// tail_n3 (per participant): the launch ends these streams -- so many of its 900 S/s samples are real (nvx_api.cpp)
int nvx_launch_locked(nvx_handle *h, const void *d_iq, size_t pitch, size_t first_sample, int n_frames, hipStream_t st,
                      const int *part = nullptr, int n_part = 0, const int *tail_n3 = nullptr);
// nvx_push_iq that reports how many samples it staged before an error stopped it
int nvx_push_iq_partial(nvx_handle *h, int stream, const int16_t *iq, size_t n, size_t *accepted);
// wait for the launched blocks in front of launch number `upto` (default: every one), append bits, run the character
// layer (handle locked)
int nvx_collect_locked(nvx_handle *h, uint64_t upto = UINT64_MAX);
// ... the same for every launched block that has ALREADY finished: never waits (handle locked)
int nvx_collect_ready_locked(nvx_handle *h);
// sets the "nvx_reset required" error text of a poisoned handle and returns NVX_ERR_STATE (handle locked)
int nvx_poisoned_error(nvx_handle *h);
// launches whose results have not been taken in yet (takes the handle's lock)
int nvx_launches_in_flight(nvx_handle *h);
// launches sent on their way since create (takes the handle's lock)
uint64_t nvx_launch_count(nvx_handle *h);
// bit-period transition tables of the demodulator FSM (nvx_fsm.h), NVX_FSM_TABLE_ALLOC entries
const uint32_t *nvx_fsm_table_host();

#endif
