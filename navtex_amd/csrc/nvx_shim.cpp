// nvx_shim.cpp -- the reference-compatible surface (header sections A, B): the three symbols
// capt_sched.c links against, the SDRplay-shaped stream callback, and the weak default add_message.
#include "nvx_handle.h"

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

// ===========================================================================
// reference-compatible push surface + stream callback (sections A, B)
// ===========================================================================
static nvx_handle *g_shim = nullptr;
static std::mutex g_shim_mu;                     // callback re-entrancy (capt_sched.c:111)
static int16_t g_shim_buf[2 * 4096];
static size_t g_shim_n = 0;
// Decode latency of this surface, booked like a capture ring's (nvx_handle.h, ArrivalClock): the moment the call that
// carried a frame's LAST sample was entered -> its bits pollable and its messages at add_message (nvx_shim_latency).
static ArrivalClock g_shim_clock;
static uint64_t g_shim_samples = 0;              // samples taken since init_fir_filter1
static bool g_shim_finished = false;             // nvx_shim_finish has ended the input: the next sample starts a new stream
static std::atomic<uint64_t> g_shim_total{ 0 }, g_shim_off_domain{ 0 };   // sample_in_1 calls since the library was loaded / outside its input domain

static void shim_fatal(const char *what)
{
    fprintf(stderr, "navtex_amd: %s: %s\n", what, nvx_last_error());
    abort();                                     // void reference entry points cannot report errors
}

// Housekeeping for the singleton: every 50 ms (the reference's own poll interval, capt_sched.c:486) take in what the GPU
// has finished, so that a message completes at add_message although no further sample arrives (nvx_poll never waits).
static std::thread g_keeper;
static std::atomic<bool> g_keeper_stop{ false };
static std::mutex g_keeper_mu; static std::condition_variable g_keeper_cv;
static bool g_keeper_kick = false;               // (under g_keeper_mu) a push has sent a launch on its way: look now
static void keeper_stop(void)
{
    { std::lock_guard<std::mutex> lk(g_keeper_mu); g_keeper_stop.store(true); }
    g_keeper_cv.notify_all();
    if (g_keeper.joinable()) g_keeper.join();
}
static void keeper_loop(nvx_handle *h)
{
    std::unique_lock<std::mutex> lk(g_keeper_mu);
    while (!g_keeper_stop.load()) {
        // 50 ms, the reference's own poll interval (capt_sched.c:486); 2 ms while a launch is in flight, so that its results
        // reach add_message within milliseconds even when no further sample arrives
        lk.unlock();
        const int wait_ms = nvx_launches_in_flight(h) ? 2 : 50;
        lk.lock();
        if (g_keeper_stop.load()) break;
        g_keeper_cv.wait_for(lk, std::chrono::milliseconds(wait_ms), [] { return g_keeper_stop.load() || g_keeper_kick; });
        g_keeper_kick = false;
        if (g_keeper_stop.load()) break;
        lk.unlock();
        if (nvx_poll(h) != NVX_OK) { fprintf(stderr, "navtex_amd: housekeeping: %s\n", nvx_last_error()); lk.lock(); break; }
        lk.lock();
    }
}

static void shim_require(void)
{
    if (g_shim) return;
    nvx_config c; nvx_config_default(&c);
    c.n_streams = 1; c.raw_rate = 0; c.chain_mask = NVX_CHAIN_518 | NVX_CHAIN_490;   // nav_sched.C:10-17
    c.max_frames = 4; c.char_layer = 1; c.push_mode = 1;
    if (const char *d = getenv("NAVTEX_AMD_DEVICE")) c.device = atoi(d);
    if (nvx_create(&c, &g_shim) != NVX_OK) shim_fatal("cannot create the GPU pipeline");
    { std::lock_guard<std::mutex> lk(g_shim->mu); g_shim->arrival[0] = &g_shim_clock; g_shim->n_arrival++; }
    // NAVTEX_AMD_TRACE=1: the character layers print what the reference's print to stdout ("phasing detected", "START OF
    // MESSAGE", "line added: ...", "END OF MESSAGE", "end of emission detected", ...: receiver/nav_b_sm.C), chain by chain as
    // the frames are decoded -- for a receiver whose operator reads those lines in the journal
    if (getenv("NAVTEX_AMD_TRACE") && atoi(getenv("NAVTEX_AMD_TRACE")))
        nvx_set_trace(g_shim, [](void *, const char *text) { fputs(text, stdout); }, nullptr);
    if (!(getenv("NAVTEX_AMD_NO_KEEPER") && atoi(getenv("NAVTEX_AMD_NO_KEEPER")))) {
        g_keeper = std::thread(keeper_loop, g_shim);
        atexit(keeper_stop);                       // joined before the HIP runtime (loaded earlier) tears down
    }
}

// a push may have sent a launch on its way: wake the housekeeping thread, which then looks every 2 ms until the results are in
// (asleep it would notice up to 50 ms later: the latency of this surface was wherever its wake-ups happened to fall)
static void shim_wake_keeper(void)
{
    static uint64_t seen = 0;                    // (one caller thread at a time: the reference's contract, and g_shim_mu)
    if (!g_keeper.joinable()) return;
    const uint64_t launched = nvx_launch_count(g_shim);
    if (launched == seen) return;                // one kick per launch: a file replayed at full speed pushes 25 000 times a second
    seen = launched;
    { std::lock_guard<std::mutex> lk(g_keeper_mu); g_keeper_kick = true; }          // (a flag, so that a kick between the keeper's look and its wait is not lost)
    g_keeper_cv.notify_one();
}

static void shim_drain(void)
{
    if (g_shim_n && nvx_push_iq(g_shim, 0, g_shim_buf, g_shim_n) != NVX_OK) shim_fatal("push failed");
    g_shim_n = 0;
    shim_wake_keeper();
}

// A NEW stream on the singleton: everything the path carries is zeroed, what was buffered but not yet decoded is dropped.
// (g_shim_mu held.)
static void shim_new_stream(void)
{
    shim_require();
    g_shim_n = 0;
    if (nvx_reset(g_shim) != NVX_OK) shim_fatal("reset failed");
    g_shim_samples = 0;                          // (the reset ended the clock's bookkeeping: frame 0 starts here again)
    { std::lock_guard<std::mutex> ck(g_shim_clock.mu); g_shim_clock.base = 0; g_shim_clock.stamped = 0; }
    g_shim_finished = false;
}

// DEVIATION from the reference (header section A): the reference's init_fir_filter1 clears FIR1's
// ring and counter and nothing else (receiver/fir1cpp.C:65-77).  capt_sched.c calls it once, before the first sample
// (:552-555), where "FIR1 cleared" and "everything cleared" are the same thing; called again in mid-stream the reference
// would go on with FIR2, FIR3, both decoders and both character layers as they were -- a state the frame algebra here has
// no word for (a frame is the span after which every decimation counter is back at phase 0 TOGETHER).  Here every call
// starts a new stream.
extern "C" void init_fir_filter1(void)           // receiver/fir1cpp.C:65-77
{
    std::lock_guard<std::mutex> lk(g_shim_mu);
    shim_new_stream();
}

// DEVIATION, the same way: the reference's second init_fir2_wrapper would clear the 518 chain's FIR2 and the mixer index
// both chains share (receiver/fir2cpp.C:90-110) in mid-stream; here it wires nothing twice and changes nothing.
extern "C" void init_fir2_wrapper(void)          // receiver/nav_sched.C:19-22
{
    std::lock_guard<std::mutex> lk(g_shim_mu);
    shim_require();                              // the object graph already exists; nothing else to wire
}

// The input domain of sample_in_1.  The reference filters whatever double it is given; its only caller hands it
// (double) of an int16 (receiver/capt_sched.c:511), and int16 is what the GPU path is built on.  A value outside that
// domain -- not an integer, beyond +-32767 / -32768, not a number -- is brought to the nearest int16 (ties to even; NaN
// to 0): defined behaviour, at most half an LSB (or the clipping) away from what the reference would have filtered,
// and COUNTED (nvx_shim_stats) -- never the undefined (int16_t) cast of an out-of-range double this once was.
extern "C" int nvx_sample_to_int16(double v, int16_t *out)
{
    int16_t r; int exact;
    if (v != v) { r = 0; exact = 0; }
    else if (v >= 32767.0) { r = 32767; exact = v == 32767.0; }
    else if (v <= -32768.0) { r = -32768; exact = v == -32768.0; }
    else { const double n = __builtin_rint(v); r = (int16_t)n; exact = n == v; }
    if (out) *out = r;
    return exact;
}

extern "C" void sample_in_1(double sample_I, double sample_Q)   // receiver/fir1cpp.C:80
{
    if (!g_shim) { std::lock_guard<std::mutex> lk(g_shim_mu); shim_require(); }
    // a sample behind nvx_shim_finish: the input that ended there cannot be continued (its filters ran past its last
    // sample), so this is the first sample of a new stream -- as if init_fir_filter1 had been called in between
    if (g_shim_finished) { std::lock_guard<std::mutex> lk(g_shim_mu); shim_new_stream(); }
    // capt_sched.c:511 passes (double) of int16 values: the conversion back is exact for those (and checked: above)
    const int ok = nvx_sample_to_int16(sample_I, &g_shim_buf[2 * g_shim_n]) & nvx_sample_to_int16(sample_Q, &g_shim_buf[2 * g_shim_n + 1]);
    // (one caller thread, the reference's contract: plain increments of what nvx_shim_stats may read from another thread)
    g_shim_total.store(g_shim_total.load(std::memory_order_relaxed) + 1, std::memory_order_relaxed);
    if (!ok && g_shim_off_domain.fetch_add(1, std::memory_order_relaxed) == 0) {
        nvx_set_error("sample_in_1(%g, %g): not an int16 value (receiver/capt_sched.c:511 passes int16); rounded / clipped, counted in nvx_shim_stats", sample_I, sample_Q);
        fprintf(stderr, "navtex_amd: %s\n", nvx_last_error());
    }
    ++g_shim_n;
    // the sample that completes a frame goes to the pipeline at once (its launch should not wait for up to 4095 more
    // samples, 16 ms at the real rate), and the frame's arrival is stamped for the latency bookkeeping
    const bool frame_complete = ++g_shim_samples % NVX_FRAME_IN == 0;
    if (frame_complete) g_shim_clock.stamp(g_shim_samples / NVX_FRAME_IN - 1, nvx_now_ns());
    if (g_shim_n == 4096 || frame_complete) shim_drain();
}

extern "C" int nvx_shim_latency(uint64_t *frames, double *p50_ms, double *p99_ms, double *max_ms, double *last_ms, int reset)
{
    if (!g_shim) { nvx_set_error("shim not initialised"); return NVX_ERR_STATE; }
    nvx_clock_report(g_shim_clock, frames, p50_ms, p99_ms, max_ms, last_ms, reset);
    return NVX_OK;
}

extern "C" int nvx_shim_flush(void)
{
    std::lock_guard<std::mutex> lk(g_shim_mu);
    if (!g_shim) { nvx_set_error("shim not initialised"); return NVX_ERR_STATE; }
    shim_drain();
    return nvx_flush(g_shim);
}

extern "C" int nvx_shim_finish(void)
{
    std::lock_guard<std::mutex> lk(g_shim_mu);
    if (!g_shim) { nvx_set_error("shim not initialised"); return NVX_ERR_STATE; }
    shim_drain();
    g_shim_finished = true;                      // whatever comes next is a new stream (sample_in_1, nvx_StreamACallback)
    return nvx_finish(g_shim);                   // the last, partial frame at its true length (capt_sched.c:509-513 stops with its last sample)
}

extern "C" int nvx_shim_stats(uint64_t *samples, uint64_t *off_domain)
{
    if (samples) *samples = g_shim_total.load();
    if (off_domain) *off_domain = g_shim_off_domain.load();
    return NVX_OK;
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
    if (!h || h == g_shim) {                     // the singleton, named or not: its samples count for nvx_shim_latency
        const int64_t t_enter = nvx_now_ns();
        shim_require();
        if (g_shim_finished) shim_new_stream();  // (as sample_in_1: samples behind nvx_shim_finish start a new stream)
        shim_drain(); h = g_shim;
        for (uint64_t f = g_shim_samples / NVX_FRAME_IN; f < (g_shim_samples + numSamples) / NVX_FRAME_IN; f++) g_shim_clock.stamp(f, t_enter);
        g_shim_samples += numSamples;
    }
    if (nvx_push_planar(h, 0, xi, xq, numSamples) != NVX_OK) shim_fatal("stream callback push failed");
    if (h == g_shim) shim_wake_keeper();
}
