// nvx_capture.cpp -- live-capture ring (header section B').
#include "nvx_handle.h"
#include <chrono>

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
    // A radio that goes silent (unplugged: receiver/capt_sched.c:210-212 only prints sdrplay_api_DeviceRemoved) must not
    // hold the other streams of the handle: after stall_ms without a sample handed on, the consumer marks its stream
    // inactive (nvx_stream_set_active) -- launches stop waiting for it -- and says so (nvx_capture_stalled).  The next
    // sample it hands on makes the stream active again; it continues from its own carried state.
    std::atomic<int> stall_ms{ 2000 };
    std::atomic<int> silent{ 0 };
    std::atomic<uint64_t> stall_events{ 0 };
    ArrivalClock clock;                         // frame arrival stamps and booked decode latencies (nvx_handle.h)
    size_t frame_in = 0;                        // complex samples per frame at the handle's input rate
    std::mutex rec_mu; nvx_wav *rec = nullptr;  // debug recording of what the consumer hands on (capt_sched.c:87-101, 516)
    std::thread worker;
};

static void capture_consumer(nvx_capture *c)
{
    // the stall clock runs from the first sample handed on (a radio may take seconds to deliver its first one: SDR
    // initialisation is not a stall) and does not count time the ring was paused
    auto last_progress = std::chrono::steady_clock::now();
    bool seen_any = false;
    for (;;) {
        {
            // The reference polls every 50 ms (capt_sched.c:486); this consumer is woken by the callback, and falls back to
            // the 50 ms.  While a launch is in flight it looks again after 2 ms, so that the results of the LAST frame before
            // the radio goes quiet (no callback left to wake it) reach the sink within milliseconds, not within a poll interval.
            const int wait_ms = nvx_launches_in_flight(c->h) ? 2 : 50;
            std::unique_lock<std::mutex> lk(c->cv_mu);
            c->cv.wait_for(lk, std::chrono::milliseconds(wait_ms), [&] {
                return c->stop.load() || (!c->paused.load() && c->head.load() != c->tail.load());
            });
        }
        // results of launches that have finished meanwhile: bits and messages reach the user within a poll interval even
        // when no further launch follows (the last message before the band goes quiet); never waits for the GPU
        if (int rc = nvx_poll(c->h); rc != NVX_OK) { c->error.store(rc); c->stop.store(true); return; }
        if (c->paused.load()) last_progress = std::chrono::steady_clock::now();
        const bool idle = c->head.load() == c->tail.load();
        const int limit = c->stall_ms.load();
        if (idle && seen_any && !c->paused.load() && !c->stop.load() && !c->silent.load() && limit > 0 &&
            std::chrono::steady_clock::now() - last_progress > std::chrono::milliseconds(limit)) {
            if (nvx_stream_set_active(c->h, c->stream, 0) == NVX_OK) { c->silent.store(1); c->stall_events.fetch_add(1); }
        }
        if (c->paused.load() && !c->stop.load()) continue;
        uint64_t t = c->tail.load(), hd = c->head.load();
        while (t != hd && !(c->paused.load() && !c->stop.load())) {      // contiguous spans, wrap split as capt_sched.c:494-503
            size_t at = (size_t)(t % c->cap);
            size_t n = (size_t)std::min<uint64_t>(hd - t, c->cap - at);
            size_t took = 0;
            int rc = nvx_push_iq_partial(c->h, c->stream, c->ring.data() + 2 * at, n, &took);
            if (rc != NVX_OK) { c->error.store(rc); c->stop.store(true); return; }      // a failed launch or HIP call: give up, report
            n = took;
            if (n) {
                std::lock_guard<std::mutex> lk(c->rec_mu);
                if (c->rec && nvx_wav_write(c->rec, c->ring.data() + 2 * at, n) != n) {     // disk full etc.: stop recording, keep decoding
                    nvx_wav_close(c->rec); c->rec = nullptr;
                }
            }
            t += n;
            c->tail.store(t);
            c->consumed.fetch_add(n);
            if (n) { last_progress = std::chrono::steady_clock::now(); seen_any = true; c->silent.store(0); }     // (the push made the stream active again)
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
    c->frame_in = h->frame_in;
    {
        // latency bookkeeping: the handle's frame count of this stream now = frame 0 of the capture (whole frames; samples
        // already staged for the stream would shift the capture's frames by less than one)
        std::lock_guard<std::mutex> lk(h->mu);
        c->clock.base = h->g0s[stream] / NVX_FRAME_Y3 + (h->fill.empty() ? 0 : h->fill[stream] / h->frame_in);
        if (!h->arrival[stream]) h->n_arrival++;
        h->arrival[stream] = &c->clock;
        c->stall_ms.store((int)(h->stall_ns[stream] / 1000000));        // one timeout for the stream: the handle's, until nvx_capture_set_stall_timeout
    }
    c->worker = std::thread(capture_consumer, c);
    *out = c;
    return NVX_OK;
}

extern "C" void nvx_capture_callback(short *xi, short *xq, void *params, unsigned int numSamples, unsigned int reset, void *cbContext)
{
    (void)params; (void)reset;
    nvx_capture *c = (nvx_capture *)cbContext;
    if (!c || !xi || !xq) return;
    const int64_t t_enter = nvx_now_ns();
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
    // frames whose last sample this call carried (counted in samples the ring took: what the handle will see)
    for (uint64_t f = hd / c->frame_in; f < (hd + n) / c->frame_in; f++) c->clock.stamp(f, t_enter);
    c->cv.notify_one();
}

void nvx_clock_report(ArrivalClock &c, uint64_t *frames, double *p50_ms, double *p99_ms, double *max_ms, double *last_ms, int reset)
{
    std::vector<float> v;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        v = c.lat_ms;
        if (frames) *frames = c.booked;
        if (max_ms) *max_ms = c.booked ? (double)c.max_ms : -1.0;
        if (last_ms) *last_ms = c.booked ? (double)c.last_ms : -1.0;
        if (reset) { c.lat_ms.clear(); c.booked = 0; c.max_ms = 0.f; c.last_ms = 0.f; }
    }
    std::sort(v.begin(), v.end());
    auto pct = [&](double q) { return v.empty() ? -1.0 : (double)v[std::min(v.size() - 1, (size_t)(q * (double)v.size()))]; };
    if (p50_ms) *p50_ms = pct(0.50);
    if (p99_ms) *p99_ms = pct(0.99);
}

extern "C" int nvx_capture_latency(nvx_capture *c, uint64_t *frames, double *p50_ms, double *p99_ms, double *max_ms, double *last_ms, int reset)
{
    if (!c) return NVX_ERR_ARG;
    nvx_clock_report(c->clock, frames, p50_ms, p99_ms, max_ms, last_ms, reset);
    return NVX_OK;
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

extern "C" int nvx_capture_error(nvx_capture *c)
{
    if (!c) return NVX_ERR_ARG;
    return c->error.load();
}

extern "C" int nvx_capture_stalled(nvx_capture *c, uint64_t *stall_events)
{
    if (!c) return NVX_ERR_ARG;
    if (stall_events) *stall_events = c->stall_events.load();
    return c->silent.load();
}

extern "C" void nvx_capture_set_stall_timeout(nvx_capture *c, double seconds)
{
    if (!c) return;
    const int ms = seconds > 0 ? (int)(seconds * 1000.0 + 0.5) : 0;
    c->stall_ms.store(ms);
    // ... and the same figure at the push level (nvx_push.cpp, lockstep_ready), so that ONE timeout says how long the other
    // streams' launches wait for this radio: the ring's (0 = for ever)
    std::lock_guard<std::mutex> lk(c->h->mu);
    c->h->stall_ns[c->stream] = (int64_t)ms * 1000000;
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
    {
        std::lock_guard<std::mutex> lk(c->h->mu);                        // no collect books into the clock any more
        if (c->h->arrival[c->stream] == &c->clock) { c->h->arrival[c->stream] = nullptr; c->h->n_arrival--; }
    }
    delete c;
    return rc;
}
