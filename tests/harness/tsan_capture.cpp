// ThreadSanitizer driver for the live-capture ring (navtex_amd/csrc/nvx_capture.cpp) without a GPU:
// the ring, its producer callback, the consumer thread, pause / overrun accounting and the debug
// recording run for real; the GPU pipeline behind nvx_push_iq / nvx_flush is replaced by a checker
// that verifies the consumer hands on exactly the samples that were accepted, in order.
// Built by tests/test_sanitizers.py with -fsanitize=thread.
#include "nvx_handle.h"

#include <chrono>

static std::atomic<uint64_t> g_pushed{ 0 }, g_bad{ 0 };
static uint64_t g_expect = 0;                    // next sample index the checker expects (consumer thread only)
static std::vector<uint64_t> g_gaps;             // sample indices dropped by the producer, in order (under g_mu)
static std::atomic<size_t> g_ngaps{ 0 };         // == g_gaps.size()
static size_t g_gap_pos = 0;                     // gaps already skipped (consumer thread only)
static std::mutex g_mu;

extern "C" void nvx_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

// sample n carries I = low 15 bits of n, Q = next 15 bits: the checker recomputes n
static inline void encode(uint64_t n, short *i, short *q) { *i = (short)(n & 0x7fff); *q = (short)((n >> 15) & 0x7fff); }

extern "C" int nvx_push_iq(nvx_handle *, int, const int16_t *iq, size_t n)
{
    for (size_t k = 0; k < n; k++) {
        if (g_ngaps.load() > g_gap_pos) {           // dropped samples are rare: take the lock only then
            std::lock_guard<std::mutex> lk(g_mu);
            while (g_gap_pos < g_gaps.size() && g_gaps[g_gap_pos] == g_expect) { g_gap_pos++; g_expect++; }
        }
        const uint64_t got = (uint64_t)(uint16_t)iq[2 * k] | ((uint64_t)(uint16_t)iq[2 * k + 1] << 15);
        if (got != (g_expect & 0x3fffffff)) g_bad++;
        g_expect++;
    }
    g_pushed += n;
    return NVX_OK;
}
// what the consumer really calls
int nvx_push_iq_partial(nvx_handle *h, int stream, const int16_t *iq, size_t n, size_t *accepted)
{
    nvx_push_iq(h, stream, iq, n);
    *accepted = n;
    return NVX_OK;
}
int64_t nvx_now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int nvx_launches_in_flight(nvx_handle *) { static std::atomic<unsigned> n{ 0 }; return (int)(++n % 3 == 0); }    // now and then "a launch is in flight": the short wait
extern "C" int nvx_flush(nvx_handle *) { return NVX_OK; }
extern "C" int nvx_poll(nvx_handle *) { return NVX_OK; }                 // the consumer's every-wake "take in what has finished"
// the consumer's silent-radio report (nvx_capture.cpp: stall timeout) lands here
static std::atomic<int> g_marked_inactive{ 0 };
extern "C" int nvx_stream_set_active(nvx_handle *, int, int active) { if (!active) g_marked_inactive++; return NVX_OK; }

int main(int argc, char **argv)
{
    nvx_handle h;                                  // only cfg / n_in are read by the capture code
    h.cfg.push_mode = 1; h.cfg.raw_rate = 0; h.cfg.wideband = 0; h.n_in = 1;
    h.frame_in = 4096; h.g0s.assign(1, 0); h.arrival.assign(1, nullptr); h.stall_ns.assign(1, 2000000000ll);      // what nvx_capture_start reads for its frame clock and timeout
    nvx_capture *cap = nullptr;
    if (nvx_capture_start(&h, 0, 0.02, &cap) != NVX_OK) return 2;      // 5040-sample ring: wraps constantly
    if (argc > 1 && nvx_capture_record(cap, argv[1]) != NVX_OK) return 3;

    const uint64_t total = 200000;
    std::thread vendor([&] {                      // the SDR library's callback thread
        std::vector<short> xi(4096), xq(4096);
        uint64_t n = 0; unsigned x = 1;
        while (n < total) {
            x = x * 1664525u + 1013904223u;
            unsigned m = 1 + (x >> 20) % 3000; if (n + m > total) m = (unsigned)(total - n);
            for (unsigned k = 0; k < m; k++) encode(n + k, &xi[k], &xq[k]);
            uint64_t r0, d0, c0; nvx_capture_stats(cap, &r0, &d0, &c0);
            nvx_capture_callback(xi.data(), xq.data(), nullptr, m, 0, cap);
            uint64_t r1, d1, c1; nvx_capture_stats(cap, &r1, &d1, &c1);
            if (d1 > d0) {                        // the newest (d1 - d0) samples of this block were dropped
                std::lock_guard<std::mutex> lk(g_mu);
                for (uint64_t k = m - (d1 - d0); k < m; k++) g_gaps.push_back(n + k);
                g_ngaps.store(g_gaps.size());
            }
            n += m;
            if ((x >> 8) % 7 == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    });
    uint64_t booked_frames = 0;
    auto book_stamped = [&](uint64_t &next) {     // the frames the callback has stamped so far and nobody has booked yet
        std::lock_guard<std::mutex> lk(h.mu);
        ArrivalClock *ac = h.arrival[0];
        if (!ac) return;
        uint64_t upto;
        { std::lock_guard<std::mutex> al(ac->mu); upto = ac->stamped; }
        while (next < upto) ac->book(next++, nvx_now_ns());
    };
    uint64_t booked_upto = 0;
    std::thread meddler([&] {                     // pauses and resumes the consumer: provokes overruns
        uint64_t &next = booked_upto;
        for (int i = 0; i < 12; i++) {
            nvx_capture_pause(cap, 1); std::this_thread::sleep_for(std::chrono::milliseconds(2));
            nvx_capture_pause(cap, 0); std::this_thread::sleep_for(std::chrono::milliseconds(3));
            // ... and plays the handle's collect: books the frames stamped so far (the frame clock the callback writes), and
            // reads the statistics, while the callback keeps stamping
            book_stamped(next);
            double p50, p99, mx, last;
            if (nvx_capture_latency(cap, &booked_frames, &p50, &p99, &mx, &last, 0) != NVX_OK) g_bad++;
        }
    });
    vendor.join(); meddler.join();
    {
        book_stamped(booked_upto);                // whatever was stamped after the meddler's last look
        double p50, p99, mx, last;
        if (nvx_capture_latency(cap, &booked_frames, &p50, &p99, &mx, &last, 0) != NVX_OK) g_bad++;
    }
    uint64_t rx, dropped, used;
    nvx_capture_stats(cap, &rx, &dropped, &used);
    if (nvx_capture_error(cap) != NVX_OK) return 6;
    if (nvx_capture_stop(cap) != NVX_OK) return 4;
    const uint64_t pushed = g_pushed.load();
    printf("received %llu dropped %llu pushed %llu bad %llu\n", (unsigned long long)rx, (unsigned long long)dropped,
           (unsigned long long)pushed, (unsigned long long)g_bad.load());
    if (booked_frames == 0 || h.arrival[0] != nullptr) return 8;       // latencies were booked; the clock was unregistered at stop
    if (rx != total || pushed + dropped != total || g_bad.load() != 0) return 5;
    printf("tsan capture ok\n");
    return 0;
}
