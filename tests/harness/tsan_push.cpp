// ThreadSanitizer driver for the host-input path (navtex_amd/csrc/nvx_push.cpp) without a GPU: the staging sets, the
// per-stream flips, partial launches, the unlocked copies with their quiesce protocol, flush, the end of the input
// (nvx_finish: a ragged tail per stream at its true length) and the activity flag run for real; the HIP calls and the launch behind them are replaced by a "device" that copies synchronously and records
// what every launch took from every stream.
// Checked: no data race; every stream's samples arrive at the "device" exactly once and in order whatever the mix of
// pusher threads, push sizes, silent streams and flushes; launches cover ascending, distinct streams.
// Built by tests/test_sanitizers.py with -fsanitize=thread.
#include "nvx_handle.h"

#include <chrono>

extern "C" void nvx_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int nvx_poisoned_error(nvx_handle *) { return NVX_ERR_STATE; }
int64_t nvx_now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int nvx_collect_locked(nvx_handle *, uint64_t) { return NVX_OK; }
int nvx_collect_ready_locked(nvx_handle *h) { h->collected = h->launched; return NVX_OK; }      // (called on the way out of every push)

// ---- the "device": d_in is host memory here; a launch appends what it was given to the stream's received sequence
static std::vector<std::vector<uint32_t>> g_got;      // per stream, under the handle's lock
static std::atomic<int> g_bad{ 0 }, g_launches{ 0 }, g_partial{ 0 }, g_tails{ 0 };

hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < height; r++) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
    return hipSuccess;
}
// (tail_n3: the launch that ends streams.  One "900 S/s sample" is 280 input samples; what reaches the device behind a
// stream's last sample must be zeros, and the stream is ended afterwards -- as the real nvx_launch_locked does)
int nvx_launch_locked(nvx_handle *h, const void *d_iq, size_t pitch, size_t, int n_frames, hipStream_t, const int *part, int n_part, const int *tail_n3)
{
    g_launches++;
    if (part && !tail_n3) g_partial++;
    if (tail_n3) g_tails++;
    const int n = part ? n_part : h->n_in;
    for (int i = 0; i < n; i++) {
        const int s = part ? part[i] : i;
        if (part && i > 0 && part[i] <= part[i - 1]) g_bad++;
        if (h->ended[s]) g_bad++;
        const uint32_t *row = (const uint32_t *)d_iq + (size_t)s * pitch;
        g_got[s].insert(g_got[s].end(), row, row + (size_t)n_frames * h->frame_in);
        if (tail_n3) { if (n_frames != 1 || tail_n3[i] < 1 || (size_t)tail_n3[i] * 280 > h->frame_in) g_bad++; h->ended[s] = 1; }
    }
    h->launched++;
    return NVX_OK;
}

int main()
{
    const int S = 6, max_frames = 2;
    nvx_handle h;
    h.cfg.push_mode = 1; h.cfg.max_frames = max_frames; h.cfg.device = 0;
    h.n_in = h.n_streams = S;
    h.frame_in = 48 * 1024;                                  // 192 KB per frame: pushes above and below the unlocked-copy threshold
    h.stage_cap = (size_t)(max_frames + 1) * h.frame_in;
    std::vector<uint32_t> stage[2], din((size_t)S * max_frames * h.frame_in);
    for (int i = 0; i < 2; i++) { stage[i].assign((size_t)S * h.stage_cap, 0); h.h_stage[i] = stage[i].data(); h.set_launch[i].assign(S, 0); }
    h.d_in = din.data();
    h.fill.assign(S, 0); h.cur.assign(S, 0); h.active.assign(S, 1); h.writing.assign(S, 0); h.pushing.assign(S, 0); h.last_push_ns.assign(S, nvx_now_ns());
    h.parity.assign(S, 0); h.g0s.assign(S, 0); h.ended.assign(S, 0); h.stall_ns.assign(S, 2000000000ll);
    g_got.assign(S, {});

    const size_t frames_total = 40;
    // per stream: whole frames, plus a ragged tail on the odd streams (stream 5's is too short for one 900 S/s sample)
    auto total_of = [&](int s) { return frames_total * h.frame_in + (s == 5 ? 100 : (s & 1) ? 1000 * (size_t)s + 7 : 0); };
    std::vector<std::thread> pushers;
    std::atomic<int> errors{ 0 };
    for (int s = 0; s < S; s++)
        pushers.emplace_back([&, s] {                        // one capture / replay thread per stream
            std::vector<int16_t> buf;
            unsigned x = 77u * (unsigned)(s + 1);
            size_t pos = 0;
            const size_t total = total_of(s);
            while (pos < total) {
                x = x * 1664525u + 1013904223u;
                size_t m = (x >> 28) < 5 ? 1 + (x >> 8) % 3000 : 20000 + (x >> 8) % 90000;      // callback-sized and replay-sized pushes
                m = std::min(m, total - pos);
                buf.resize(2 * m);
                for (size_t k = 0; k < m; k++) { const uint32_t v = (uint32_t)(pos + k) ^ ((uint32_t)s << 28); buf[2 * k] = (int16_t)(v & 0xffff); buf[2 * k + 1] = (int16_t)(v >> 16); }
                if (nvx_push_iq(&h, s, buf.data(), m) != NVX_OK) errors++;
                pos += m;
                if (s == 2 && (x >> 20) % 16 == 0) std::this_thread::sleep_for(std::chrono::milliseconds(3));     // a slow radio: the others run ahead of it
                if (s == 4 && pos > total / 3 && pos < total / 3 + 100000) { nvx_stream_set_active(&h, 4, 0); std::this_thread::sleep_for(std::chrono::milliseconds(20)); }
            }
        });
    std::thread flusher([&] { for (int i = 0; i < 30; i++) { std::this_thread::sleep_for(std::chrono::milliseconds(2)); if (nvx_flush(&h) != NVX_OK) errors++; } });
    for (auto &t : pushers) t.join();
    flusher.join();
    if (nvx_flush(&h) != NVX_OK) return 2;
    // a flush leaves the ragged tails staged ...
    for (int s = 0; s < S; s++) if (g_got[s].size() != frames_total * h.frame_in) { fprintf(stderr, "stream %d: a flush launched a partial frame\n", s); return 6; }
    // ... the end of the input runs them: ONE launch, the streams that hold at least one 900 S/s sample, a frame each, zeros behind
    if (nvx_finish(&h) != NVX_OK) return 2;
    for (int s = 0; s < S; s++) {
        const size_t total = total_of(s), tail = total - frames_total * h.frame_in;
        const size_t want = frames_total * h.frame_in + (tail >= 280 ? h.frame_in : 0);
        if (g_got[s].size() != want) { fprintf(stderr, "stream %d: %zu of %zu samples reached the device\n", s, g_got[s].size(), want); return 3; }
        for (size_t k = 0; k < want; k++)
            if (g_got[s][k] != (k < total ? ((uint32_t)k ^ ((uint32_t)s << 28)) : 0u)) { fprintf(stderr, "stream %d: sample %zu is wrong\n", s, k); return 4; }
        if ((h.ended[s] != 0) != (tail > 0)) { fprintf(stderr, "stream %d: ended flag %d with a tail of %zu\n", s, (int)h.ended[s], tail); return 7; }
        int16_t one[2] = { 1, 1 };
        if ((nvx_push_iq(&h, s, one, 1) == NVX_OK) != (tail == 0)) { fprintf(stderr, "stream %d: push after the end\n", s); return 8; }
    }
    printf("launches %d (partial %d, tails %d), bad %d, errors %d\n", g_launches.load(), g_partial.load(), g_tails.load(), g_bad.load(), errors.load());
    if (g_bad || errors || g_partial == 0 || g_tails != 1) return 5;
    printf("tsan push ok\n");
    return 0;
}
