// ThreadSanitizer driver for the host-input path (navtex_amd/csrc/nvx_push.cpp) without a GPU: the staging sets, the
// per-stream flips, partial launches, the unlocked copies with their quiesce protocol, flush, the end of the input
// (nvx_finish: a ragged tail per stream at its true length) and the activity flag run for real; the HIP calls and the launch behind them are replaced by a "device" that copies synchronously and records
// what every launch took from every stream.
// Checked: no data race; every stream's samples arrive at the "device" exactly once and in order whatever the mix of
// pusher threads, push sizes, silent streams and flushes; launches cover ascending, distinct streams.
// Second part (finish_while_pushing): streams are ENDED (nvx_stream_finish, nvx_finish) while their pushers are in the
// middle of calls -- a push call is atomic against the end of its stream (accepted whole, or refused whole with
// NVX_ERR_STATE), no launch ever names an ended stream, and the other streams' pushers never see an error.
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
    for (int i = 0; i < n; i++)                              // (as the real one: a launch that names an ended stream is refused whole)
        if (h->ended[part ? part[i] : i]) { g_bad++; return NVX_ERR_STATE; }
    for (int i = 0; i < n; i++) {
        const int s = part ? part[i] : i;
        if (part && i > 0 && part[i] <= part[i - 1]) g_bad++;
        const uint32_t *row = (const uint32_t *)d_iq + (size_t)s * pitch;
        g_got[s].insert(g_got[s].end(), row, row + (size_t)n_frames * h->frame_in);
        h->g0s[s] += (unsigned long long)(tail_n3 ? tail_n3[i] : n_frames * NVX_FRAME_Y3);       // (the stream has been through so much: nvx_finish asks)
        if (tail_n3) { if (n_frames != 1 || tail_n3[i] < 1 || (size_t)tail_n3[i] * 280 > h->frame_in) g_bad++; h->ended[s] = 1; }
    }
    h->launched++;
    return NVX_OK;
}

struct Rig {                                                 // a push-mode handle on the "device" above
    nvx_handle h;
    std::vector<uint32_t> stage[2], din;
    Rig(int S, int max_frames, int eager)
    {
        h.cfg.push_mode = 1; h.cfg.max_frames = max_frames; h.cfg.device = 0; h.cfg.eager_launch = eager;
        h.n_in = h.n_streams = S;
        h.frame_in = 48 * 1024;                              // 192 KB per frame: pushes above and below the unlocked-copy threshold
        h.stage_cap = (size_t)(max_frames + 1) * h.frame_in;
        din.assign((size_t)S * max_frames * h.frame_in, 0);
        for (int i = 0; i < 2; i++) { stage[i].assign((size_t)S * h.stage_cap, 0); h.h_stage[i] = stage[i].data(); h.set_launch[i].assign(S, 0); }
        h.d_in = din.data();
        h.fill.assign(S, 0); h.cur.assign(S, 0); h.active.assign(S, 1); h.writing.assign(S, 0); h.pushing.assign(S, 0); h.closing.assign(S, 0);
        h.last_push_ns.assign(S, nvx_now_ns());
        h.parity.assign(S, 0); h.g0s.assign(S, 0); h.ended.assign(S, 0); h.stall_ns.assign(S, 2000000000ll);
        g_got.assign(S, {});
    }
};

static uint32_t sample_of(int s, size_t k) { return (uint32_t)k ^ ((uint32_t)s << 28); }
static void fill_samples(std::vector<int16_t> &buf, int s, size_t pos, size_t m)
{
    buf.resize(2 * m);
    for (size_t k = 0; k < m; k++) { const uint32_t v = sample_of(s, pos + k); buf[2 * k] = (int16_t)(v & 0xffff); buf[2 * k + 1] = (int16_t)(v >> 16); }
}

// Streams ended under their pushers (the advisor's finish-while-pushing case, r5).  Four pusher threads push without
// pause; the main thread ends stream 1 alone, a little later streams 0..3 all at once, while the calls are in progress.
// A call is accepted whole or refused whole; what reached the "device" per stream is exactly the accepted samples, in
// order, then zeros to the end of the tail's frame; nothing is staged in an ended stream; nobody but the ended stream's
// own pusher sees an error, and only NVX_ERR_STATE.
static int finish_while_pushing(int eager)
{
    const int S = 4, max_frames = 2;
    Rig rig(S, max_frames, eager);
    nvx_handle &h = rig.h;
    const int bad0 = g_bad.load();
    std::atomic<int> stop{ 0 }, wrong_error{ 0 }, partial_accept{ 0 };
    std::vector<size_t> accepted(S, 0);
    std::vector<std::thread> pushers;
    for (int s = 0; s < S; s++)
        pushers.emplace_back([&, s] {
            std::vector<int16_t> buf;
            unsigned x = 991u * (unsigned)(s + 3);
            size_t pos = 0;
            while (!stop.load()) {
                x = x * 1664525u + 1013904223u;
                const size_t m = (x >> 28) < 6 ? 1 + (x >> 8) % 3000 : 30000 + (x >> 8) % 120000;    // (the big ones wait for launches and copy unlocked)
                fill_samples(buf, s, pos, m);
                size_t took = 0;
                const int rc = nvx_push_iq_partial(&h, s, buf.data(), m, &took);
                if (rc == NVX_OK) { if (took != m) partial_accept++; pos += m; accepted[s] = pos; continue; }
                if (rc != NVX_ERR_STATE) wrong_error++;
                if (took != 0) partial_accept++;              // refused means: nothing of this call was staged
                break;                                        // the stream has ended under this pusher: it stops
            }
        });
    auto leave = [&](int rc) { stop = 1; for (auto &t : pushers) t.join(); return rc; };
    auto reached0 = [&] { std::lock_guard<std::mutex> lk(h.mu); return g_got[0].size(); };
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
    if (nvx_stream_finish(&h, 1) != NVX_OK) return leave(20);
    std::this_thread::sleep_for(std::chrono::milliseconds(30));    // the other three go on launching meanwhile
    const size_t got0_then = reached0();
    std::this_thread::sleep_for(std::chrono::milliseconds(60));
    if (reached0() == got0_then) { fprintf(stderr, "the handle made no progress behind an ended stream\n"); return leave(21); }
    if (nvx_finish(&h) != NVX_OK) return leave(22);
    leave(0);
    for (int s = 0; s < S; s++) {
        const size_t n = accepted[s], tail = n % h.frame_in;
        const size_t want = n - tail + (tail >= 280 ? h.frame_in : 0);
        if (g_got[s].size() != want) { fprintf(stderr, "stream %d: %zu samples reached the device, %zu were accepted (%zu expected)\n", s, g_got[s].size(), n, want); return 23; }
        for (size_t k = 0; k < want; k++)
            if (g_got[s][k] != (k < n ? sample_of(s, k) : 0u)) { fprintf(stderr, "stream %d: sample %zu is wrong\n", s, k); return 24; }
        if (h.fill[s] != 0) { fprintf(stderr, "stream %d: %zu samples staged behind its end\n", s, h.fill[s]); return 25; }
        if (n && !h.ended[s]) return 26;
    }
    if (wrong_error || partial_accept || g_bad.load() != bad0) { fprintf(stderr, "wrong errors %d, calls cut in two %d, launches naming an ended stream %d\n", wrong_error.load(), partial_accept.load(), g_bad.load() - bad0); return 27; }
    printf("finish while pushing (eager %d): accepted %zu %zu %zu %zu\n", eager, accepted[0], accepted[1], accepted[2], accepted[3]);
    return 0;
}

int main()
{
    const int S = 6, max_frames = 2;
    Rig rig(S, max_frames, 0);
    nvx_handle &h = rig.h;

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
                fill_samples(buf, s, pos, m);
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
            if (g_got[s][k] != (k < total ? sample_of(s, k) : 0u)) { fprintf(stderr, "stream %d: sample %zu is wrong\n", s, k); return 4; }
        // every stream has had input: every one is ended, whatever its length modulo the frame, and refuses a push
        if (!h.ended[s] || h.active[s]) { fprintf(stderr, "stream %d: ended %d, active %d after nvx_finish (tail %zu)\n", s, (int)h.ended[s], (int)h.active[s], tail); return 7; }
        int16_t one[2] = { 1, 1 };
        if (nvx_push_iq(&h, s, one, 1) != NVX_ERR_STATE) { fprintf(stderr, "stream %d: push after the end\n", s); return 8; }
    }
    printf("launches %d (partial %d, tails %d), bad %d, errors %d\n", g_launches.load(), g_partial.load(), g_tails.load(), g_bad.load(), errors.load());
    if (g_bad || errors || g_partial == 0 || g_tails != 1) return 5;
    for (int eager = 0; eager < 2; eager++) { const int rc = finish_while_pushing(eager); if (rc) return rc; }
    printf("tsan push ok\n");
    return 0;
}
