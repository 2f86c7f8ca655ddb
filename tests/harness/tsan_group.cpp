// ThreadSanitizer driver for the multi-device group (navtex_amd/csrc/nvx_group.cpp) without a GPU: the member workers,
// their job queues, the parked messages and the index map run for real; the per-device handle behind them is replaced
// by a stand-in that keeps the real one's locking contract (every entry point takes the handle's mutex, messages are
// delivered through cfg.on_message with the mutex held) and produces a known message stream.
// Checked: no data race; every message is delivered exactly once, with the GLOBAL stream id, in member order within a
// fetch; errors of a member surface at the next fetch with its text; destroy joins everything.
// Built by tests/test_sanitizers.py with -fsanitize=thread.
#include "nvx_handle.h"

#include <chrono>
#include <map>

static thread_local std::string t_err;
extern "C" void nvx_set_error(const char *fmt, ...) { char b[512]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); t_err = b; }
extern "C" const char *nvx_last_error(void) { return t_err.c_str(); }
extern "C" int add_message(char *, char *, int) { return 0; }
hipError_t hipDeviceGetPCIBusId(char *, int, int) { return hipErrorInvalidDevice; }      // no sysfs walk in the harness

// ---- the stand-in handle: "launch k of member m" completes one message per stream 100 us later
struct Stand { uint64_t launched = 0, delivered = 0; int fail_at = -1; };
static std::mutex g_reg_mu;
static std::map<nvx_handle *, Stand> g_reg;
static Stand &stand(nvx_handle *h) { std::lock_guard<std::mutex> lk(g_reg_mu); return g_reg[h]; }

extern "C" int nvx_create(const nvx_config *cfg, nvx_handle **out)
{
    nvx_handle *h = new nvx_handle();
    h->cfg = *cfg; h->n_in = h->n_streams = cfg->n_streams;
    stand(h);
    *out = h;
    return NVX_OK;
}
extern "C" void nvx_destroy(nvx_handle *h) { { std::lock_guard<std::mutex> lk(g_reg_mu); g_reg.erase(h); } delete h; }
static void deliver(nvx_handle *h, Stand &s, uint64_t upto)          // handle locked
{
    for (; s.delivered < upto; s.delivered++)
        for (int st = 0; st < h->n_streams; st++) {
            char text[64]; snprintf(text, sizeof text, "launch %llu", (unsigned long long)s.delivered);
            h->cfg.on_message(h->cfg.user, st, "AB01", text, 518);
        }
}
extern "C" int nvx_process_resident(nvx_handle *h, const void *, size_t, size_t, int, void *)
{
    std::lock_guard<std::mutex> lk(h->mu);
    Stand &s = stand(h);
    std::this_thread::sleep_for(std::chrono::microseconds(100));
    if (s.fail_at >= 0 && (int)s.launched == s.fail_at) { s.fail_at = -1; nvx_set_error("stand-in: launch %llu failed", (unsigned long long)s.launched); return NVX_ERR_HIP; }
    s.launched++;
    if (s.launched - s.delivered > 2) deliver(h, s, s.delivered + 1);      // a full ring takes in its oldest result, on the worker's thread
    return NVX_OK;
}
extern "C" int nvx_fetch_bits(nvx_handle *h) { std::lock_guard<std::mutex> lk(h->mu); Stand &s = stand(h); deliver(h, s, s.launched); return NVX_OK; }
extern "C" int nvx_flush(nvx_handle *h) { return nvx_fetch_bits(h); }
extern "C" int nvx_finish(nvx_handle *h) { return nvx_fetch_bits(h); }
extern "C" int nvx_reset(nvx_handle *h) { std::lock_guard<std::mutex> lk(h->mu); Stand &s = stand(h); s.launched = s.delivered = 0; return NVX_OK; }
static std::atomic<int> g_in_push{ 0 }, g_overlapped{ 0 };                  // how many pushes are inside a handle at once
extern "C" int nvx_push_iq(nvx_handle *h, int stream, const int16_t *, size_t)
{
    std::lock_guard<std::mutex> lk(h->mu);
    if (stream < 0 || stream >= h->n_streams) return NVX_ERR_ARG;
    // two members' pushes in flight together = the group does not serialise them
    if (g_in_push.fetch_add(1) > 0) g_overlapped.fetch_add(1);
    std::this_thread::sleep_for(std::chrono::microseconds(200));      // "memcpy into pinned staging"
    g_in_push.fetch_sub(1);
    h->cfg.on_message(h->cfg.user, stream, "PU01", "pushed", 490);      // a push that completes a message on the caller's thread
    return NVX_OK;
}
extern "C" size_t nvx_poll_bits(nvx_handle *h, int, int, char *, size_t) { std::lock_guard<std::mutex> lk(h->mu); return (size_t)stand(h).delivered; }
extern "C" size_t nvx_bit_count(nvx_handle *h, int, int) { std::lock_guard<std::mutex> lk(h->mu); return (size_t)stand(h).delivered; }

// ---- the user's sink (runs on the API caller's thread, member after member)
struct Seen { std::vector<int> streams; uint64_t launches = 0, pushes = 0; int bad = 0; };
static void sink(void *user, int stream, const char *bbbb, const char *, int freq)
{
    Seen *s = (Seen *)user;
    if (freq == 518 && !strcmp(bbbb, "AB01")) s->launches++; else if (freq == 490 && !strcmp(bbbb, "PU01")) s->pushes++; else s->bad++;
    s->streams.push_back(stream);
}

int main()
{
    const int n_members = 4, total = 4 * 8 + 2;                 // 34 streams over 4 members: 9, 9, 8, 8
    int devices[n_members] = { 0, 0, 0, 0 };
    Seen seen;
    nvx_config cfg{};
    cfg.struct_size = sizeof cfg;
    cfg.n_streams = total; cfg.max_frames = 1; cfg.on_message = sink; cfg.user = &seen;
    nvx_group *g = nullptr;
    if (nvx_group_create(devices, n_members, &cfg, &g) != NVX_OK || nvx_group_size(g) != n_members) return 2;
    int first = -1, count = 0, sum = 0;
    for (int m = 0; m < n_members; m++) { nvx_group_member(g, m, nullptr, &first, &count, nullptr); if (first != sum) return 3; sum += count; }
    if (sum != total || nvx_group_member_of(g, 8) != 0 || nvx_group_member_of(g, 9) != 1 || nvx_group_member_of(g, total - 1) != 3 || nvx_group_member_of(g, total) != -1) return 4;

    const void *bufs[n_members] = { &seen, &seen, &seen, &seen };      // never dereferenced by the stand-in
    std::atomic<bool> stop{ false };
    std::atomic<int> push_errors{ 0 };
    std::thread poller([&] {                   // poll / count from another thread while launches and fetches run
        unsigned x = 1;
        while (!stop) { x = x * 1664525u + 1013904223u; char b[8]; nvx_group_poll_bits(g, (int)(x >> 8) % total, 0, b, sizeof b); nvx_group_bit_count(g, (int)(x >> 16) % total, 1); }
    });
    std::thread pusher([&] {                   // a capture thread pushing into streams of every member
        for (int i = 0; i < 400; i++) { int16_t iq[2] = { 0, 0 }; if (nvx_group_push_iq(g, (i * 7) % total, iq, 1) != NVX_OK) push_errors++; }
    });
    // ... and one capture thread PER MEMBER, each feeding only its own member's streams: nvx_group_push_iq holds no
    // group-wide lock, so these run side by side (and beside the launches, fetches and the deliveries of parked messages)
    std::vector<std::thread> member_pushers;
    for (int m = 0; m < n_members; m++)
        member_pushers.emplace_back([&, m] {
            int f = 0, c = 0; nvx_group_member(g, m, nullptr, &f, &c, nullptr);
            for (int i = 0; i < 150; i++) { int16_t iq[2] = { 0, 0 }; if (nvx_group_push_iq(g, f + i % c, iq, 1) != NVX_OK) push_errors++; }
        });
    const int rounds = 120, per_round = 3;
    int rc_fail = 0;
    for (int r = 0; r < rounds; r++) {
        for (int k = 0; k < per_round; k++) if (nvx_group_process_resident(g, bufs, 0, 0, 1) != NVX_OK) return 5;
        const size_t at = seen.streams.size();
        if (nvx_group_fetch_bits(g) != NVX_OK) rc_fail++;
        // within one fetch the launch messages come member after member: global ids never decrease across members' blocks
        // (pushed messages are parked with their member too, so they sit inside their member's block)
        int last_member = 0;
        for (size_t i = at; i < seen.streams.size(); i++) {
            const int m = nvx_group_member_of(g, seen.streams[i]);
            if (m < last_member) seen.bad++;
            last_member = m;
        }
    }
    pusher.join();
    for (auto &t : member_pushers) t.join();
    if (nvx_group_flush(g) != NVX_OK) return 6;
    stop = true; poller.join();

    // a failing member: the error surfaces at the fetch, with the member named, and the group keeps working
    { nvx_handle *h1 = nullptr; nvx_group_member(g, 1, nullptr, nullptr, nullptr, &h1); std::lock_guard<std::mutex> lk(h1->mu); Stand &s = stand(h1); s.fail_at = (int)s.launched; }
    nvx_group_process_resident(g, bufs, 0, 0, 1);
    const int rc = nvx_group_fetch_bits(g);
    const bool named = strstr(nvx_last_error(), "member 1") && strstr(nvx_last_error(), "stand-in");
    if (rc != NVX_ERR_HIP || !named) { fprintf(stderr, "error path: rc %d, text '%s'\n", rc, nvx_last_error()); return 7; }
    if (nvx_group_process_resident(g, bufs, 0, 0, 1) != NVX_OK || nvx_group_fetch_bits(g) != NVX_OK) return 8;
    if (nvx_group_reset(g) != NVX_OK) return 9;
    nvx_group_destroy(g);

    // launches: rounds * per_round on 4 members + (1 failed on member 1 -> 3 members) + 1 more on all; one message per stream each
    const uint64_t want = (uint64_t)rounds * per_round * total + (uint64_t)(total - 9) + (uint64_t)total;
    const uint64_t want_pushes = 400 + 150 * (uint64_t)n_members;
    printf("launch messages %llu (want %llu), pushed %llu (want %llu), overlapping pushes %d, bad %d, fetch errors %d\n",
           (unsigned long long)seen.launches, (unsigned long long)want, (unsigned long long)seen.pushes, (unsigned long long)want_pushes,
           g_overlapped.load(), seen.bad, rc_fail);
    if (seen.launches != want || seen.pushes != want_pushes || seen.bad || rc_fail || push_errors) return 10;
    if (g_overlapped.load() == 0) return 11;       // pushes into different members really ran at the same time
    printf("tsan group ok\n");
    return 0;
}
