// nvx_group.cpp -- several GPUs behind one object (header section C').
//
// NAVTEX chains share no state (receiver/decoder.h:31-60, receiver/nav_b_sm.h:92-114; FIR1 and mixer state per
// stream: receiver/fir1cpp.C:57-60, receiver/fir2cpp.C:74-83), so the streams shard one contiguous subset per device
// and there is NO exchange between devices: a group is n handles, n host threads and an index map.
//   member m owns global streams first(m) .. first(m) + count(m) - 1
//   every member has a worker thread (bound to its device's NUMA node) that issues its launches and runs its
//   collect -- the per-device host work (kernel launches, bit unpacking, character layers) of all devices overlaps
//   messages are parked per member while the workers run and handed to the user afterwards by the calling thread,
//   member after member: the order one handle of all streams would produce, whatever the devices' timing
#include "nvx_handle.h"

#include <deque>
#include <functional>
#include <sched.h>

struct GroupMsg { int stream; std::string bbbb, text; int freq; };

struct Member {
    nvx_group *g = nullptr;
    int index = 0, device = 0, first = 0, count = 0;
    nvx_handle *h = nullptr;
    std::vector<uint8_t> masks;
    std::vector<int> labels;
    std::mutex parked_mu;                         // parked: written by whoever runs the member's collect -- its worker, or a thread
    std::vector<GroupMsg> parked;                 // pushing into one of its streams -- and emptied by the thread that delivers
    // worker
    std::thread worker;
    std::mutex mu; std::condition_variable cv;
    std::deque<std::function<int()>> jobs;
    bool quit = false, busy = false;
    int rc = NVX_OK; std::string err;             // first failure since the last fetch / flush
};

struct nvx_group {
    nvx_config cfg{};
    int total = 0;                                // input streams (wideband: 2.016 MS/s inputs)
    int per_in = 1;                               // decoded streams per input stream: 8 in wideband mode (stream 8 * w + k)
    std::vector<Member *> members;
    std::mutex api_mu;                            // one CONTROL call (reset, launch, fetch, flush) at a time.  nvx_group_push_iq does
                                                  // not take it: capture threads feeding different members run side by side, each
                                                  // serialised only by its member's own handle lock
};

static void member_on_message(void *user, int stream, const char *bbbb, const char *message, int freq)
{
    Member *m = (Member *)user;
    std::lock_guard<std::mutex> lk(m->parked_mu);
    m->parked.push_back(GroupMsg{ m->g->per_in * m->first + stream, bbbb, message, freq });
}

static void member_loop(Member *m)
{
    static const bool numa = !(getenv("NVX_GROUP_NUMA") && atoi(getenv("NVX_GROUP_NUMA")) == 0);
    if (numa) nvx_bind_thread_to_device(m->device);
    for (;;) {
        std::function<int()> job;
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv.wait(lk, [&] { return m->quit || !m->jobs.empty(); });
            if (m->jobs.empty()) return;          // quit, nothing left
            job = std::move(m->jobs.front()); m->jobs.pop_front();
            m->busy = true;
        }
        const int rc = job();
        {
            std::lock_guard<std::mutex> lk(m->mu);
            if (rc != NVX_OK && m->rc == NVX_OK) { m->rc = rc; m->err = nvx_last_error(); }   // the error text is thread-local
            m->busy = false;
        }
        m->cv.notify_all();
    }
}

static void member_post(Member *m, std::function<int()> job)
{
    { std::lock_guard<std::mutex> lk(m->mu); m->jobs.push_back(std::move(job)); }
    m->cv.notify_all();
}

// wait until the member's queue has drained; returns (and clears) its first error
static int member_join(Member *m)
{
    std::unique_lock<std::mutex> lk(m->mu);
    m->cv.wait(lk, [&] { return m->jobs.empty() && !m->busy; });
    const int rc = m->rc;
    if (rc != NVX_OK) nvx_set_error("group member %d (device %d): %s", m->index, m->device, m->err.c_str());
    m->rc = NVX_OK; m->err.clear();
    return rc;
}

static void deliver_parked(nvx_group *g)
{
    for (Member *m : g->members) {
        std::vector<GroupMsg> batch;
        { std::lock_guard<std::mutex> lk(m->parked_mu); batch.swap(m->parked); }     // a pushing thread may park more meanwhile: next delivery
        for (auto &msg : batch) {
            if (g->cfg.on_message) g->cfg.on_message(g->cfg.user, msg.stream, msg.bbbb.c_str(), msg.text.c_str(), msg.freq);
            else add_message((char *)msg.bbbb.c_str(), (char *)msg.text.c_str(), msg.freq);          // receiver/message_store.h:7
        }
    }
}

// every member's queue drained; the first error wins, messages are delivered either way
static int group_join(nvx_group *g)
{
    int rc = NVX_OK;
    std::string first;
    for (Member *m : g->members) {
        const int r = member_join(m);
        if (r != NVX_OK && rc == NVX_OK) { rc = r; first = nvx_last_error(); }
    }
    deliver_parked(g);
    if (rc != NVX_OK) nvx_set_error("%s", first.c_str());
    return rc;
}

extern "C" void nvx_group_destroy(nvx_group *g)
{
    if (!g) return;
    for (Member *m : g->members) {
        if (m->worker.joinable()) {
            { std::lock_guard<std::mutex> lk(m->mu); m->quit = true; }
            m->cv.notify_all();
            m->worker.join();
        }
        if (m->h) nvx_destroy(m->h);
        delete m;
    }
    delete g;
}

extern "C" int nvx_group_create(const int *devices, int n_members, const nvx_config *cfg, nvx_group **out)
{
    if (!devices || !cfg || !out) { nvx_set_error("nvx_group_create: null argument"); return NVX_ERR_ARG; }
    *out = nullptr;
    if (cfg->struct_size != sizeof(nvx_config)) {          // (as nvx_create: checked before anything else of the struct is read)
        nvx_set_error("nvx_group_create: nvx_config.struct_size is %u, this library's nvx_config has %zu bytes (ABI %d)", cfg->struct_size, sizeof(nvx_config), NVX_ABI_VERSION);
        return NVX_ERR_ARG;
    }
    if (n_members < 1 || cfg->n_streams < n_members) {
        nvx_set_error("nvx_group_create: bad argument (need at least one stream per member)");
        return NVX_ERR_ARG;
    }
    nvx_group *g = new nvx_group();
    g->cfg = *cfg; g->total = cfg->n_streams;
    // streams addressed by the caller: wideband inputs or plain streams; masks / labels are per DECODED stream
    const int per_in = g->per_in = cfg->wideband ? NVX_WB_SUBBANDS : 1;
    const int base = g->total / n_members, extra = g->total % n_members;
    int hw = (int)std::thread::hardware_concurrency(); if (hw < 1) hw = 1;
    int first = 0;
    for (int i = 0; i < n_members; i++) {
        Member *m = new Member();
        g->members.push_back(m);
        m->g = g; m->index = i; m->device = devices[i]; m->first = first; m->count = base + (i < extra ? 1 : 0);
        first += m->count;
        nvx_config c = *cfg;
        c.device = m->device; c.n_streams = m->count;
        if (cfg->chain_masks) { m->masks.assign(cfg->chain_masks + (size_t)m->first * per_in, cfg->chain_masks + (size_t)(m->first + m->count) * per_in); c.chain_masks = m->masks.data(); }
        if (cfg->labels) { m->labels.assign(cfg->labels + (size_t)m->first * per_in * 2, cfg->labels + (size_t)(m->first + m->count) * per_in * 2); c.labels = m->labels.data(); }
        c.on_message = member_on_message; c.user = m;
        if (c.host_threads <= 0) c.host_threads = std::max(1, std::min(16, hw / n_members));     // each member its share of the cores
        const int rc = nvx_create(&c, &m->h);
        if (rc != NVX_OK) { nvx_group_destroy(g); return rc; }
        m->worker = std::thread(member_loop, m);
    }
    *out = g;
    return NVX_OK;
}

extern "C" int nvx_group_size(const nvx_group *g) { return g ? (int)g->members.size() : 0; }

extern "C" int nvx_group_member(nvx_group *g, int mi, int *device, int *first_stream, int *n_streams, nvx_handle **h)
{
    if (!g || mi < 0 || mi >= (int)g->members.size()) { nvx_set_error("nvx_group_member: no member %d", mi); return NVX_ERR_ARG; }
    Member *m = g->members[mi];
    if (device) *device = m->device;
    if (first_stream) *first_stream = m->first;
    if (n_streams) *n_streams = m->count;
    if (h) *h = m->h;
    return NVX_OK;
}

extern "C" int nvx_group_member_of(const nvx_group *g, int s)
{
    if (!g || s < 0 || s >= g->total) return -1;
    // first(m) = m * base + min(m, extra): the first `extra` members hold base + 1 streams
    const int n = (int)g->members.size(), base = g->total / n, extra = g->total % n;
    const int split = extra * (base + 1);
    return s < split ? s / (base + 1) : extra + (s - split) / base;
}

extern "C" int nvx_group_reset(nvx_group *g)
{
    if (!g) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->api_mu);
    for (Member *m : g->members) member_post(m, [m] { return nvx_reset(m->h); });
    int rc = NVX_OK;
    for (Member *m : g->members) { const int r = member_join(m); if (r != NVX_OK && rc == NVX_OK) rc = r; }
    for (Member *m : g->members) { std::lock_guard<std::mutex> pl(m->parked_mu); m->parked.clear(); }
    return rc;
}

extern "C" int nvx_group_process_resident(nvx_group *g, const void *const *d_iq, size_t pitch, size_t first_frame, int n_frames)
{
    if (!g || !d_iq) { nvx_set_error("nvx_group_process_resident: null argument"); return NVX_ERR_ARG; }
    std::lock_guard<std::mutex> lk(g->api_mu);
    for (size_t i = 0; i < g->members.size(); i++) if (!d_iq[i]) { nvx_set_error("nvx_group_process_resident: member %zu has no buffer", i); return NVX_ERR_ARG; }
    for (size_t i = 0; i < g->members.size(); i++) {
        Member *m = g->members[i];
        const void *p = d_iq[i];
        member_post(m, [m, p, pitch, first_frame, n_frames] { return nvx_process_resident(m->h, p, pitch, first_frame, n_frames, nullptr); });
    }
    return NVX_OK;
}

extern "C" int nvx_group_fetch_bits(nvx_group *g)
{
    if (!g) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->api_mu);
    for (Member *m : g->members) member_post(m, [m] { return nvx_fetch_bits(m->h); });
    return group_join(g);
}

extern "C" int nvx_group_push_iq(nvx_group *g, int s, const int16_t *iq, size_t n)
{
    const int mi = nvx_group_member_of(g, s);
    if (mi < 0) { nvx_set_error("nvx_group_push_iq: no stream %d", s); return NVX_ERR_ARG; }
    Member *m = g->members[mi];
    // On the caller's thread (a capture thread owns its stream) and WITHOUT the group's lock: the member's handle locks
    // itself, so N capture threads feeding N members copy into N pinned staging areas at once -- under one group-wide
    // lock the host-fed rate of eight GPUs was one thread's memcpy rate.  Messages a launch completes meanwhile stay parked.
    return nvx_push_iq(m->h, s - m->first, iq, n);
}

extern "C" int nvx_group_flush(nvx_group *g)
{
    if (!g) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->api_mu);
    for (Member *m : g->members) member_post(m, [m] { return nvx_flush(m->h); });
    return group_join(g);
}

extern "C" int nvx_group_finish(nvx_group *g)
{
    if (!g) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->api_mu);
    for (Member *m : g->members) member_post(m, [m] { return nvx_finish(m->h); });
    return group_join(g);
}

// decoded streams: in wideband mode global decoded stream 8 * w + k belongs to input stream w
extern "C" size_t nvx_group_poll_bits(nvx_group *g, int s, int chain, char *out, size_t cap)
{
    if (!g || s < 0) return 0;
    const int mi = nvx_group_member_of(g, s / g->per_in);
    if (mi < 0) return 0;
    Member *m = g->members[mi];
    return nvx_poll_bits(m->h, s - g->per_in * m->first, chain, out, cap);
}

extern "C" size_t nvx_group_bit_count(nvx_group *g, int s, int chain)
{
    if (!g || s < 0) return 0;
    const int mi = nvx_group_member_of(g, s / g->per_in);
    if (mi < 0) return 0;
    Member *m = g->members[mi];
    return nvx_bit_count(m->h, s - g->per_in * m->first, chain);
}

// ------------------------------------------------------------------ NUMA placement
// cpulist syntax of sysfs: "0-15,128-143"
static int parse_cpulist(const char *s, cpu_set_t *set)
{
    CPU_ZERO(set);
    int n = 0;
    while (*s) {
        char *end = nullptr;
        long a = strtol(s, &end, 10);
        if (end == s) break;
        long b = a;
        if (*end == '-') { s = end + 1; b = strtol(s, &end, 10); if (end == s) break; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0) { CPU_SET((int)c, set); n++; }
        s = end;
        while (*s == ',' || *s == ' ' || *s == '\n') s++;
    }
    return n;
}

extern "C" int nvx_bind_thread_to_device(int device)
{
    char bdf[64] = "";
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess) { nvx_set_error("nvx_bind_thread_to_device: no PCI bus id for device %d", device); return NVX_ERR_NODEV; }
    for (char *p = bdf; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');      // sysfs spells it in lower case
    char path[160], buf[4096];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!ok) return 0;
    cpu_set_t want, have, both;
    if (parse_cpulist(buf, &want) == 0) return 0;
    if (sched_getaffinity(0, sizeof have, &have) != 0) return 0;
    CPU_AND(&both, &want, &have);                     // never widen what the container allows
    const int n = CPU_COUNT(&both);
    if (n == 0) return 0;
    if (sched_setaffinity(0, sizeof both, &both) != 0) return 0;
    return n;
}
