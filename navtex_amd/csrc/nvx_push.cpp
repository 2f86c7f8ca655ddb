// nvx_push.cpp -- host-input path: pinned staging sets, hipMemcpyAsync to the device, launches whenever every active
// stream has a whole frame (or one stream cannot wait any longer); the WAV file path on top of it.
#include "nvx_handle.h"

// ------------------------------------------------------- host-input path
// The streams of a handle are independent receivers (receiver/decoder.h:31-60, receiver/nav_b_sm.h:92-114): nothing ties
// them to a common clock.  Normally they advance together -- a launch goes out when every ACTIVE stream has a whole
// frame staged, and covers all of them.  A stream that falls behind (its radio stalled, was unplugged --
// receiver/capt_sched.c:210-212 only prints sdrplay_api_DeviceRemoved -- or simply runs a few ppm slow) does not hold
// the others: when a stream's staging is full, the launch goes out with the streams that HAVE a frame, each from its
// own carried state; the late one joins a later launch and continues bit-exactly from where it stopped.
static int n_whole_frames(const nvx_handle *h, int s) { return (int)(h->fill[s] / h->frame_in); }

// A stream fed directly through nvx_push_* has no capture ring to declare it silent: one that has delivered nothing for
// its stall timeout (cfg.stall_timeout_ms, default 2 s; a capture ring attached to the stream hands its own timeout on:
// nvx_capture_set_stall_timeout; 0 = wait for ever) is not waited for either, so that the healthy streams keep launching
// frame by frame instead of only when their staging is full.  Its next push counts again at once.

// every active stream has a whole frame (and there is at least one that is waited for and has one)
static bool lockstep_ready(const nvx_handle *h)
{
    bool any = false;
    if (h->cfg.eager_launch) {                       // cfg.eager_launch: whoever has a whole frame goes now (a handful of free-running radios)
        for (int s = 0; s < h->n_in; s++) if (!h->ended[s] && h->fill[s] >= h->frame_in) return true;
        return false;
    }
    const int64_t now = nvx_now_ns();
    for (int s = 0; s < h->n_in; s++) {
        if (!h->active[s] || h->ended[s]) continue;      // (an ended stream holds nothing and is waited for by nobody)
        if (h->fill[s] < h->frame_in) {
            if (h->stall_ns[s] > 0 && now - h->last_push_ns[s] > h->stall_ns[s]) continue;       // gone quiet: not waited for
            return false;
        }
        any = true;
    }
    return any;
}

static int submit_part_locked(nvx_handle *h, const std::vector<int> &part, int frames, const int *tail_n3);

// Launch the streams that have at least one whole frame staged, with as many frames as all of THEM have.
static int submit_locked(nvx_handle *h)
{
    std::vector<int> part;
    int frames = h->cfg.max_frames;
    for (int s = 0; s < h->n_in; s++) {
        const int f = n_whole_frames(h, s);
        if (f < 1 || h->ended[s]) continue;          // (an ended stream never vetoes the others' launch: nvx_launch_locked refuses a list that names one)
        part.push_back(s);
        frames = std::min(frames, f);
    }
    if (part.empty()) return NVX_OK;
    return submit_part_locked(h, part, frames, nullptr);
}

// One launch out of the staging sets: `frames` frames of the streams in `part` (ascending).  tail_n3: the launch ENDS
// these streams (nvx_finish) -- each has less than a frame staged, the rest of the frame is zeroed, and tail_n3[i] of the
// launch's 900 S/s samples are real (nvx_launch_locked).
static int submit_part_locked(nvx_handle *h, const std::vector<int> &part, int frames, const int *tail_n3)
{
    const size_t take = (size_t)frames * h->frame_in;
    const size_t dpitch = (size_t)h->cfg.max_frames * h->frame_in;
    // a participant flips to its other staging set: that set must have left the copy engine (it was read by the copy
    // of the stream's previous launch; copies run in launch order on one stream, so one wait covers everything older)
    uint64_t need = 0;
    for (int s : part) need = std::max(need, h->set_launch[h->cur[s] ^ 1][s]);
    if (need > h->copies_synced) {
        // (a result slot reused since then carries a LATER copy's event: waiting for that covers the older one too)
        HIP_TRY(hipEventSynchronize(h->res[(need - 1) % RESULT_SLOTS].copied));
        h->copies_synced = need;
    }
    // d_in is reused by every launch: stream order makes its previous reader finish first (the cascade or the fused
    // wideband kernel on h->stream).  Runs of neighbouring participants that fill the same set go as one 2-D copy.
    hipStream_t cs = h->stream;
    if (tail_n3)                                         // what lies behind a stream's last sample is not signal: zeros, for determinism
        for (int s : part) memset(h->h_stage[h->cur[s]] + (size_t)s * h->stage_cap + h->fill[s], 0, (take - h->fill[s]) * 4);
    for (size_t i = 0; i < part.size();) {
        size_t j = i + 1;
        while (j < part.size() && part[j] == part[j - 1] + 1 && h->cur[part[j]] == h->cur[part[i]]) j++;
        const int s0 = part[i];
        HIP_TRY(hipMemcpy2DAsync(h->d_in + (size_t)s0 * dpitch, dpitch * 4, h->h_stage[h->cur[s0]] + (size_t)s0 * h->stage_cap, h->stage_cap * 4,
                                 take * 4, j - i, hipMemcpyHostToDevice, cs));
        i = j;
    }
    Result &r = h->res[h->launched % RESULT_SLOTS];
    HIP_TRY(hipEventRecord(r.copied, cs));           // (re-recording an event a later wait may still name: see above)
    const uint64_t this_launch = h->launched + 1;
    const bool all = (int)part.size() == h->n_in;
    int rc = nvx_launch_locked(h, h->d_in, dpitch, 0, frames, h->stream, (all && !tail_n3) ? nullptr : part.data(), (all && !tail_n3) ? 0 : (int)part.size(), tail_n3);
    if (rc != NVX_OK) return rc;
    // what was not submitted moves over to the participant's other set, which it fills from now on
    for (int s : part) {
        const int c = h->cur[s], n = c ^ 1;
        h->set_launch[c][s] = this_launch;
        const size_t rest = tail_n3 ? 0 : h->fill[s] - take;
        if (rest) memcpy(h->h_stage[n] + (size_t)s * h->stage_cap, h->h_stage[c] + (size_t)s * h->stage_cap + take, rest * 4);
        h->fill[s] = rest;
        h->cur[s] = (uint8_t)n;
    }
    return NVX_OK;
}

// pushes of at least this many bytes copy into the staging WITHOUT the handle's lock (nvx_handle.h: writing / writers)
#define NVX_UNLOCKED_COPY_BYTES (128u << 10)

// accepted (optional): how many of the n samples were staged -- all of them on NVX_OK, fewer on an error
template <typename F>
static int push_common(nvx_handle *h, int stream, size_t n, F copy_in, size_t *accepted = nullptr)
{
    if (accepted) *accepted = 0;
    if (!h || stream < 0 || stream >= h->n_in) { nvx_set_error("nvx_push: bad stream"); return NVX_ERR_ARG; }
    if (!h->cfg.push_mode) { nvx_set_error("nvx_push: handle was not created with push_mode"); return NVX_ERR_STATE; }
    std::unique_lock<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    // a stream has ONE pusher at a time, for the whole call (the lock is released while this one waits for a launch or
    // copies a large chunk: a second pusher of the same stream must not interleave its chunks with this one's); and a
    // call that arrives while the stream is being ended or restarted (StreamClose, nvx_handle.h) waits for the outcome
    h->wr_cv.wait(lk, [&] { return !h->pushing[stream] && !h->closing[stream]; });
    auto refused = [&]() -> int {                        // looked at whenever the lock has been away
        if (h->poisoned) return nvx_poisoned_error(h);
        if (h->ended[stream]) { nvx_set_error("nvx_push: stream %d has ended (nvx_finish); nvx_stream_reset or nvx_reset starts a new one", stream); return NVX_ERR_STATE; }
        return NVX_OK;
    };
    { int rc = refused(); if (rc != NVX_OK) return rc; }
    h->pushing[stream] = 1;
    struct Release {
        nvx_handle *h; int stream;
        ~Release() { h->pushing[stream] = 0; h->wr_cv.notify_all(); }       // (the handle is locked on every path out of here)
    } release{ h, stream };
    if (n) { h->active[stream] = 1; h->last_push_ns[stream] = nvx_now_ns(); }    // a stream that delivers is (again) one the others wait for
    size_t done = 0;
    // a launch out of the staging sets: with every unlocked copy committed, and only if `still` holds afterwards (another
    // thread may have launched while this one waited)
    auto launch_if = [&](auto still) -> int {
        StagingQuiesce quiet(h, lk);
        return still() ? submit_locked(h) : NVX_OK;
    };
    while (done < n) {
        if (h->quiesce) {                            // somebody is rearranging the staging sets
            h->wr_cv.wait(lk, [&] { return h->quiesce == 0; });
            // (a StreamClose waits for this call to end before it ends or restarts the stream, so `ended` cannot have
            // come up under a push in progress; a launch that failed meanwhile has poisoned the handle)
            int rc = refused();
            if (rc != NVX_OK) { if (accepted) *accepted = done; return rc; }
        }
        size_t room = h->stage_cap - h->fill[stream];
        if (room == 0) {
            // this stream is max_frames + 1 frames ahead of a launch: go with the streams that have a frame
            int rc = launch_if([&] { return h->fill[stream] == h->stage_cap; });
            if (rc != NVX_OK) { if (accepted) *accepted = done; return rc; }
            continue;                                // (the wait above released the lock: look again)
        }
        const size_t m = std::min(room, n - done);
        uint32_t *dst = h->h_stage[h->cur[stream]] + (size_t)stream * h->stage_cap + h->fill[stream];
        if (m * 4 >= NVX_UNLOCKED_COPY_BYTES) {
            // the copy itself needs no lock: nobody moves this stream's fill / cur while writing[stream] is up
            h->writing[stream] = 1; h->writers++;
            lk.unlock();
            copy_in(dst, done, m);
            lk.lock();
            h->writing[stream] = 0; h->writers--;
            h->wr_cv.notify_all();
        } else {
            copy_in(dst, done, m);
        }
        h->fill[stream] += m;
        done += m;
        if (lockstep_ready(h)) {
            int rc = launch_if([&] { return lockstep_ready(h); });
            if (rc != NVX_OK) { if (accepted) *accepted = done; return rc; }
        }
    }
    // Launched work that has finished meanwhile is taken in on the way out (never waits): a caller that only ever pushes --
    // the stream callback on a handle of its own, no ring, no poller -- gets its bits and messages within one callback
    // interval of their completion instead of with the next launch, a frame later.
    if (h->collected != h->launched) {
        int rc = nvx_collect_ready_locked(h);
        if (rc != NVX_OK) { if (accepted) *accepted = n; return rc; }
    }
    if (accepted) *accepted = n;
    return NVX_OK;
}

// the capture ring's consumer: a full staging set is back-pressure there, not an error
int nvx_push_iq_partial(nvx_handle *h, int stream, const int16_t *iq, size_t n, size_t *accepted)
{
    if (n && !iq) { if (accepted) *accepted = 0; nvx_set_error("nvx_push_iq: null samples"); return NVX_ERR_ARG; }
    return push_common(h, stream, n, [&](uint32_t *dst, size_t off, size_t m) { memcpy(dst, iq + 2 * off, m * 4); }, accepted);
}

extern "C" int nvx_push_iq(nvx_handle *h, int stream, const int16_t *iq, size_t n)
{
    if (n && !iq) { nvx_set_error("nvx_push_iq: null samples"); return NVX_ERR_ARG; }
    return push_common(h, stream, n, [&](uint32_t *dst, size_t off, size_t m) { memcpy(dst, iq + 2 * off, m * 4); });
}

extern "C" int nvx_push_planar(nvx_handle *h, int stream, const int16_t *xi, const int16_t *xq, size_t n)
{
    if (n && (!xi || !xq)) { nvx_set_error("nvx_push_planar: null samples"); return NVX_ERR_ARG; }
    return push_common(h, stream, n, [&](uint32_t *dst, size_t off, size_t m) {
        for (size_t k = 0; k < m; k++)                     // interleave as capt_sched.c:120-129 does
            dst[k] = (uint32_t)(uint16_t)xi[off + k] | ((uint32_t)(uint16_t)xq[off + k] << 16);
    });
}

// whatever is staged in whole frames goes out, stream by stream as far as each has got (staging quiesced by the caller)
static int submit_whole_frames_locked(nvx_handle *h)
{
    for (;;) {
        const uint64_t before = h->launched;
        int rc = submit_locked(h);
        if (rc != NVX_OK) return rc;
        if (h->launched == before) return NVX_OK;
    }
}

extern "C" int nvx_flush(nvx_handle *h)
{
    if (!h) return NVX_ERR_ARG;
    std::unique_lock<std::mutex> lk(h->mu);
    if (h->poisoned) return nvx_poisoned_error(h);
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->cfg.push_mode) {
        StagingQuiesce quiet(h, lk);                 // pushes in flight on other threads commit first
        int rc = submit_whole_frames_locked(h);
        if (rc != NVX_OK) return rc;
    }
    return nvx_collect_locked(h);
}

// End of input (header: nvx_finish).  The reference's loop hands every sample it is given to sample_in_1 and stops
// (receiver/capt_sched.c:509-513); its decoder has then seen floor(n / 280) samples at 900 S/s (y3[k] is complete
// with input sample 280 k + 279) and decided the bits those samples decide (receiver/decoder.C:73-137) -- no more.  So:
// whole frames go out as in nvx_flush; then ONE launch carries the streams that still hold a partial frame, each with its
// true count of 900 S/s samples: the cascade runs the frame (zeros behind the last sample; every y3[k] with k below the
// count depends on real samples only), the demodulator stops at the count.  stream < 0: every stream.
static int finish_locked(nvx_handle *h, std::unique_lock<std::mutex> &lk, int stream)
{
    if (h->poisoned) return nvx_poisoned_error(h);
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->cfg.push_mode) {
        // A push call is atomic against the end of its stream: the calls in progress on the streams that are ending run
        // to their end first, calls that arrive meanwhile wait and then find the stream ended (nvx_handle.h: StreamClose).
        StreamClose closing(h, lk, stream < 0 ? 0 : stream, stream < 0 ? h->n_in : stream + 1);
        if (h->poisoned) return nvx_poisoned_error(h);
        StagingQuiesce quiet(h, lk);
        int rc = submit_whole_frames_locked(h);
        if (rc != NVX_OK) return rc;
        const size_t per_y3 = (h->cfg.raw_rate || h->cfg.wideband) ? (size_t)(280 * NVX_DECIM0) : (size_t)280;     // input samples per 900 S/s sample
        std::vector<int> part, n3;
        for (int s = (stream < 0 ? 0 : stream); s < (stream < 0 ? h->n_in : stream + 1); s++) {
            if (h->ended[s]) continue;
            // One rule for every length (r6; before, a stream that stopped exactly on a frame boundary stayed live, so whether
            // a later push was accepted depended on the input's length modulo the frame): the end of the input ENDS the
            // stream.  On a frame boundary nothing is left to launch; a stream that has had no input at all has no end.
            if (h->fill[s] == 0) {
                if (h->g0s[s] > 0) { h->ended[s] = 1; h->active[s] = 0; }
                continue;
            }
            const int t = (int)(h->fill[s] / per_y3);
            if (t > 0) { part.push_back(s); n3.push_back(t); }
            else { h->fill[s] = 0; h->ended[s] = 1; h->active[s] = 0; }  // too short for one more 900 S/s sample: nothing to decode
        }
        if (!part.empty()) {
            rc = submit_part_locked(h, part, 1, n3.data());
            if (rc != NVX_OK) return rc;
            for (int s : part) h->active[s] = 0;                         // nobody waits for an ended stream
        }
    }
    return nvx_collect_locked(h);
}

extern "C" int nvx_finish(nvx_handle *h)
{
    if (!h) return NVX_ERR_ARG;
    std::unique_lock<std::mutex> lk(h->mu);
    return finish_locked(h, lk, -1);
}

extern "C" int nvx_stream_finish(nvx_handle *h, int stream)
{
    if (!h || stream < 0 || stream >= h->n_in) { nvx_set_error("nvx_stream_finish: bad stream"); return NVX_ERR_ARG; }
    std::unique_lock<std::mutex> lk(h->mu);
    return finish_locked(h, lk, stream);
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
    // the file has ended: its last, partial frame runs at its true length (no padding: the reference's loop stops with the
    // last sample, receiver/capt_sched.c:509-513)
    rc = nvx_stream_finish(h, stream);
    if (rc != NVX_OK) return rc;
    // (whatever the file's length its stream is ended now -- nvx_stream_finish ends a stream that has had input -- so the next
    // file on this stream needs its nvx_stream_reset and is never silently decoded as the continuation of this one; an
    // empty file on a fresh stream ends nothing)
    return (int)((total + h->frame_in - 1) / h->frame_in);
}

// ------------------------------------------------------------------ stream activity
extern "C" int nvx_stream_set_active(nvx_handle *h, int stream, int active)
{
    if (!h || stream < 0 || stream >= h->n_in) { nvx_set_error("nvx_stream_set_active: bad stream"); return NVX_ERR_ARG; }
    if (!h->cfg.push_mode) { nvx_set_error("nvx_stream_set_active: handle was not created with push_mode"); return NVX_ERR_STATE; }
    std::unique_lock<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->ended[stream]) active = 0;                    // an ended stream (nvx_finish) delivers nothing more: nobody waits for it
    h->active[stream] = active ? 1 : 0;
    // the others may have been waiting for exactly this stream; and when the LAST active stream goes silent, whole frames
    // that were waiting for company go out now rather than when a radio comes back
    auto due = [&] { return lockstep_ready(h) || std::none_of(h->active.begin(), h->active.end(), [](uint8_t a) { return a != 0; }); };
    if (!active && due()) {
        StagingQuiesce quiet(h, lk);
        if (due()) return submit_locked(h);
    }
    return NVX_OK;
}

extern "C" int nvx_stream_stats(nvx_handle *h, int stream, int *active, uint64_t *frames_done, uint64_t *partial_launches)
{
    if (!h || stream < 0 || stream >= h->n_in) { nvx_set_error("nvx_stream_stats: bad stream"); return NVX_ERR_ARG; }
    std::lock_guard<std::mutex> lk(h->mu);
    if (active) *active = h->active.empty() ? 1 : (int)h->active[stream];
    if (frames_done) *frames_done = h->g0s[stream] / NVX_FRAME_Y3;
    if (partial_launches) *partial_launches = h->partial_launches;
    return NVX_OK;
}
