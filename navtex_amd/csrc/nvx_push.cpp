// nvx_push.cpp -- host-input path: pinned staging sets, hipMemcpyAsync to the device, launches
// whenever every stream has a whole frame; the WAV file path on top of it.
#include "nvx_handle.h"

// ------------------------------------------------------- host-input path
// Submit the largest common whole-frame prefix of the current staging set.
static int submit_locked(nvx_handle *h)
{
    size_t minfill = *std::min_element(h->fill.begin(), h->fill.end());
    int frames = (int)std::min<size_t>(minfill / h->frame_in, (size_t)h->cfg.max_frames);
    if (frames < 1) return NVX_OK;
    const int cur = h->cur, nxt = cur ^ 1;
    const size_t take = (size_t)frames * h->frame_in;
    const size_t dpitch = (size_t)h->cfg.max_frames * h->frame_in;
    // the other staging set must have left the copy engine before it is refilled
    if (h->stage_busy[nxt]) { HIP_TRY(hipEventSynchronize(h->stage_free[nxt])); h->stage_busy[nxt] = false; }
    // d_in is reused by every launch: stream order makes its previous reader finish first (the cascade or the fused
    // wideband kernel on h->stream; in the two-kernel wideband form the channeliser on stream3, which is why the copy
    // goes there then)
    const bool on_stream3 = h->cfg.wideband && !nvx_wb_fused();
    hipStream_t cs = on_stream3 ? h->stream3 : h->stream;
    HIP_TRY(hipMemcpy2DAsync(h->d_in, dpitch * 4, h->h_stage[cur], h->stage_cap * 4, take * 4, (size_t)h->n_in,
                             hipMemcpyHostToDevice, cs));
    HIP_TRY(hipEventRecord(h->stage_free[cur], cs));
    h->stage_busy[cur] = true;
    int rc = nvx_launch_locked(h, h->d_in, dpitch, 0, frames, h->stream, on_stream3);
    if (rc != NVX_OK) return rc;
    // carry what was not submitted over to the other set
    for (int s = 0; s < h->n_in; s++) {
        size_t rest = h->fill[s] - take;
        if (rest) memcpy(h->h_stage[nxt] + (size_t)s * h->stage_cap, h->h_stage[cur] + (size_t)s * h->stage_cap + take, rest * 4);
        h->fill[s] = rest;
    }
    h->cur = nxt;
    return NVX_OK;
}

// accepted (optional): how many of the n samples were staged -- all of them on NVX_OK, fewer on NVX_ERR_FULL
template <typename F>
static int push_common(nvx_handle *h, int stream, size_t n, F copy_in, size_t *accepted = nullptr)
{
    if (accepted) *accepted = 0;
    if (!h || stream < 0 || stream >= h->n_in) { nvx_set_error("nvx_push: bad stream"); return NVX_ERR_ARG; }
    if (!h->cfg.push_mode) { nvx_set_error("nvx_push: handle was not created with push_mode"); return NVX_ERR_STATE; }
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    size_t done = 0;
    while (done < n) {
        size_t room = h->stage_cap - h->fill[stream];
        if (room == 0) {
            int rc = submit_locked(h);
            if (rc != NVX_OK) return rc;
            room = h->stage_cap - h->fill[stream];
            if (room == 0) {
                nvx_set_error("stream %d is a whole staging buffer ahead of the slowest stream", stream);
                if (accepted) *accepted = done;
                return NVX_ERR_FULL;
            }
        }
        size_t m = std::min(room, n - done);
        copy_in(h->h_stage[h->cur] + (size_t)stream * h->stage_cap + h->fill[stream], done, m);
        h->fill[stream] += m;
        done += m;
        size_t minfill = *std::min_element(h->fill.begin(), h->fill.end());
        if (minfill >= h->frame_in) { int rc = submit_locked(h); if (rc != NVX_OK) { if (accepted) *accepted = done; return rc; } }
    }
    if (accepted) *accepted = n;
    return NVX_OK;
}

// the capture ring's consumer: a full staging set is back-pressure there, not an error
int nvx_push_iq_partial(nvx_handle *h, int stream, const int16_t *iq, size_t n, size_t *accepted)
{
    return push_common(h, stream, n, [&](uint32_t *dst, size_t off, size_t m) { memcpy(dst, iq + 2 * off, m * 4); }, accepted);
}

extern "C" int nvx_push_iq(nvx_handle *h, int stream, const int16_t *iq, size_t n)
{
    return push_common(h, stream, n, [&](uint32_t *dst, size_t off, size_t m) { memcpy(dst, iq + 2 * off, m * 4); });
}

extern "C" int nvx_push_planar(nvx_handle *h, int stream, const int16_t *xi, const int16_t *xq, size_t n)
{
    return push_common(h, stream, n, [&](uint32_t *dst, size_t off, size_t m) {
        for (size_t k = 0; k < m; k++)                     // interleave as capt_sched.c:120-129 does
            dst[k] = (uint32_t)(uint16_t)xi[off + k] | ((uint32_t)(uint16_t)xq[off + k] << 16);
    });
}

extern "C" int nvx_flush(nvx_handle *h)
{
    if (!h) return NVX_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->cfg.push_mode) {
        for (;;) {
            size_t minfill = *std::min_element(h->fill.begin(), h->fill.end());
            if (minfill < h->frame_in) break;
            int rc = submit_locked(h);
            if (rc != NVX_OK) return rc;
        }
    }
    return nvx_collect_locked(h);
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
    size_t pad = (h->frame_in - total % h->frame_in) % h->frame_in;       // silence up to a whole frame
    std::fill(buf.begin(), buf.end(), (int16_t)0);
    while (pad) {
        size_t m = std::min<size_t>(pad, 65536);
        rc = nvx_push_iq(h, stream, buf.data(), m);
        if (rc != NVX_OK) return rc;
        pad -= m;
    }
    rc = nvx_flush(h);
    if (rc != NVX_OK) return rc;
    return (int)((total + h->frame_in - 1) / h->frame_in);
}
