// nvx_wideband.cpp -- the STAND-ALONE channeliser's entry points (header section G: nvx_channelise_resident, its timing, the
// handle's stream).  A wideband HANDLE does not come through here: its launches run nvx_wideband_fused + nvx_fir3
// (nvx_api.cpp, nvx_launch_locked), where the sub-bands never leave the LDS.
#include "nvx_handle.h"

// ------------------------------------------------------------ wideband front-end
extern "C" void *nvx_handle_stream(nvx_handle *h) { return h ? (void *)h->stream : nullptr; }

static struct ChanTiming {
    std::mutex mu;
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool, pending;
    double sum_ms = 0.0; uint64_t n = 0;
} g_ct;

extern "C" void nvx_channelise_timing(int enable) { std::lock_guard<std::mutex> lk(g_ct.mu); g_ct.on = enable != 0; }

extern "C" int nvx_channelise_time_stats(double *sum_ms, uint64_t *launches, int reset)
{
    std::lock_guard<std::mutex> lk(g_ct.mu);
    for (auto &p : g_ct.pending) {
        HIP_TRY(hipEventSynchronize(p.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        g_ct.sum_ms += ms; g_ct.n++;
        g_ct.pool.push_back(p);
    }
    g_ct.pending.clear();
    if (sum_ms) *sum_ms = g_ct.sum_ms;
    if (launches) *launches = g_ct.n;
    if (reset) { g_ct.sum_ms = 0.0; g_ct.n = 0; }
    return NVX_OK;
}

extern "C" int nvx_channelise_resident(int device, const void *d_raw, size_t pitch_raw, size_t first_sample, int n_wide,
                                       size_t n_out, const void *d_hist_in, void *d_hist_out, void *d_sub, size_t pitch_sub,
                                       size_t sub_first, void *hip_stream)
{
    if (!d_raw || !d_sub || n_wide < 1 || n_out == 0 || (n_out % 64) || (pitch_raw & 3) || (first_sample & 3) ||
        (d_hist_in && d_hist_in == d_hist_out)) {
        nvx_set_error("nvx_channelise_resident: bad argument (n_out must be a multiple of 64, pitches/offsets of 4)");
        return NVX_ERR_ARG;
    }
    int rc = nvx_select_device(device); if (rc != NVX_OK) return rc;
    // every operand's last element must lie inside its allocation (nvx_handle.h: a faulting kernel can take the node down)
    if ((rc = nvx_check_device_span(d_raw, ((size_t)(n_wide - 1) * pitch_raw + first_sample + 8 * n_out) * 4, "nvx_channelise_resident: raw input")) != NVX_OK) return rc;
    if ((rc = nvx_check_device_span(d_sub, ((size_t)(n_wide * NVX_WB_SUBBANDS - 1) * pitch_sub + sub_first + n_out) * 4, "nvx_channelise_resident: sub-band output")) != NVX_OK) return rc;
    if (d_hist_in && (rc = nvx_check_device_span(d_hist_in, (size_t)n_wide * 40 * 4, "nvx_channelise_resident: history in")) != NVX_OK) return rc;
    if (d_hist_out && (rc = nvx_check_device_span(d_hist_out, (size_t)n_wide * 40 * 4, "nvx_channelise_resident: history out")) != NVX_OK) return rc;
    nvx_channelise_args a{};
    a.raw = (const uint32_t *)d_raw; a.pitch_raw = pitch_raw; a.first_sample = first_sample; a.n_wide = n_wide; a.n_out = n_out;
    a.hist_in = (const uint32_t *)d_hist_in; a.hist_out = (uint32_t *)d_hist_out;
    a.sub = (uint32_t *)d_sub; a.pitch_sub = pitch_sub; a.sub_first = sub_first;
    // enough blocks to fill the chip several times over, long enough spans to amortise the 40-sample halo
    const size_t n_chunks = n_out / 64;
    size_t cpb = (n_chunks * (size_t)n_wide + 16383) / 16384;
    if (cpb < 8) cpb = 8;
    a.chunks_per_block = (int)std::min<size_t>(cpb, n_chunks);
    std::pair<hipEvent_t, hipEvent_t> ev{ nullptr, nullptr };
    bool timed = false;
    {
        std::lock_guard<std::mutex> lk(g_ct.mu);
        if (g_ct.on) {
            if (g_ct.pool.empty()) { HIP_TRY(hipEventCreate(&ev.first)); HIP_TRY(hipEventCreate(&ev.second)); }
            else { ev = g_ct.pool.back(); g_ct.pool.pop_back(); }
            timed = true;
        }
    }
    if (timed) HIP_TRY(hipEventRecord(ev.first, (hipStream_t)hip_stream));
    HIP_TRY(nvx_launch_channelise(&a, (hipStream_t)hip_stream));
    if (timed) {
        HIP_TRY(hipEventRecord(ev.second, (hipStream_t)hip_stream));
        std::lock_guard<std::mutex> lk(g_ct.mu);
        g_ct.pending.push_back(ev);
    }
    return NVX_OK;
}
