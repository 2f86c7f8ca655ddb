// nvx_synth_dev.cpp -- device launcher of the synthetic source (header section F).
#include "nvx_handle.h"

// ------------------------------------------------------------ synthetic source
extern "C" int nvx_synth_device(int device, const nvx_synth_stream *streams, int n_streams,
                                uint32_t sample_rate, size_t n, void *d_out, size_t pitch)
{
    if (!streams || n_streams < 1 || !d_out || (sample_rate != NVX_RATE_RAW && sample_rate != NVX_RATE_IN) || pitch < n || (pitch & 3)) {
        nvx_set_error("nvx_synth_device: bad argument"); return NVX_ERR_ARG;
    }
    int rc = nvx_select_device(device); if (rc != NVX_OK) return rc;
    if ((rc = nvx_check_device_span(d_out, ((size_t)(n_streams - 1) * pitch + n) * 4, "nvx_synth_device: output")) != NVX_OK) return rc;
    const uint32_t spb = sample_rate / 100;
    std::vector<nvx_synth_desc> desc(n_streams);
    std::vector<nvx_period> pool;
    for (int s = 0; s < n_streams; s++) {
        const nvx_synth_stream &st = streams[s];
        if (st.n_carriers < 0 || st.n_carriers > NVX_SYNTH_MAX_CARRIERS) { nvx_set_error("nvx_synth_device: stream %d: bad carrier count", s); return NVX_ERR_ARG; }
        nvx_synth_desc &d = desc[s];
        memset(&d, 0, sizeof d);
        d.seed = st.seed; d.noise_amp = st.noise_amp; d.n_carriers = st.n_carriers;
        for (int c = 0; c < st.n_carriers; c++) {
            if (st.carrier[c].bit_offset >= spb) { nvx_set_error("nvx_synth_device: bit_offset >= samples per bit"); return NVX_ERR_ARG; }
            size_t periods = (n + st.carrier[c].bit_offset) / spb + 2;
            d.amp[c] = st.carrier[c].amplitude; d.bit_offset[c] = st.carrier[c].bit_offset;
            d.pool_off[c] = (uint32_t)pool.size();
            pool.resize(pool.size() + periods);
            nvx_synth_periods(&st.carrier[c], sample_rate, 0, periods, pool.data() + d.pool_off[c]);
        }
    }
    nvx_synth_desc *d_desc = nullptr; nvx_period *d_pool = nullptr;
    HIP_TRY(hipMalloc(&d_desc, desc.size() * sizeof(nvx_synth_desc)));
    hipError_t e = hipMalloc(&d_pool, std::max<size_t>(pool.size(), 1) * sizeof(nvx_period));
    if (e != hipSuccess) { hipFree(d_desc); nvx_set_error("hipMalloc pool failed: %s", hipGetErrorString(e)); return NVX_ERR_NOMEM; }
    hipMemcpy(d_desc, desc.data(), desc.size() * sizeof(nvx_synth_desc), hipMemcpyHostToDevice);
    if (!pool.empty()) hipMemcpy(d_pool, pool.data(), pool.size() * sizeof(nvx_period), hipMemcpyHostToDevice);
    nvx_synth_args a{};
    a.desc = d_desc; a.pool = d_pool; a.out = (uint32_t *)d_out; a.pitch = pitch; a.n = n; a.spb = spb;
    e = nvx_launch_synth(&a, n_streams, nullptr);
    hipError_t e2 = hipDeviceSynchronize();
    hipFree(d_desc); hipFree(d_pool);
    if (e != hipSuccess || e2 != hipSuccess) {
        nvx_set_error("synth kernel failed: %s", hipGetErrorString(e != hipSuccess ? e : e2)); return NVX_ERR_HIP;
    }
    return NVX_OK;
}
