// nvx_synth.hip -- deterministic CPFSK test source on the device (gfx950; no reference counterpart)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nvx_synth.h"
#include "nvx_kernels.h"

// ===========================================================================
// synthetic source
// ===========================================================================
__global__ __launch_bounds__(256) void nvx_synth_kernel(nvx_synth_args a)
{
    const int stream = blockIdx.y;
    const size_t quad = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // 4 samples per thread
    const size_t n0 = quad * 4;
    if (n0 >= a.n) return;
    const nvx_synth_desc *d = a.desc + stream;
    int32_t I[4] = { 0, 0, 0, 0 }, Q[4] = { 0, 0, 0, 0 };
    const int nc = d->n_carriers;
    for (int c = 0; c < nc; c++) {                    // carrier-major: the 4-sample arrays keep static indices
        const uint64_t g = (uint64_t)n0 + d->bit_offset[c];
        size_t b = (size_t)(g / a.spb);
        uint32_t r = (uint32_t)(g - (uint64_t)b * a.spb);
        const nvx_period *pool = a.pool + d->pool_off[c];
        nvx_period p = pool[b];
        uint32_t ph = p.phase + r * p.inc;
        const int32_t amp = d->amp[c];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            nvx_synth_tone(ph, amp, &I[k], &Q[k]);
            ph += p.inc;
            if (++r == a.spb) { r = 0; b++; p = pool[b]; ph = p.phase; }     // next bit period
        }
    }
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (d->noise_amp > 0) nvx_synth_noise(d->seed, (uint64_t)(n0 + k), d->noise_amp, &I[k], &Q[k]);
        w[k] = nvx_synth_pack(I[k], Q[k]);
    }
    uint32_t *out = a.out + (size_t)stream * a.pitch + n0;
    if (n0 + 4 <= a.n) {
        *(uint4 *)out = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (size_t k = 0; n0 + k < a.n; k++) out[k] = w[k];
    }
}

extern "C" hipError_t nvx_launch_synth(const nvx_synth_args *a, int n_streams, hipStream_t s)
{
    const size_t quads = (a->n + 3) / 4;
    dim3 grid((unsigned)((quads + 255) / 256), (unsigned)n_streams), block(256);
    hipLaunchKernelGGL(nvx_synth_kernel, grid, block, 0, s, *a);
    return hipGetLastError();
}
