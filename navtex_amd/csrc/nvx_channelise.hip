// nvx_channelise.hip -- wideband front-end kernel (gfx950): 2.016 MS/s -> 8 x 252 kS/s (no reference counterpart)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nvx_kernels.h"
#include "nvx_device.h"

// ===========================================================================
// wideband front-end: 8-channel polyphase channeliser (build-owned, integer)
// ===========================================================================
// One 2.016 MS/s stream -> eight 252 kS/s sub-bands centred at k * 252 kHz, each
// ready for the 252 kS/s cascade (two NAVTEX chains per sub-band).  Definition
// (the test suite holds an independent scalar restatement of exactly these steps):
//   u[p]  = (sum_{j = p mod 8} h[47-j] * x[8m-40+j] + 16) >> 5
//   Y[k]  = radix-2 DIT DFT_8(u), 45-degree twiddles = 23170 / 2^15 with floor shifts
//   out_k = clamp16((Y[k] + 4096) >> 13)
// Mapping: one wave per span of one wideband stream; a chunk = 64 output instants =
// 512 raw samples (2 KiB in, 8 x 256 B out).  The 48-sample window of lane m is
// LDS words 8m .. 8m+47; the arithmetic of one instant is nvx_pfb.h.
#include "nvx_pfb.h"

__global__ __launch_bounds__(64) void nvx_channelise(nvx_channelise_args a)
{
    __shared__ __attribute__((aligned(16))) unsigned win[40 + 512];
    const int lane = threadIdx.x;
    const int wide = blockIdx.y;
    const size_t n_chunks = a.n_out / 64;
    const size_t c0 = (size_t)blockIdx.x * a.chunks_per_block;
    if (c0 >= n_chunks) return;
    const size_t c1 = min(n_chunks, c0 + (size_t)a.chunks_per_block);
    const unsigned *raw = a.raw + (size_t)wide * a.pitch_raw + a.first_sample;

    // history: the 40 raw samples in front of this span
    if (lane < 40) {
        unsigned v = 0;
        if (c0 > 0) v = raw[c0 * 512 - 40 + lane];
        else if (a.hist_in) v = a.hist_in[(size_t)wide * 40 + lane];
        win[lane] = v;
    }
    for (size_t c = c0; c < c1; c++) {
        const u32x4 *src = (const u32x4 *)(raw + c * 512) + lane;
        const u32x4 v0 = __builtin_nontemporal_load(src), v1 = __builtin_nontemporal_load(src + 64);
        *(u32x4 *)&win[40 + 4 * lane] = v0;
        *(u32x4 *)&win[40 + 256 + 4 * lane] = v1;
        NVX_WAVE_LDS_FENCE();

        int yr[8], yi[8];
        nvx_pfb_instant(&win[8 * lane], yr, yi);
        unsigned y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) y[k] = ((unsigned)yr[k] & 0xffffu) | ((unsigned)yi[k] << 16);

        unsigned *out = a.sub + (size_t)wide * 8 * a.pitch_sub + a.sub_first + c * 64 + lane;
#pragma unroll
        for (int k = 0; k < 8; k++) out[(size_t)k * a.pitch_sub] = y[k];

        // slide: the newest 40 raw samples become the history of the next chunk
        NVX_WAVE_LDS_FENCE();
        unsigned t = 0;
        if (lane < 40) t = win[512 + lane];
        NVX_WAVE_LDS_FENCE();
        if (lane < 40) win[lane] = t;
        NVX_WAVE_LDS_FENCE();
    }
    if (c1 == n_chunks && a.hist_out && lane < 40) a.hist_out[(size_t)wide * 40 + lane] = win[lane];
}

extern "C" hipError_t nvx_launch_channelise(const nvx_channelise_args *a, hipStream_t s)
{
    const size_t n_chunks = a->n_out / 64;
    const unsigned bx = (unsigned)((n_chunks + a->chunks_per_block - 1) / a->chunks_per_block);
    hipLaunchKernelGGL(nvx_channelise, dim3(bx, (unsigned)a->n_wide), dim3(64), 0, s, *a);
    return hipGetLastError();
}
