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
// LDS words 8m .. 8m+47 (twelve ds_read_b128); the branch sums use v_dot2_i32_i16
// with (h, 0) / (0, h) selector constants, so no sign extension is needed.
#define NVX_PFB_TABLE static constexpr
#include "nvx_pfb_taps.h"

__device__ __forceinline__ int mulc45(int t) { return (int)(((long long)t * NVX_PFB_C45) >> 15); }
__device__ __forceinline__ unsigned pack_clamp16(int re, int im)
{
    re = (re + 4096) >> 13; im = (im + 4096) >> 13;
    re = re > 32767 ? 32767 : (re < -32768 ? -32768 : re);
    im = im > 32767 ? 32767 : (im < -32768 ? -32768 : im);
    return ((unsigned)re & 0xffffu) | ((unsigned)im << 16);
}

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

        int ur[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, ui[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < 12; r++) {
            const u32x4 w = *(const u32x4 *)&win[8 * lane + 4 * r];
            const unsigned ww[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int j = 4 * r + e;
                const int h = NVX_PFB_H[47 - j];
                const nvx_short2 hI = { (short)h, 0 }, hQ = { 0, (short)h };
                ur[j & 7] = __builtin_amdgcn_sdot2(as_short2(ww[e]), hI, ur[j & 7], false);
                ui[j & 7] = __builtin_amdgcn_sdot2(as_short2(ww[e]), hQ, ui[j & 7], false);
            }
        }
#pragma unroll
        for (int p = 0; p < 8; p++) { ur[p] = (ur[p] + 16) >> 5; ui[p] = (ui[p] + 16) >> 5; }

        int ar[8], ai[8], br[8], bi[8];
        ar[0] = ur[0] + ur[4]; ai[0] = ui[0] + ui[4];  ar[1] = ur[0] - ur[4]; ai[1] = ui[0] - ui[4];
        ar[2] = ur[2] + ur[6]; ai[2] = ui[2] + ui[6];  ar[3] = ur[2] - ur[6]; ai[3] = ui[2] - ui[6];
        ar[4] = ur[1] + ur[5]; ai[4] = ui[1] + ui[5];  ar[5] = ur[1] - ur[5]; ai[5] = ui[1] - ui[5];
        ar[6] = ur[3] + ur[7]; ai[6] = ui[3] + ui[7];  ar[7] = ur[3] - ur[7]; ai[7] = ui[3] - ui[7];
        br[0] = ar[0] + ar[2]; bi[0] = ai[0] + ai[2];  br[2] = ar[0] - ar[2]; bi[2] = ai[0] - ai[2];
        br[1] = ar[1] + ai[3]; bi[1] = ai[1] - ar[3];  br[3] = ar[1] - ai[3]; bi[3] = ai[1] + ar[3];
        br[4] = ar[4] + ar[6]; bi[4] = ai[4] + ai[6];  br[6] = ar[4] - ar[6]; bi[6] = ai[4] - ai[6];
        br[5] = ar[5] + ai[7]; bi[5] = ai[5] - ar[7];  br[7] = ar[5] - ai[7]; bi[7] = ai[5] + ar[7];
        const int w1r = mulc45(br[5] + bi[5]), w1i = mulc45(bi[5] - br[5]);
        const int w3r = mulc45(bi[7] - br[7]), w3i = mulc45(-br[7] - bi[7]);
        unsigned y[8];
        y[0] = pack_clamp16(br[0] + br[4], bi[0] + bi[4]);  y[4] = pack_clamp16(br[0] - br[4], bi[0] - bi[4]);
        y[1] = pack_clamp16(br[1] + w1r, bi[1] + w1i);      y[5] = pack_clamp16(br[1] - w1r, bi[1] - w1i);
        y[2] = pack_clamp16(br[2] + bi[6], bi[2] - br[6]);  y[6] = pack_clamp16(br[2] - bi[6], bi[2] + br[6]);
        y[3] = pack_clamp16(br[3] + w3r, bi[3] + w3i);      y[7] = pack_clamp16(br[3] - w3r, bi[3] - w3i);

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
