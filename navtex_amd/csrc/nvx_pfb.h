// nvx_pfb.h -- arithmetic of the 8-channel polyphase channeliser for ONE output instant (device code), shared by the
// stand-alone kernel (nvx_channelise.hip) and the fused wideband kernel (nvx_wideband_fused.hip).
// Definition (integer only; the test suite holds an independent scalar restatement of exactly these steps):
//   u[p]  = (sum_{j = p mod 8} h[47-j] * x[8m-40+j] + 16) >> 5
//   Y[k]  = radix-2 DIT DFT_8(u), 45-degree twiddles = 23170 / 2^15 with floor shifts
//   out_k = clamp16((Y[k] + 4096) >> 13)
#ifndef NVX_PFB_INCLUDED
#define NVX_PFB_INCLUDED

#include <hip/hip_runtime.h>
#include "nvx_device.h"

#define NVX_PFB_TABLE static constexpr
#include "nvx_pfb_taps.h"

__device__ __forceinline__ int mulc45(int t) { return (int)(((long long)t * NVX_PFB_C45) >> 15); }
__device__ __forceinline__ int round_clamp16(int v)
{
    v = (v + 4096) >> 13;
    return v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
}

// win: the 48 packed IQ words x[8m-40 .. 8m+7] of this instant in LDS (16-byte aligned); yr / yi: the eight sub-band
// samples as integers in the int16 range.  Twelve ds_read_b128; the branch sums use v_dot2_i32_i16 with (h, 0) / (0, h)
// selector constants, so no sign extension is needed.
__device__ __forceinline__ void nvx_pfb_instant(const unsigned *win, int (&yr)[8], int (&yi)[8])
{
    int ur[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, ui[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int r = 0; r < 12; r++) {
        const u32x4 w = *(const u32x4 *)&win[4 * r];
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
    yr[0] = round_clamp16(br[0] + br[4]); yi[0] = round_clamp16(bi[0] + bi[4]);
    yr[4] = round_clamp16(br[0] - br[4]); yi[4] = round_clamp16(bi[0] - bi[4]);
    yr[1] = round_clamp16(br[1] + w1r);   yi[1] = round_clamp16(bi[1] + w1i);
    yr[5] = round_clamp16(br[1] - w1r);   yi[5] = round_clamp16(bi[1] - w1i);
    yr[2] = round_clamp16(br[2] + bi[6]); yi[2] = round_clamp16(bi[2] - br[6]);
    yr[6] = round_clamp16(br[2] - bi[6]); yi[6] = round_clamp16(bi[2] + br[6]);
    yr[3] = round_clamp16(br[3] + w3r);   yi[3] = round_clamp16(bi[3] + w3i);
    yr[7] = round_clamp16(br[3] - w3r);   yi[7] = round_clamp16(bi[3] - w3i);
}

// The same instant split over a PAIR of lanes, one per component (c = 0: real, 1: imaginary), so that 32 instants fill a
// wave.  A lane accumulates its own component of all eight branches (the (h, 0) / (0, h) selector is per lane), runs the
// butterflies of its component and fetches the partner's value (DPP pair swap) where the transform rotates by -j or by
// 45 degrees; with sigma = +1 on the real lane, -1 on the imaginary one, those steps read
//   (x + jy) * (-j):  mine + sigma * other        W1: mulc45(mine + sigma * other)        W3: mulc45(-mine + sigma * other)
// Same integer operations on the same operands as nvx_pfb_instant, in the same order: bit-identical results.
// y[k]: this lane's component of sub-band k, an integer in the int16 range.
__device__ __forceinline__ void nvx_pfb_instant_split(const unsigned *win, int c, int (&y)[8])
{
    const int sh = 16 * c, sm = -c;                  // selector shift; sign mask: (v ^ sm) - sm = sigma * v
    // The rounding constant of the branch sums, (sum + 16) >> 5, rides on each branch's FIRST product: the VOP3P form of
    // the dot product takes it as an inline constant.  (Accumulators that start at 16 compiled to eight v_mov_b32 per
    // instant and component in front of the v_dot2c chain: r6, counted in the ISA, tests/test_isa.py.)
    int u[8];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        const u32x4 w = *(const u32x4 *)&win[4 * r];
        const unsigned ww[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int j = 4 * r + e;
            const unsigned hs = ((unsigned)(unsigned short)NVX_PFB_H[47 - j]) << sh;
            if (j < 8) asm("v_dot2_i32_i16 %0, %1, %2, 16" : "=v"(u[j]) : "v"(ww[e]), "v"(hs));
            else u[j & 7] = __builtin_amdgcn_sdot2(as_short2(ww[e]), as_short2(hs), u[j & 7], false);
        }
    }
#pragma unroll
    for (int p = 0; p < 8; p++) u[p] >>= 5;
    // Branch 0 enters every output of the transform with weight +1 and none of the values fetched from the partner lane
    // (a3, a7, b5, b6, b7 below) contains it, so the output rounding constant (Y + 4096) >> 13 is added once, here.
    u[0] += 4096;
    auto swap = [](int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); };      // the partner lane's value
    auto sig = [sm](int v) { return (v ^ sm) - sm; };
    auto out = [](int v) { v >>= 13; return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); };
    int a[8], b[8];
    a[0] = u[0] + u[4]; a[1] = u[0] - u[4]; a[2] = u[2] + u[6]; a[3] = u[2] - u[6];
    a[4] = u[1] + u[5]; a[5] = u[1] - u[5]; a[6] = u[3] + u[7]; a[7] = u[3] - u[7];
    const int o3 = sig(swap(a[3])), o7 = sig(swap(a[7]));
    b[0] = a[0] + a[2]; b[2] = a[0] - a[2]; b[4] = a[4] + a[6]; b[6] = a[4] - a[6];
    b[1] = a[1] + o3;   b[3] = a[1] - o3;   b[5] = a[5] + o7;   b[7] = a[5] - o7;
    const int o5 = sig(swap(b[5])), o6 = sig(swap(b[6])), o7b = sig(swap(b[7]));
    const int w1 = mulc45(b[5] + o5), w3 = mulc45(o7b - b[7]);
    y[0] = out(b[0] + b[4]); y[4] = out(b[0] - b[4]);
    y[1] = out(b[1] + w1);   y[5] = out(b[1] - w1);
    y[2] = out(b[2] + o6);   y[6] = out(b[2] - o6);
    y[3] = out(b[3] + w3);   y[7] = out(b[3] - w3);
}

#endif
