// nvx_fir3.hip -- FIR3 (71 taps, /10: 9 kS/s -> 900 S/s, receiver/fir3cpp.C:22-60, taps receiver/fir3cpp.h:16-88) as a
// kernel of its own, behind the fused wideband kernel, whose waves end at FIR2 (r4).
//
// Why, and why only there: inside a cascade wave FIR3 runs on 32 (two chains: 2 x 16) of the 64 lanes with a 4.8 KB input
// buffer per wave.  In the fused wideband kernel -- bound by fp64 issue, one 8-wave workgroup per CU, so a CU has wave
// slots, registers and 50 KB of LDS to spare -- taking it out made the kernel 4.8 % faster and the step 4.7 % (same box,
// interleaved: 18.61 -> 17.72 ms, 18.9 -> 18.0 ms); here every lane computes an output and the work runs beside the
// NEXT launch on the second stream, like the demodulator it feeds.  The single-wave 252 kS/s cascade (Variant A) gained
// 2.4 % in the kernel and nothing in the step: its persistent grid fills every CU, nvx_fir3 beside it takes 42 ms
// instead of 4.6, and the chain FIR3 -> demodulator then sets the step (profiles/r04/e0_*; profiles/TUNING.md); it
// keeps FIR3 inside.  So do the raw-rate kernels: HBM-bound, and 9 kS/s fp64 out and back would be + 3.6 % traffic.
//
// Arithmetic contract, unchanged: y3[k] = sum_{i=0..70} h3[i] * y2[10k + 9 - i], ONE lane per (output, component),
// acc = 0.0 then acc = acc + h3[i] * x in tap order, product and sum rounded separately (-ffp-contract=off).
//
// Mapping: a workgroup = one wave = one frame (288 outputs) of one chain, nine tiles of 32 outputs x {I, Q}.  A tile's
// input -- 320 new FIR2 outputs behind 70 of history, 6.2 KB -- is staged in LDS with coalesced 16-byte loads, one tile
// ahead in registers; lane = (output, component) reads its 71 samples from there (the cascade's own FIR3 access pattern).
// History across launches: a row of the y2 buffer has NVX_Y2_PREFIX entries in front of the launch's outputs; the wave of
// a row's LAST frame copies the row's last 70 outputs into the prefix of the OTHER buffer's row, where the stream's next
// launch (other parity) finds them (nvx_kernels.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nvx_tables.h"
#include "nvx_kernels.h"
#include "nvx_device.h"
#include "nvx_cascade_wave.h"          // NVX_F23_AHEAD

#define F3_TILE_OUT 32                                  /* outputs per tile                              */
#define F3_TILE_IN (10 * F3_TILE_OUT)                   /* new FIR2 outputs per tile                     */
#define F3_WIN (F3_TILE_IN + 70)                        /* window of a tile                              */
#define F3_LOADS ((F3_WIN + 63) / 64)                   /* 16-byte loads per lane per tile               */
static_assert(NVX_Y3_PER_FRAME % F3_TILE_OUT == 0 && NVX_Y2_PREFIX >= 70, "tiles fill a frame; the prefix holds the history");

__global__ __launch_bounds__(64) void nvx_fir3(nvx_fir3_args a)
{
    __shared__ double2 win[F3_WIN + 2];
    const int lane = threadIdx.x, frame = blockIdx.y;
    // which chain: every slot (inactive ones leave at once), or the slots of the launch's participants (nvx_kernels.h)
    int slot = blockIdx.x, parity = 0;
    if (a.part) {
        const int per = 2 * a.per_part, e = slot / per;
        parity = a.part[e].parity;
        slot = a.part[e].stream * per + (slot - e * per);
    }
    const int row = a.y2_row[slot];
    if (row < 0) return;
    // without a list the host passes the buffer the launch's cascade wrote as [0]
    const double2 *in = (a.part ? a.y2[parity] : a.y2[0]) + (size_t)row * a.y2_pitch + NVX_Y2_PREFIX;     // the launch's first output
    double2 *y3 = a.y3 + (size_t)slot * a.y3_cap + a.y3_base + (size_t)frame * NVX_Y3_PER_FRAME;
    const int o = lane >> 1, comp = lane & 1;

    const double2 *src = in + (size_t)frame * NVX_Y2_PER_FRAME - 70;       // the frame's window starts 70 outputs earlier
    nvx_d2 pf[F3_LOADS];
    auto request = [&](const double2 *p) {
#pragma unroll
        for (int j = 0; j < F3_LOADS; j++) {
            const int i = lane + 64 * j;
            pf[j] = *(const nvx_d2 *)(p + (i < F3_WIN ? i : F3_WIN - 1));
        }
    };
    request(src);
    constexpr int TILES = NVX_Y3_PER_FRAME / F3_TILE_OUT;
    for (int t = 0; t < TILES; t++) {
        NVX_WAVE_LDS_FENCE();
#pragma unroll
        for (int j = 0; j < F3_LOADS; j++) {
            const int i = lane + 64 * j;
            if (i < F3_WIN) *(lds_vd2 *)&win[i] = pf[j];
        }
        if (t + 1 < TILES) request(src + (size_t)(t + 1) * F3_TILE_IN);
        NVX_WAVE_LDS_FENCE();
        // y3[k] with k = 32 t + o: samples y2[10k + 9 - i] = win[10 o + 79 - i]
        const lds_vdouble *yb = (const lds_vdouble *)((const double *)&win[10 * o] + comp);
        double xs3[NVX_T3], acc = 0.0;
#pragma unroll
        for (int i = 0; i < NVX_F23_AHEAD; i++) xs3[i] = yb[2 * (79 - i)];
        nvx_static_for<0, NVX_T3>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i + NVX_F23_AHEAD < NVX_T3) { NVX_PIN_AFTER(acc); xs3[i + NVX_F23_AHEAD] = yb[2 * (79 - (i + NVX_F23_AHEAD))]; }
            acc += NVX_H3[i] * xs3[i];
        });
        ((double *)(y3 + t * F3_TILE_OUT + o))[comp] = acc;
    }
    // the row's last 70 outputs become the history of the stream's next launch: prefix of the other buffer's row
    if (frame == a.n_frames - 1) {
        double2 *next = (a.part ? a.y2[parity ^ 1] : a.y2[1]) + (size_t)row * a.y2_pitch + (NVX_Y2_PREFIX - 70);
        const double2 *tail = in + (size_t)a.n_frames * NVX_Y2_PER_FRAME - 70;
        for (int i = lane; i < 70; i += 64) *(nvx_d2 *)(next + i) = *(const nvx_d2 *)(tail + i);
    }
}

extern "C" hipError_t nvx_launch_fir3(const nvx_fir3_args *a, hipStream_t s)
{
    const unsigned chains = a->part ? (unsigned)(2 * a->per_part * a->n_part) : (unsigned)a->n_slots;
    hipLaunchKernelGGL(nvx_fir3, dim3(chains, (unsigned)a->n_frames), dim3(64), 0, s, *a);
    return hipGetLastError();
}
