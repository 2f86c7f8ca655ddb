// nvx_demod.hip -- the demodulator kernels (gfx950): nvx_demod_front + nvx_demod_fsm, 900 S/s -> 'B'/'Y' bits
//        discriminator          receiver/decoder.C:42-59
//        bit-timing filter      receiver/decoder.C:142-255
//        mark/space decision    receiver/decoder.C:73-137
// Compiled with -ffp-contract=off: the only v_fma_f64 in the ISA are the explicit error-free
// transformations of nvx_atan2 and the expansion of IEEE division.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "nvx_tables.h"
#include "nvx_atan2.h"
#include "nvx_fsm.h"
#include "nvx_kernels.h"

// ===========================================================================
// demodulator (receiver/decoder.C), split by what is parallel in time
// ===========================================================================
// With g = index of a 900 S/s sample since reset, the reference's counters are
// pure functions of g (decoder.C:142-255):
//   delta-phi ring primed at g = 8   -> |corr| value kappa = g - 8 written to
//                                       ring position kappa mod 567
//   |corr| ring primed at g = 574    -> one class sum per sample, class
//                                       c(g) = (g - 574) mod 9, over ring
//                                       positions c, c+9, ... in ASCENDING
//                                       POSITION order (not time order)
//   class sums primed at g = 582     -> arg-max over the 9 sums when
//                                       (g - 582) mod 9 == 0; at that moment
//                                       csa[i] = S(g - 8 + i)
// so delta-phi, |corr|, the class sums S(g) and the arg-max are computed for all
// samples of a launch in parallel (nvx_demod_front, one workgroup per chain,
// time-tiled through LDS), and only the two tiny state machines (timing slew
// limiter, mark/space bit FSM with its five-sample mixed-precision
// accumulation) run sequentially, one lane per chain (nvx_demod_fsm).
// Every floating-point sum keeps the reference's operand order.
//
// The mark/space decision of a bit depends only on the five consecutive samples
// of its window (decoder.C:96-125: the sums are zeroed when the window opens),
// so the decision "if a window ended at sample t" is evaluated for EVERY t in
// parallel too; the sequential kernel then only picks the one the bit FSM lands
// on.  Per bit period m the front kernel hands over one 16-bit word:
//   bits 0..8  decision for a window ending at local sample 9m+k ('B' = 1)
//   bits 12..15 arg-max of that period's timing evaluation, 15 = none yet
//
// Per-slot double state (AoS): last 4 samples {I,Q} (the newest is the
// discriminator's prevI/prevQ), last 8 delta-phi, last 8 class sums, last 567
// |corr| values in time order.
enum { DS_Y3 = 0, DS_DPHI = 8, DS_S = 16, DS_C = 24, DS_COUNT = 24 + 567 };
// Per-slot int state (SoA over slots)
enum { DI_SYNCED = 0, DI_SYNC_OFF, DI_NEXT_SYNC_OFF, DI_PHASE, DI_PREV_OFFSET, DI_COUNT };
static_assert(DI_COUNT == NVX_DEMOD_INTS && DS_COUNT == NVX_DEMOD_DOUBLES && DI_PREV_OFFSET == NVX_DI_PREV_OFFSET &&
              DI_PHASE == NVX_DI_PHASE, "state layout");

// Workgroup shape: 128 threads on a time tile of 432 samples (a multiple of 9: whole bit periods) = 15.4 KB of LDS, so that a workgroup
// finds room on a CU beside the NEXT launch's persistent cascade grid (the host runs the demodulator of launch k
// beside the cascade of launch k + 1).  Stand-alone the shapes 256 / 1152 and 192 / 576 were equally fast in round 1
// (0.68 / 0.72 ms).
#ifndef NVX_FRONT_THREADS
#define NVX_FRONT_THREADS 128
#endif
#ifndef DTL
#define DTL 432
static_assert(DTL % 9 == 0, "a tile is a whole number of bit periods");
#endif
#define FRONT_SLIDE ((567 + NVX_FRONT_THREADS - 1) / NVX_FRONT_THREADS)
#define G_DAB 8
#define G_CB 574
#define G_CSA 582

// sample t of the launch, t >= -4: history for negative t
__device__ __forceinline__ double2 y3_at(const double2 *y3, const double *hist, int t)
{
    if (t >= 0) return y3[t];
    double2 r; r.x = hist[2 * (4 + t)]; r.y = hist[2 * (4 + t) + 1];
    return r;
}

// ---- the per-sample pieces, shared by the two forms of the front kernel ------------------------------------------
// discriminator, decoder.C:48-52
__device__ __forceinline__ double front_dphi(double2 s, double2 p)
{
    const double prodReal = s.x * p.x + s.y * p.y;
    const double prodImg  = s.y * p.x - s.x * p.y;
    return nvx_atan2(prodImg, prodReal);
}
// mark/space decision for a window ENDING at sample t, decoder.C:115-132: float*float product, double*float product,
// double sum, accumulate in double, round to float -- five samples, filter index 0..4
__device__ __forceinline__ unsigned char front_decision(double2 w0, double2 w1, double2 w2, double2 w3, double2 w4)     // w_i = sample t - 4 + i
{
    float BR = 0.0f, BI = 0.0f, YR = 0.0f, YI = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const double2 w = i == 0 ? w0 : i == 1 ? w1 : i == 2 ? w2 : i == 3 ? w3 : w4;
        const float fR = NVX_BF_R[i], fI = NVX_BF_I[i];
        const double sampleR = w.x, sampleI = w.y;
        YR = (float)((double)YR + ((double)((float)sampleR * fR) - sampleI * (double)fI));
        YI = (float)((double)YI + ((double)((float)sampleR * fI) + sampleI * (double)fR));
        BR = (float)((double)BR + ((double)((float)sampleR * fR) + sampleI * (double)fI));
        BI = (float)((double)BI + ((double)((float)(-sampleR) * fI) + sampleI * (double)fR));
    }
    const float Brot = BR * BR + BI * BI;
    const float Yrot = YR * YR + YI * YI;
    return (Brot > Yrot) ? 1 : 0;
}
// transition correlator, decoder.C:157-177: mask[i] * dphi[g-8+i], i ascending; dp points at dphi of sample g-8
__device__ __forceinline__ double front_corr(const double *dp)
{
    double temp = 0.0;
#pragma unroll
    for (int i = 0; i < 9; i++) temp += (double)NVX_CORR_MASK[i] * dp[i];
    return __builtin_fabs(temp);
}
// class sum, decoder.C:181-197: ring positions c, c+9, ... ascending.  Position p holds the newest value kappa' <= kappa
// with kappa' = p (mod 567), i.e. the value d = (kappa - p) mod 567 samples back.  The order is a rotation of the time
// order: with d0 = (kappa - c) mod 567 and jw = d0 / 9 the terms are the samples d0, d0-9, ..., d0-9*jw back (ascending
// in time, stride 9), then those 558+r, ..., d0+9 back (r = d0 mod 9) -- two runs of one stride-9 walk through the
// time-ordered values, the second one starting 567 entries lower.  u == g - 574 (mod 5103 = 9 * 567), cnow points at
// |corr| of the sample itself (older samples at lower addresses).
__device__ __forceinline__ double front_class_sum(const double *cnow, unsigned u)
{
    const unsigned c = u % 9u;
    const int d0 = (int)((u + 566u - c) % 567u);         // (kappa - c) mod 567, kappa = g - 8
    const int jw = d0 / 9;
    const double *run1 = cnow - d0;                      // terms j = 0 .. jw
    const double *run2 = run1 - 567;                     // terms j = jw+1 .. 62
    double temp = 0.0;
#pragma unroll
    for (int j = 0; j < 63; j++) temp += (j <= jw ? run1 : run2)[9 * j];
    return temp;
}
// One word per bit period: nine window decisions + the arg-max of the timing evaluation (decoder.C:202-215: csa[i] =
// S(g-8+i), strict '>' from -1.0 => first maximum wins), which falls on the period's sample 6.  sv points at S of the
// period's sample -2 (so sv[i] = S(g-8+i) at the evaluation); evaluate: the class sums are primed (g >= 582).
struct FrontTies { unsigned near, evals; float minm; };
__device__ __forceinline__ FrontTies front_ties_init() { FrontTies t; t.near = 0; t.evals = 0; t.minm = __uint_as_float(0x7f800000u); return t; }
__device__ __forceinline__ unsigned short front_word(const unsigned char *d9, const double *sv, bool evaluate, FrontTies &ties)
{
    unsigned w = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) w |= (unsigned)d9[k] << k;
    unsigned max_index = 15;
    if (evaluate) {
        double temp_max = -1.0, runner_up = -1.0;
        max_index = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double v = sv[i];
            if (v > temp_max) { runner_up = temp_max; temp_max = v; max_index = i; }
            else if (v > runner_up) runner_up = v;
        }
        // instrumentation only (nvx_tie_stats): how close was that decision?
        if (temp_max > 0.0) {
            const double margin = temp_max - runner_up;
            ties.evals++;
            if (margin < temp_max * 0x1p-40) ties.near++;
            ties.minm = fminf(ties.minm, (float)(margin / temp_max));
        }
    }
    return (unsigned short)(w | (max_index << 12));
}

// which chain a block works on, which of the two state blocks it reads (it writes the other), the chain's sample count
// and how many samples at 900 S/s this launch holds for it -- whole frames, except in the launch that ENDS the chain's
// stream (nvx_finish: nvx_part.n3), where the count is what the stream's real input produced and the last bit period may
// be ragged: the reference's decoder simply stops with its last sample (receiver/capt_sched.c:509-513)
struct FrontChain { int slot; unsigned long long g0; const double *st_rd; double *st_wr; int n3; };
__device__ __forceinline__ FrontChain front_chain(const nvx_demod_args &a, int chain_index)
{
    // every slot, or the slots of the launch's participants (nvx_kernels.h, nvx_part: entry e covers the decoded streams
    // per_part * stream .. + per_part - 1, two slots each, and carries their common sample count and parity)
    FrontChain c;
    c.slot = chain_index; c.g0 = a.g0; c.n3 = a.n3;
    int parity = 0;
    if (a.part) {
        const int per = 2 * a.per_part, e = chain_index / per;
        c.slot = a.part[e].stream * per + (chain_index - e * per);
        c.g0 = a.part[e].g0; parity = a.part[e].parity; c.n3 = a.part[e].n3;
    }
    c.st_rd = (parity ? a.dstate[1] : a.dstate[0]) + (size_t)c.slot * NVX_DEMOD_DOUBLES;
    c.st_wr = (parity ? a.dstate[0] : a.dstate[1]) + (size_t)c.slot * NVX_DEMOD_DOUBLES;
    return c;
}

__device__ __forceinline__ void front_publish_ties(const nvx_demod_args &a, const FrontTies &t, unsigned *s_near, unsigned *s_evals, unsigned *s_minm, int tid)
{
    if (t.evals) { atomicAdd(s_near, t.near); atomicAdd(s_evals, t.evals); atomicMin(s_minm, __float_as_uint(t.minm)); }
    __syncthreads();
    if (tid == 0 && *s_evals) {
        if (*s_near) atomicAdd(&a.ties->near_ties, (unsigned long long)*s_near);
        atomicAdd(&a.ties->evaluations, (unsigned long long)*s_evals);
        atomicMin(&a.ties->min_margin_bits, *s_minm);
    }
}

// ---- form 1: one workgroup per chain walks the launch tile by tile (many chains: the headline) ---------------------
// n_tiles_here > 0: only the first n_tiles_here tiles, and the carried state is NOT stored (the head of the tile-parallel
// form below, whose last tile stores it).
struct FrontWalkLds {
    double dphi[8 + DTL];
    double S[8 + DTL];
    double C[567 + DTL];
    unsigned char D[DTL];
    unsigned near, evals, minm;
};
__device__ __forceinline__ void front_sequential(const nvx_demod_args &a, int chain_index, int n_tiles_here, FrontWalkLds &lds)
{
    double *const s_dphi = lds.dphi, *const s_S = lds.S, *const s_C = lds.C;
    unsigned char *const s_D = lds.D;
    unsigned &s_near = lds.near, &s_evals = lds.evals, &s_minm = lds.minm;
    const int tid = threadIdx.x;
    const FrontChain ch = front_chain(a, chain_index);
    const int slot = ch.slot;
    if (!a.slot_active[slot]) return;                    // uniform over the block
    if (tid == 0) { s_near = 0; s_evals = 0; s_minm = 0x7f800000u; }
    FrontTies ties = front_ties_init();

    const double *st = ch.st_rd;
    const double2 *y3 = a.y3 + (size_t)slot * a.y3_cap + a.y3_base;
    double *dphi_out = a.dphi ? a.dphi + (size_t)slot * a.y3_cap + a.y3_base : nullptr;
    const double *hist = st + DS_Y3;                     // read in place: only the first 4 samples need it
    if (tid < 8) { s_dphi[tid] = st[DS_DPHI + tid]; s_S[tid] = st[DS_S + tid]; }
    for (int i = tid; i < 567; i += NVX_FRONT_THREADS) s_C[i] = st[DS_C + i];
    __syncthreads();

    const int n3 = ch.n3;
    const int n_here = n_tiles_here > 0 ? min(n3, n_tiles_here * DTL) : n3;
    for (int ta = 0; ta < n_here; ta += DTL) {
        const int tl = min(DTL, n_here - ta);
        const int tl9 = (tl + 8) / 9;                    // bit periods of the tile, the last one perhaps ragged (end of the stream)
        const unsigned long long gt = ch.g0 + (unsigned long long)ta;    // g of L = 0
        for (int L = tid; L < tl; L += NVX_FRONT_THREADS) {
            const int t = ta + L;
            const double2 w3 = y3_at(y3, hist, t - 1), w4 = y3[t];
            const double ds = front_dphi(w4, w3);
            s_dphi[8 + L] = ds;
            if (dphi_out) dphi_out[t] = ds;
            s_D[L] = front_decision(y3_at(y3, hist, t - 4), y3_at(y3, hist, t - 3), y3_at(y3, hist, t - 2), w3, w4);
        }
        if (tid < 9 * tl9 - tl) s_D[tl + tid] = 0;       // a ragged last period: no window ends on samples the stream never had
        __syncthreads();
        for (int L = tid; L < tl; L += NVX_FRONT_THREADS)
            s_C[567 + L] = (gt + L >= G_DAB) ? front_corr(&s_dphi[L]) : 0.0;      // (0: never read; keeps the carried state deterministic)
        __syncthreads();
        // t_cb = (g of L = 0) - 574 mod 5103 keeps the index arithmetic in 32 bits
        const unsigned t_cb = (unsigned)((gt % 5103u + (5103u - G_CB % 5103u)) % 5103u);
        for (int L = tid; L < tl; L += NVX_FRONT_THREADS)
            s_S[8 + L] = (gt + L >= G_CB) ? front_class_sum(&s_C[567 + L], t_cb + (unsigned)L) : 0.0;
        __syncthreads();
        for (int M = tid; M < tl9; M += NVX_FRONT_THREADS) {
            const int L = 9 * M + (G_CSA % 9);             // (L >= tl: the stream ended before this period's timing evaluation)
            a.words[(size_t)slot * (a.y3_cap / 9) + (ta / 9 + M)] = front_word(&s_D[9 * M], &s_S[L], L < tl && gt + L >= G_CSA, ties);
        }
        __syncthreads();
        // ---- slide the histories to the front for the next tile / the next launch
        double h_d = 0.0, h_s = 0.0, h_c[FRONT_SLIDE];
        if (tid < 8) { h_d = s_dphi[tl + tid]; h_s = s_S[tl + tid]; }
#pragma unroll
        for (int k = 0; k < FRONT_SLIDE; k++) { const int i = tid + NVX_FRONT_THREADS * k; h_c[k] = (i < 567) ? s_C[tl + i] : 0.0; }
        __syncthreads();
        if (tid < 8) { s_dphi[tid] = h_d; s_S[tid] = h_s; }
#pragma unroll
        for (int k = 0; k < FRONT_SLIDE; k++) { const int i = tid + NVX_FRONT_THREADS * k; if (i < 567) s_C[i] = h_c[k]; }
        __syncthreads();
    }

    front_publish_ties(a, ties, &s_near, &s_evals, &s_minm, tid);
    if (n_tiles_here > 0) return;                        // the last tile of the tile-parallel form stores the state
    double *sw = ch.st_wr;
    if (tid < 4 && n3 >= 4) { const double2 l = y3[n3 - 4 + tid]; sw[DS_Y3 + 2 * tid] = l.x; sw[DS_Y3 + 2 * tid + 1] = l.y; }
    if (tid < 8) { sw[DS_DPHI + tid] = s_dphi[tid]; sw[DS_S + tid] = s_S[tid]; }
    for (int i = tid; i < 567; i += NVX_FRONT_THREADS) sw[DS_C + i] = s_C[i];
}

__global__ __launch_bounds__(NVX_FRONT_THREADS) void nvx_demod_front(nvx_demod_args a)
{
    __shared__ FrontWalkLds lds;
    front_sequential(a, blockIdx.x, 0, lds);
}

// ---- form 2: tile-parallel (few chains, long launches: one channel replayed from a recording, BASELINE configs[1]) --
// Everything a tile's words need is a pure function of the 577 samples in front of it (delta-phi 8 back -> |corr| 567
// back -> class sums 2 back of the tile's first sample) and of g, so from the third tile on a tile is a workgroup of its
// own that rebuilds that look-back from the launch's own 900 S/s samples -- 2.3 x the arithmetic, all tiles at once.
// The first two tiles reach back before the launch: one "head" workgroup walks them with the carried state, as form 1
// does.  The workgroup of the LAST tile stores the carried state: it holds the last 567 |corr| values, the last eight
// delta-phi and class sums.  The state is double-buffered by the stream's parity (read one block, write the other), so
// nobody overwrites what the head still reads.
// (Head and tiles are two kernels: as one kernel with a branch on the workgroup's role this compiler's SimplifyCFG pass
// crashes.  They touch disjoint outputs and the same read-only inputs, so their order does not matter.)
#define FRONT_LOOKBACK 576                                   /* samples of delta-phi in front of the tile (>= 8 + 566 + 2) */
__global__ __launch_bounds__(NVX_FRONT_THREADS) void nvx_demod_front_head(nvx_demod_args a)
{
    __shared__ FrontWalkLds lds;
    front_sequential(a, blockIdx.x, 2, lds);
}
__global__ __launch_bounds__(NVX_FRONT_THREADS) void nvx_demod_front_tiles(nvx_demod_args a, int wgs_per_chain)
{
    const int chain_index = blockIdx.x / wgs_per_chain, wg = blockIdx.x - chain_index * wgs_per_chain;
    __shared__ double p_dphi[FRONT_LOOKBACK + DTL];          // delta-phi of samples ta - 576 .. ta + tl - 1
    __shared__ double p_C[FRONT_LOOKBACK - 8 + DTL];         // |corr|    of samples ta - 568 .. (needs delta-phi 8 back)
    __shared__ double p_S[2 + DTL];                          // class sum of samples ta - 2 ..   (needs |corr| 566 back)
    __shared__ unsigned char p_D[DTL];
    __shared__ unsigned s_near, s_evals, s_minm;
    const int tid = threadIdx.x;
    const FrontChain ch = front_chain(a, chain_index);
    const int slot = ch.slot;
    if (!a.slot_active[slot]) return;
    if (tid == 0) { s_near = 0; s_evals = 0; s_minm = 0x7f800000u; }
    FrontTies ties = front_ties_init();
    const int ta = (wg + 2) * DTL;                           // tiles 2, 3, ...: ta >= 864 > the look-back
    const int tl = min(DTL, a.n3 - ta);
    const double2 *y3 = a.y3 + (size_t)slot * a.y3_cap + a.y3_base;
    double *dphi_out = a.dphi ? a.dphi + (size_t)slot * a.y3_cap + a.y3_base : nullptr;
    const unsigned long long g_ta = ch.g0 + (unsigned long long)ta;

    for (int i = tid; i < FRONT_LOOKBACK + tl; i += NVX_FRONT_THREADS) {
        const int t = ta - FRONT_LOOKBACK + i;
        const double ds = front_dphi(y3[t], y3[t - 1]);
        p_dphi[i] = ds;
        if (i >= FRONT_LOOKBACK) {
            if (dphi_out) dphi_out[t] = ds;
            p_D[i - FRONT_LOOKBACK] = front_decision(y3[t - 4], y3[t - 3], y3[t - 2], y3[t - 1], y3[t]);      // t - 4 >= 0: no history needed
        }
    }
    __syncthreads();
    // |corr| of sample t = ta - 568 + i uses delta-phi of t - 8 .. t = p_dphi[i .. i + 8]
    for (int i = tid; i < FRONT_LOOKBACK - 8 + tl; i += NVX_FRONT_THREADS)
        p_C[i] = (g_ta - 568 + (unsigned long long)i >= G_DAB) ? front_corr(&p_dphi[i]) : 0.0;
    __syncthreads();
    // class sum of sample t = ta - 2 + i: |corr| of t is p_C[566 + i]
    const unsigned t_cb = (unsigned)(((g_ta - 2) % 5103u + (5103u - G_CB % 5103u)) % 5103u);      // g(ta - 2) - 574 mod 5103
    for (int i = tid; i < 2 + tl; i += NVX_FRONT_THREADS)
        p_S[i] = (g_ta - 2 + (unsigned long long)i >= G_CB) ? front_class_sum(&p_C[566 + i], t_cb + (unsigned)i) : 0.0;
    __syncthreads();
    // period M of the tile: decisions p_D[9M .. 9M+8]; evaluation on its sample 6 with S of samples 9M-2 .. 9M+6 = p_S[9M ..]
    for (int M = tid; M < tl / 9; M += NVX_FRONT_THREADS)
        a.words[(size_t)slot * (a.y3_cap / 9) + (ta / 9 + M)] = front_word(&p_D[9 * M], &p_S[9 * M], g_ta + 9 * M + (G_CSA % 9) >= G_CSA, ties);
    front_publish_ties(a, ties, &s_near, &s_evals, &s_minm, tid);
    if (ta + tl >= a.n3) {
        // ---- the last tile: what the next launch carries
        double *sw = ch.st_wr;
        if (tid < 4) { const double2 l = y3[a.n3 - 4 + tid]; sw[DS_Y3 + 2 * tid] = l.x; sw[DS_Y3 + 2 * tid + 1] = l.y; }
        if (tid < 8) { sw[DS_DPHI + tid] = p_dphi[FRONT_LOOKBACK + tl - 8 + tid]; sw[DS_S + tid] = p_S[2 + tl - 8 + tid]; }
        for (int i = tid; i < 567; i += NVX_FRONT_THREADS) sw[DS_C + i] = p_C[FRONT_LOOKBACK - 8 + tl - 567 + i];
    }
}

// Sequential part: the timing slew limiter (decoder.C:217-249) and the bit FSM
// (decoder.C:62-137), integers only, one lane per chain.  Both are stated per
// sample in nvx_fsm.h; the kernel advances a whole bit period at a time with
// the transition table generated from that statement (29 KB, copied to LDS):
// the dependent chain per period is the slew rule, one LDS lookup and a few
// bit operations instead of nine sample steps.
__global__ __launch_bounds__(64) void nvx_demod_fsm(nvx_demod_args a)
{
    __shared__ uint32_t s_tab[NVX_FSM_TABLE_ALLOC];
    {
        const uint4 *src = (const uint4 *)a.fsm_table;
        uint4 *dst = (uint4 *)s_tab;
        for (int i = threadIdx.x; i < NVX_FSM_TABLE_ALLOC / 4; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    int slot = blockIdx.x * 64 + threadIdx.x;
    const int nc = a.n_slots;
    int n3 = a.n3;
    if (a.part) {                                        // the slots of the launch's participants, as in the front kernel
        const int per = 2 * a.per_part, e = slot / per;
        if (e >= a.n_part) return;
        slot = a.part[e].stream * per + (slot - e * per);
        n3 = a.part[e].n3;
    }
    if (slot >= nc) return;
    if (!a.slot_active[slot]) return;
    int *si = a.state_i;
#define SI(f) si[(size_t)(f) * nc + slot]
    nvx_fsm_regs r;
    r.so = SI(DI_SYNCED) ? SI(DI_SYNC_OFF) : NVX_FSM_UNSYNCED;
    r.nso = SI(DI_NEXT_SYNC_OFF);
    r.phase1 = SI(DI_PHASE) + 1;
    r.prev_offset = SI(DI_PREV_OFFSET);

    // decoded bits are packed ('B' = 1, LSB first) and stored one 32-bit word at a
    // time: byte stores would sit in front of every prefetched load in the in-order
    // vmcnt queue
    unsigned *bits = (unsigned *)(a.bits + (size_t)slot * a.bits_cap);
    const int cap_words = a.bits_cap / 4;
    unsigned long long acc = 0;                   // pending bits, LSB first
    int nacc = 0, nwords = 0;                     // bits pending in acc (< 64), words already stored
    const int periods = n3 / 9;                   // whole frames (288 * frames), except in the launch that ends the stream
    // one row of 16-bit words per chain: eight bit periods per 16-byte load, requested one group ahead
    const unsigned short *wrow = a.words + (size_t)slot * (a.y3_cap / 9);
    const uint4 *words = (const uint4 *)wrow;
    uint4 wnext = words[0];

    int m0 = 0;
    for (; m0 + 8 <= periods; m0 += 8) {
        const uint4 wv = wnext;
        if (m0 + 16 <= periods) wnext = words[m0 / 8 + 1];
        const unsigned wcur[8] = { wv.x & 0xffffu, wv.x >> 16, wv.y & 0xffffu, wv.y >> 16,
                                   wv.z & 0xffffu, wv.z >> 16, wv.w & 0xffffu, wv.w >> 16 };
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int n;
            const unsigned b = nvx_fsm_period(s_tab, wcur[i], &r, &n);
            acc |= (unsigned long long)(b & ((1u << n) - 1u)) << nacc;
            nacc += n;
        }
        // at most 10 bits per 8 periods: one store check per group
        if (nacc >= 32) {
            if (nwords < cap_words) bits[nwords] = (unsigned)acc;
            nwords++; acc >>= 32; nacc -= 32;
        }
    }
    // the end of a stream (nvx_finish): the whole periods that do not fill a group of eight, then the samples of the ragged
    // last period one at a time, by the per-sample rule the table is generated from (nvx_fsm.h; decoder.C:62-137, 202-249)
    for (; m0 < periods; m0++) {
        int n;
        const unsigned b = nvx_fsm_period(s_tab, wrow[m0], &r, &n);
        acc |= (unsigned long long)(b & ((1u << n) - 1u)) << nacc;
        nacc += n;
        if (nacc >= 32) { if (nwords < cap_words) bits[nwords] = (unsigned)acc; nwords++; acc >>= 32; nacc -= 32; }
    }
    if (const int rem = n3 - 9 * periods) {
        const unsigned w = wrow[periods];
        int n;
        const unsigned b = nvx_fsm_partial_period(w, rem, &r, &n);
        acc |= (unsigned long long)(b & ((1u << n) - 1u)) << nacc;
        nacc += n;
        if (nacc >= 32) { if (nwords < cap_words) bits[nwords] = (unsigned)acc; nwords++; acc >>= 32; nacc -= 32; }
    }
    if (nacc > 0 && nwords < cap_words) bits[nwords] = (unsigned)acc;

    a.nbits[slot] = nwords * 32 + nacc;
    const int synced = r.so != NVX_FSM_UNSYNCED;
    SI(DI_SYNCED) = synced; SI(DI_SYNC_OFF) = synced ? r.so : 0; SI(DI_NEXT_SYNC_OFF) = r.nso;
    SI(DI_PHASE) = r.phase1 - 1; SI(DI_PREV_OFFSET) = r.prev_offset;
#undef SI
}

extern "C" hipError_t nvx_launch_demod_front(const nvx_demod_args *a, hipStream_t s)
{
    const unsigned chains = a->part ? (unsigned)(2 * a->per_part * a->n_part) : (unsigned)a->n_slots;
    // Few chains and a long launch: one workgroup per tile instead of one per chain (NVX_DEMOD_TILES=0/1 forces the
    // choice: tests, A/B runs).  Otherwise the walk: the tile form does 2.3 x the arithmetic in workgroups of 29 KB of
    // LDS, which find no room on a CU beside the persistent grid of the NEXT cascade launch -- the demodulator runs
    // beside it -- once that grid fills the chip (64 streams x 62 frames: step 3.1 ms with tiles, cascade + demodulator
    // one after the other).  So: only while the launch's own cascade units (at most chains x frames) leave the chip
    // at least half empty.
    static const int force = getenv("NVX_DEMOD_TILES") ? atoi(getenv("NVX_DEMOD_TILES")) : -1;
    const int tiles = (a->n3 + DTL - 1) / DTL;
    const long long chain_frames = (long long)chains * (a->n3 / NVX_Y3_PER_FRAME);
    // (the tile form works on a.n3: every chain whole frames.  A launch that ends streams is one frame long: the walk.)
    const bool parallel = tiles >= 3 && (force >= 0 ? force != 0 : chain_frames <= 2560);
    if (parallel) {
        const int wgs = tiles - 2;                           // one workgroup per tile from the third on; the head walks the first two
        hipLaunchKernelGGL(nvx_demod_front_head, dim3(chains), dim3(NVX_FRONT_THREADS), 0, s, *a);
        hipLaunchKernelGGL(nvx_demod_front_tiles, dim3(chains * (unsigned)wgs), dim3(NVX_FRONT_THREADS), 0, s, *a, wgs);
    } else {
        hipLaunchKernelGGL(nvx_demod_front, dim3(chains), dim3(NVX_FRONT_THREADS), 0, s, *a);
    }
    return hipGetLastError();
}

extern "C" hipError_t nvx_launch_demod_fsm(const nvx_demod_args *a, hipStream_t s)
{
    const int chains = a->part ? 2 * a->per_part * a->n_part : a->n_slots;
    hipLaunchKernelGGL(nvx_demod_fsm, dim3((unsigned)((chains + 63) / 64)), dim3(64), 0, s, *a);
    return hipGetLastError();
}

// host-callable copy of the device atan2, for tests (tests/test_atan2.py)
extern "C" __attribute__((visibility("default"))) double nvx_atan2_host(double y, double x) { return nvx_atan2(y, x); }
