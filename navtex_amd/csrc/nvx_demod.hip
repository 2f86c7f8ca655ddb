// nvx_demod.hip -- the demodulator kernels (gfx950): nvx_demod_front + nvx_demod_fsm, 900 S/s -> 'B'/'Y' bits
//        discriminator          receiver/decoder.C:42-59
//        bit-timing filter      receiver/decoder.C:142-255
//        mark/space decision    receiver/decoder.C:73-137
// Compiled with -ffp-contract=off: the only v_fma_f64 in the ISA are the explicit error-free
// transformations of nvx_atan2 and the expansion of IEEE division.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

__global__ __launch_bounds__(NVX_FRONT_THREADS) void nvx_demod_front(nvx_demod_args a)
{
    __shared__ double s_dphi[8 + DTL];
    __shared__ double s_S[8 + DTL];
    __shared__ double s_C[567 + DTL];
    __shared__ unsigned char s_D[DTL];
    __shared__ unsigned s_near, s_evals, s_minm;
    // block -> chain: every slot, or the slots of the launch's participants (nvx_kernels.h, nvx_part: entry e covers the
    // decoded streams per_part * stream .. + per_part - 1, two slots each, and carries their common sample count g0)
    const int tid = threadIdx.x;
    int slot = blockIdx.x;
    unsigned long long g0 = a.g0;
    if (a.part) {
        const int per = 2 * a.per_part, e = blockIdx.x / per;
        slot = a.part[e].stream * per + (blockIdx.x - e * per);
        g0 = a.part[e].g0;
    }
    if (!a.slot_active[slot]) return;                    // uniform over the block
    if (tid == 0) { s_near = 0; s_evals = 0; s_minm = 0x7f800000u; }
    unsigned my_near = 0, my_evals = 0;
    float my_minm = __uint_as_float(0x7f800000u);

    double *st = a.dstate + (size_t)slot * NVX_DEMOD_DOUBLES;
    const double2 *y3 = a.y3 + (size_t)slot * a.y3_cap + a.y3_base;
    double *dphi_out = a.dphi ? a.dphi + (size_t)slot * a.y3_cap + a.y3_base : nullptr;
    const double *hist = st + DS_Y3;                     // read in place: only the first 4 samples need it
    if (tid < 8) { s_dphi[tid] = st[DS_DPHI + tid]; s_S[tid] = st[DS_S + tid]; }
    for (int i = tid; i < 567; i += NVX_FRONT_THREADS) s_C[i] = st[DS_C + i];
    __syncthreads();

    for (int ta = 0; ta < a.n3; ta += DTL) {
        const int tl = min(DTL, a.n3 - ta);
        const unsigned long long gt = g0 + (unsigned long long)ta;       // g of L = 0
        for (int L = tid; L < tl; L += NVX_FRONT_THREADS) {
            const int t = ta + L;
            // ---- discriminator, decoder.C:48-52
            const double2 s = y3[t];
            const double2 p = y3_at(y3, hist, t - 1);
            const double prodReal = s.x * p.x + s.y * p.y;
            const double prodImg  = s.y * p.x - s.x * p.y;
            const double ds = nvx_atan2(prodImg, prodReal);
            s_dphi[8 + L] = ds;
            if (dphi_out) dphi_out[t] = ds;
            // ---- mark/space decision for a window ending here, decoder.C:115-132:
            // float*float product, double*float product, double sum, accumulate in
            // double, round to float -- five samples, filter index 0..4
            float BR = 0.0f, BI = 0.0f, YR = 0.0f, YI = 0.0f;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const double2 w = (i == 4) ? s : ((i == 3) ? p : y3_at(y3, hist, t - 4 + i));
                const float fR = NVX_BF_R[i], fI = NVX_BF_I[i];
                const double sampleR = w.x, sampleI = w.y;
                YR = (float)((double)YR + ((double)((float)sampleR * fR) - sampleI * (double)fI));
                YI = (float)((double)YI + ((double)((float)sampleR * fI) + sampleI * (double)fR));
                BR = (float)((double)BR + ((double)((float)sampleR * fR) + sampleI * (double)fI));
                BI = (float)((double)BI + ((double)((float)(-sampleR) * fI) + sampleI * (double)fR));
            }
            const float Brot = BR * BR + BI * BI;
            const float Yrot = YR * YR + YI * YI;
            s_D[L] = (Brot > Yrot) ? 1 : 0;
        }
        __syncthreads();
        // ---- transition correlator, decoder.C:157-177: mask[i] * dphi[g-8+i], i ascending
        for (int L = tid; L < tl; L += NVX_FRONT_THREADS) {
            if (gt + L >= G_DAB) {
                double temp = 0.0;
#pragma unroll
                for (int i = 0; i < 9; i++) temp += (double)NVX_CORR_MASK[i] * s_dphi[L + i];
                s_C[567 + L] = __builtin_fabs(temp);
            } else {
                s_C[567 + L] = 0.0;                      // never read; keeps the carried state deterministic
            }
        }
        __syncthreads();
        // ---- class sum, decoder.C:181-197: ring positions c, c+9, ... ascending.
        // Position p holds the newest value kappa' <= kappa with kappa' = p (mod 567),
        // i.e. the value d = (kappa - p) mod 567 samples back.
        // The order is a rotation of the time order: with d0 = (kappa - c) mod 567 and jw = d0 / 9 the terms are
        // the samples d0, d0-9, ..., d0-9*jw back (ascending in time, stride 9), then those 558+r, ..., d0+9 back
        // (r = d0 mod 9) -- two runs of one stride-9 walk through the time-ordered buffer, the second one starting
        // 567 entries lower.  t_cb = (g of L = 0) - 574 mod 5103 (= 9 * 567) keeps the index arithmetic in 32 bits.
        const unsigned t_cb = (unsigned)((gt % 5103u + (5103u - G_CB % 5103u)) % 5103u);
        for (int L = tid; L < tl; L += NVX_FRONT_THREADS) {
            if (gt + L >= G_CB) {
                const unsigned u = t_cb + (unsigned)L;               // == g - 574 (mod 5103)
                const unsigned c = u % 9u;
                const int d0 = (int)((u + 566u - c) % 567u);         // (kappa - c) mod 567, kappa = g - 8
                const int jw = d0 / 9;
                const double *run1 = &s_C[567 + L - d0];             // terms j = 0 .. jw
                const double *run2 = run1 - 567;                     // terms j = jw+1 .. 62
                double temp = 0.0;
#pragma unroll
                for (int j = 0; j < 63; j++) temp += (j <= jw ? run1 : run2)[9 * j];
                s_S[8 + L] = temp;
            } else {
                s_S[8 + L] = 0.0;
            }
        }
        __syncthreads();
        // ---- one word per bit period: nine window decisions + the arg-max of the
        // timing evaluation (decoder.C:202-215: csa[i] = S(g-8+i), strict '>' from
        // -1.0 => first maximum wins), which falls on local sample 9m+6
        for (int M = tid; M < tl / 9; M += NVX_FRONT_THREADS) {
            unsigned w = 0;
#pragma unroll
            for (int k = 0; k < 9; k++) w |= (unsigned)s_D[9 * M + k] << k;
            const int L = 9 * M + (G_CSA % 9);
            unsigned max_index = 15;
            if (gt + L >= G_CSA) {
                double temp_max = -1.0, runner_up = -1.0;
                max_index = 0;
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const double v = s_S[L + i];
                    if (v > temp_max) { runner_up = temp_max; temp_max = v; max_index = i; }
                    else if (v > runner_up) runner_up = v;
                }
                // instrumentation only (nvx_tie_stats): how close was that decision?
                if (temp_max > 0.0) {
                    const double margin = temp_max - runner_up;
                    my_evals++;
                    if (margin < temp_max * 0x1p-40) my_near++;
                    my_minm = fminf(my_minm, (float)(margin / temp_max));
                }
            }
            a.words[(size_t)slot * (a.y3_cap / 9) + (ta / 9 + M)] = (unsigned short)(w | (max_index << 12));
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

    if (my_evals) { atomicAdd(&s_near, my_near); atomicAdd(&s_evals, my_evals); atomicMin(&s_minm, __float_as_uint(my_minm)); }
    __syncthreads();
    if (tid == 0 && s_evals) {
        if (s_near) atomicAdd(&a.ties->near_ties, (unsigned long long)s_near);
        atomicAdd(&a.ties->evaluations, (unsigned long long)s_evals);
        atomicMin(&a.ties->min_margin_bits, s_minm);
    }
    if (tid < 4 && a.n3 >= 4) { const double2 l = y3[a.n3 - 4 + tid]; st[DS_Y3 + 2 * tid] = l.x; st[DS_Y3 + 2 * tid + 1] = l.y; }
    if (tid < 8) { st[DS_DPHI + tid] = s_dphi[tid]; st[DS_S + tid] = s_S[tid]; }
    for (int i = tid; i < 567; i += NVX_FRONT_THREADS) st[DS_C + i] = s_C[i];
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
    if (a.part) {                                        // the slots of the launch's participants, as in the front kernel
        const int per = 2 * a.per_part, e = slot / per;
        if (e >= a.n_part) return;
        slot = a.part[e].stream * per + (slot - e * per);
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
    const int periods = a.n3 / 9;                 // launches are whole frames: n3 = 288 * frames
    // one row of 16-bit words per chain: eight bit periods per 16-byte load, requested one group ahead
    const uint4 *words = (const uint4 *)(a.words + (size_t)slot * (a.y3_cap / 9));
    uint4 wnext = words[0];

    for (int m0 = 0; m0 < periods; m0 += 8) {     // periods is a multiple of 32
        const uint4 wv = wnext;
        if (m0 + 8 < periods) wnext = words[m0 / 8 + 1];
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
    hipLaunchKernelGGL(nvx_demod_front, dim3(chains), dim3(NVX_FRONT_THREADS), 0, s, *a);
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
