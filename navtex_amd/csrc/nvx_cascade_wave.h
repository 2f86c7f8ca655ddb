// nvx_cascade_wave.h -- the 252 kS/s cascade as ONE WAVE executes it (device code, gfx950): LDS map, carried state,
// and one pass of FIR1 -> mixer -> FIR2 -> FIR3 over 256 new samples that already sit in the wave's polyphase window.
// Shared by the two kernels that feed that window differently:
//   nvx_fir_cascade   (nvx_cascade.hip)        one wave per work unit; stage 0 or the int16 conversion fills the window
//   nvx_wideband_fused (nvx_wideband_fused.hip) eight waves per work unit; the channeliser fills all eight windows
//
//        FIR1 37 taps /4        receiver/fir1cpp.C:80-136
//        mixer +-14 kHz         receiver/fir2cpp.C:112-128
//        FIR2 47 taps /7        receiver/fir2cpp.C:131-215
//        FIR3 71 taps /10       receiver/fir3cpp.C:22-60
//
// Arithmetic contract (what makes results bit-identical to the reference's x86-64 build): every FIR output is
// accumulated by ONE lane, acc = 0.0 then acc = acc + h[i]*x in tap order, product and sum rounded separately
// (-ffp-contract=off), I and Q independently, fp64 throughout.
#ifndef NVX_CASCADE_WAVE_H
#define NVX_CASCADE_WAVE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nvx_tables.h"
#include "nvx_kernels.h"
#include "nvx_device.h"

// ------------------------------------------------------------------ LDS map (per wave)
// X: eight polyphase arrays P_r[e] of 37 double2: entry e of phase r holds sample 8*(e - 5) + r of the pass
//    (5 history entries: a lane reaches back 33 samples; 32 new ones).  37 is odd and = 5 mod 8, which spreads
//    both write patterns (stage 0: ds_write_b64, lane pairs walk the phases; 252 kS/s input: ds_write_b128,
//    lane parity picks the phase quartet) over all banks.
// U[c]:  mixer output buffer, 46 history + pending (struct Geo: at most 256 / 160)
// Y2[c]: FIR2 output buffer, 70 history + one FIR3 batch
// MIX:   2 signs x 2 periods x 9 x (cos, -+sin): index mixbase + (o mod 9) + 1 <= 17 needs no wrap, and the lane's
//        sign of the cross product (step 4) is part of its table address
#define XPH 8
#define XH 5
#define XS 37
#define X_ENTRIES (XPH * XS)
#define NVX_Y2_RUN 160                    /* single-chain kernel: FIR2 outputs per FIR3 run (16 outputs x {I,Q} = 32 lanes) */

// Batching geometry.  One chain: FIR2 runs on 224 pending mixer outputs (32 outputs x {I,Q} = 64
// lanes), FIR3 on NVX_Y2_RUN pending FIR2 outputs.  Two chains: both chains share a run
// (lane = chain x output x component), so half the batch fills the wave and the pending
// buffers -- and with them the LDS footprint -- halve: 17.4 KB instead of 24 KB, 9 instead of 6
// waves per CU.  Either way a frame (20160 / 2880 outputs) is a whole number of runs.
// Fill levels.  A pass adds 64 mixer outputs, a FIR2 run takes U_RUN of them as soon as that many are pending: the
// pending count walks 64, 128, 192, 256 -> 32, 96, 160, 224 -> 0 (one chain; a pre-roll starts the walk at 96) or
// 64, 128 -> 16, 80, 144 -> 32, ... 112 -> 0 (two chains), so at most U_PEND_MAX are pending before a run and at most
// U_LEFT_MAX after it.  FIR2 outputs collect to exactly Y2_RUN before FIR3 takes them all.
template <int NCH> struct Geo;
template <> struct Geo<1> { static constexpr int U_RUN = 224, Y2_PER_RUN = 32, Y2_RUN = NVX_Y2_RUN, Y3_PER_RUN = NVX_Y2_RUN / 10, U_PEND_MAX = 256, U_LEFT_MAX = 32; };
template <> struct Geo<2> { static constexpr int U_RUN = 112, Y2_PER_RUN = 16, Y2_RUN = 80, Y3_PER_RUN = 8, U_PEND_MAX = 160, U_LEFT_MAX = 48; };
template <int NCH> struct GeoSizes {
    static constexpr int U_ENTRIES = ((46 + Geo<NCH>::U_PEND_MAX) + 7) / 8 * 8;
    static constexpr int Y2_ENTRIES = ((70 + Geo<NCH>::Y2_RUN) + 7) / 8 * 8;
};
// FIR3 takes its batch the moment it is complete and leaves nothing pending (the slide behind it keeps the 70 history
// entries only): a batch must be a whole number of FIR2 runs
static_assert(Geo<1>::Y2_RUN % Geo<1>::Y2_PER_RUN == 0 && Geo<2>::Y2_RUN % Geo<2>::Y2_PER_RUN == 0, "NVX_Y2_RUN: a whole number of FIR2 runs (160)");
static_assert(Geo<1>::U_RUN % 32 == 0 && Geo<2>::U_RUN % 16 == 0 && 46 + Geo<1>::U_LEFT_MAX <= 128 && 46 + Geo<2>::U_LEFT_MAX <= 128, "slide covers two rows of 64");

// F3IN: FIR3 runs inside the wave (the single-wave cascade kernels) and its input buffer Y2 lives here; without it (the
// fused wideband kernel, r4) the wave ends at FIR2, whose outputs go straight to HBM (nvx_kernels.h, nvx_fir3.hip)
template <int NCH, bool F3IN> struct CascadeLdsY2 { double2 Y2[NCH][GeoSizes<NCH>::Y2_ENTRIES]; };
template <int NCH> struct CascadeLdsY2<NCH, false> {};
template <int NCH, bool F3IN = true>
struct CascadeLds : CascadeLdsY2<NCH, F3IN> {
    double2 X[X_ENTRIES];
    double2 U[NCH][GeoSizes<NCH>::U_ENTRIES];
    double2 mix[2][2 * NVX_MIX_N];      // [sign of the cross term][two periods of (cos, -+sin)]
};

__device__ __forceinline__ int dpp_swap_pairs(int v)
{
    // quad_perm [1,0,3,2]: every lane reads its lane^1 neighbour
    return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ double dpp_swap_pairs_f64(double v)
{
    return __hiloint2double(dpp_swap_pairs(__double2hiint(v)), dpp_swap_pairs(__double2loint(v)));
}

// FIR1 window: s_j = x[8*half + 7 - j] is component comp of X[r * XS + XH + half + fl] with 7 - j = 8 * fl + r;
// offset in doubles from the lane's base pointer
#define NVX_F1_GROUP 4                    /* LDS reads per wait */
#define NVX_F1_AHEAD 3                    /* groups in flight ahead of the arithmetic */
#define NVX_F23_AHEAD 12                  /* FIR2 / FIR3: taps read ahead */
__device__ __forceinline__ constexpr int f1_offset(int j)
{
    const int t = 7 - j, r = t & 7, fl = (t - r) / 8;
    return 2 * (r * XS + fl);
}

// The FIR1 taps live in VGPRs for the whole kernel (22 distinct values = 44 registers).  As literals they cost two
// s_mov_b32 per tap per pass (74 scalar instructions beside 154 fp64 ones: the 252 kS/s kernels are bound by the
// instruction issue slots of their few resident waves, profiles/r02) and their SGPR pressure makes the compiler
// park loop invariants in VGPR lanes (v_readlane / v_writelane in the pass loop).
template <int N> struct TapIndex {
    int first[N];
    constexpr TapIndex(const double (&h)[N]) : first{}
    {
        for (int i = 0; i < N; i++) {
            int f = i;
            for (int k = 0; k < i; k++) if (h[k] == h[i]) { f = k; break; }
            first[i] = f;
        }
    }
};
static constexpr TapIndex<NVX_T1> NVX_H1_FIRST(NVX_H1);

// The state block is the only memory one unit writes and another unit (usually on another XCD,
// behind another L2) reads within a launch.  Every access to it is an agent-scope relaxed atomic
// (global_load / global_store ... sc1: coherent at the device level per instruction), so the
// hand-off needs no whole-cache maintenance: a release / acquire FENCE at agent scope costs an L2
// write-back (buffer_wbl2 sc1) and an L1 + L2 invalidate (buffer_inv sc1) per unit on gfx950, paid by
// every wave that shares the XCD.
__device__ __forceinline__ double2 state_load(const double2 *p)
{
    double2 r;
    r.x = __hip_atomic_load(&p->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.y = __hip_atomic_load(&p->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return r;
}
__device__ __forceinline__ void state_store(double2 *p, double2 v)
{
    __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the state block's seal (nvx_kernels.h): every 8-byte pattern a unit stores, rotated by its position in the block
// and XOR-folded over the wave, mixed with the block's tag (stream, position of the stream in thirds of a frame).
// Position = (slot of the entry in the lane's list: a compile-time rotation) and (lane: the rotations of the butterfly
// below), added mod 64: what the word is FOR is a stale or torn block -- a block of another position, stream or launch
// (the tag), a block of which only part arrived -- and any single changed bit; it is not a cryptographic permutation
// check (two words whose slot and lane rotations add up to the same total could be swapped unnoticed, and FIR3's history
// of a wideband handle and the demodulator's carried state lie outside it: they only ever cross kernel boundaries).
// No lane-dependent shift counts, which the compiler would keep in VGPRs across the whole kernel (the
// headline kernel sits 9 registers under its occupancy limit).
template <int R> __device__ __forceinline__ unsigned long long seal_rotc(unsigned long long v)
{
    constexpr int r = R & 63;
    if constexpr (r == 0) return v;
    else return (v << r) | (v >> (64 - r));
}
template <int SLOT> __device__ __forceinline__ unsigned long long seal_pair(double2 v)
{
    return seal_rotc<14 * SLOT + 3>((unsigned long long)__double_as_longlong(v.x)) ^ seal_rotc<14 * SLOT + 10>((unsigned long long)__double_as_longlong(v.y));
}
__device__ __forceinline__ unsigned long long seal_tag(int stream, unsigned third)
{
    unsigned long long t = ((unsigned long long)(unsigned)stream << 32) | third;
    t *= 0x9E3779B97F4A7C15ull; t ^= t >> 29; t *= 0xBF58476D1CE4E5B9ull; t ^= t >> 32;
    return t;
}
// A wave-uniform XOR over the lanes l of rot(v_l, r(l)) with r distinct for every lane.  Within a row of 16 lanes four
// DPP exchanges, each rotating what it takes in: lane ^ 1 (by 1), lane ^ 2 (by 2), mirror of the half row (by 4), mirror
// of the row (by 8) -- lane 0 of a row ends with every lane of the row under its own rotation 0..15 (the subset sums of
// 1, 2, 4, 8 along the lane's path); the four rows meet in scalar registers under rotations 0, 16, 32, 48.  DPP moves and
// v_readlane take immediates: no index register lives in a VGPR across the kernel (as __shfl_xor's would), and no trip
// through the LDS crossbar sits between a unit's state loads and its first pass.  All lanes active.
template <int CTRL, int R> __device__ __forceinline__ unsigned long long seal_step(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
    return v ^ seal_rotc<R>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned long long seal_row(unsigned lo, unsigned hi, int lane0)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, lane0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, lane0);
}
__device__ __forceinline__ unsigned long long wave_fold64(unsigned long long v)
{
    v = seal_step<0xB1, 1>(v);           // quad_perm [1,0,3,2]
    v = seal_step<0x4E, 2>(v);           // quad_perm [2,3,0,1]
    v = seal_step<0x141, 4>(v);          // row_half_mirror
    v = seal_step<0x140, 8>(v);          // row_mirror
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    return seal_row(lo, hi, 0) ^ seal_rotc<16>(seal_row(lo, hi, 16)) ^ seal_rotc<32>(seal_row(lo, hi, 32)) ^ seal_rotc<48>(seal_row(lo, hi, 48));
}
// the seal entry: { fold ^ seal_tag, the tag in clear (diagnostics) }
__device__ __forceinline__ void seal_store(double2 *st, unsigned long long fold, int stream, unsigned third)
{
    unsigned long long *p = (unsigned long long *)(st + NVX_STATE_SEAL);
    __hip_atomic_store(p, fold ^ seal_tag(stream, third), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, ((unsigned long long)(unsigned)stream << 32) | third, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the consumer's side: the seal word is requested together with the block (every lane loads the same word) ...
__device__ __forceinline__ unsigned long long seal_load(const double2 *st_in)
{
    return __hip_atomic_load((const unsigned long long *)(st_in + NVX_STATE_SEAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and judged once the fold over what was loaded is known: true when `got` is the seal of (fold, stream, third)
__device__ __forceinline__ bool seal_ok(unsigned long long got, unsigned long long fold, int stream, unsigned third)
{
    const unsigned long long d = got ^ fold ^ seal_tag(stream, third);
    return __builtin_amdgcn_readfirstlane((int)((unsigned)d | (unsigned)(d >> 32))) == 0;
}

// One wave's share of the cascade.  lane = 2 * (pair / output index) + component, in stage 0, FIR1, the mixer, FIR2
// and FIR3 alike.
template <int NCH, bool F3IN = true>
struct CascadeWave {
    CascadeLds<NCH, F3IN> *lds;
    int lane, half, comp;
    const lds_vdouble *xrv;              // FIR1 read base: sample 8*half + t is component comp of X[(t & 7) * XS + XH + half + floor(t / 8)]
    const lds_vd2 *mixrow;               // this lane's row of the mixer table (set per unit)
    int lane_mod9;
    double h1v[NVX_T1];
    // per unit
    unsigned mask;                       // chains of the stream that are decoded
    int chain_of_slot0;                  // NCH == 1: the single active chain; NCH == 2: chain slot c is chain c
    int n_u, n_y2, n3_done, mixbase;
    nvx_d2 mix_next;                     // mixer table entry of the NEXT pass's first output (= this pass's second one)
    bool emit;                           // outputs are written (false during a pre-roll)
    double2 *y3; size_t y3_row0, y3_cap;
    // !F3IN: where this lane's FIR2 outputs go -- the unit's first output of the lane's chain in the y2 buffer (nullptr: the
    // chain is not decoded) -- and how many outputs per chain the unit has written so far
    double2 *y2_lane; int n2_done;

    // once per kernel: lane constants, the taps, the mixer table of this wave's LDS block
    __device__ __forceinline__ void init(CascadeLds<NCH, F3IN> *l, int lane_)
    {
        lds = l; lane = lane_; half = lane >> 1; comp = lane & 1;
        xrv = (const lds_vdouble *)((const double *)&lds->X[XH + half] + comp);
        lane_mod9 = (2 * half) % 9;
        if (lane < 4 * NVX_MIX_N) {
            // constant-index selects keep the tables out of scratch
            const int j9 = lane % NVX_MIX_N;
            double cr = 0.0, ci = 0.0;
#pragma unroll
            for (int j = 0; j < NVX_MIX_N; j++) if (j9 == j) { cr = NVX_MIX_CR[j]; ci = NVX_MIX_CI[j]; }
            lds->mix[0][lane % (2 * NVX_MIX_N)] = double2{ cr, ci };      // both copies are written by two lanes each: same value
            lds->mix[1][lane % (2 * NVX_MIX_N)] = double2{ cr, -ci };
        }
#pragma unroll
        for (int i = 0; i < NVX_T1; i++)
            if (NVX_H1_FIRST.first[i] == i) { h1v[i] = NVX_H1[i]; asm volatile("" : "+v"(h1v[i])); }
    }

    // once per unit: which chains (state_in / state_out need no more than this) ...
    __device__ __forceinline__ void set_mask(unsigned chain_mask)
    {
        mask = chain_mask;
        chain_of_slot0 = (NCH == 1) ? ((mask & 1u) ? 0 : 1) : 0;
    }
    // ... and the rest, once it is known whether the unit starts from carried state or from a pre-roll
    __device__ __forceinline__ void begin_unit(unsigned chain_mask, double2 *y3_, size_t row0, size_t cap, int mixbase0, int n_u0, int n_y20, bool emit0)
    {
        set_mask(chain_mask);
        // mixer table row of this lane: its cross term carries the sign of the 518 chain (I lanes negated) -- or, when the
        // unit's only chain is the 490 one, of that chain (Q lanes negated); see step 4
        mixrow = (const lds_vd2 *)&lds->mix[(comp ^ (NCH == 1 ? chain_of_slot0 : 0)) ? 0 : 1][lane_mod9];
        y3 = y3_; y3_row0 = row0; y3_cap = cap;
        mixbase = mixbase0; n_u = n_u0; n_y2 = n_y20; n3_done = 0; n2_done = 0; emit = emit0;
        mix_next = mixrow[mixbase];      // (behind init()'s table writes in this wave's LDS queue)
    }

    // !F3IN, once per unit: the y2 buffer of the stream's parity, the rows of its two chains (-1: not decoded), the unit's
    // first output within a row
    __device__ __forceinline__ void begin_unit_y2(double2 *y2buf, size_t pitch, int row0, int row1, size_t first)
    {
        const int ch = (NCH == 1) ? chain_of_slot0 : (lane >> 5);           // the chain this lane computes FIR2 outputs of
        const int row = ch ? row1 : row0;
        y2_lane = row < 0 ? nullptr : y2buf + ((size_t)row * pitch + NVX_Y2_PREFIX + first);
    }

    // filter histories from the state block: 36 newest 252 kS/s samples (oldest first), then per chain 46 mixer outputs
    // and 70 FIR2 outputs.  Returns this lane's share of the seal's fold over what it LOADED (nvx_kernels.h).
    __device__ __forceinline__ unsigned long long state_in(const double2 *st_in)
    {
        unsigned long long f = 0;
        if (lane < 36) {
            const int v = lane + 4;                    // sample -36+lane = 8*((v>>3) - XH) + (v&7)
            const double2 x = state_load(st_in + lane);
            f ^= seal_pair<0>(x);
            lds->X[(v & 7) * XS + (v >> 3)] = x;
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int ch = (NCH == 1) ? chain_of_slot0 : c;
            const double2 *su = st_in + 36 + ch * (46 + 70);
            if (lane < 46) { const double2 u = state_load(su + lane); f ^= (c ? seal_pair<4>(u) : seal_pair<1>(u)); lds->U[c][lane] = u; }
            if constexpr (F3IN) {
                const double2 y = state_load(su + 46 + lane);
                f ^= (c ? seal_pair<5>(y) : seal_pair<2>(y));
                lds->Y2[c][lane] = y;
                if (lane < 6) { const double2 t = state_load(su + 46 + 64 + lane); f ^= (c ? seal_pair<6>(t) : seal_pair<3>(t)); lds->Y2[c][64 + lane] = t; }
            }
        }
        return f;
    }
    // ... or silence in front of a pre-roll (independent units)
    __device__ __forceinline__ void state_zero()
    {
        const double2 zero = { 0.0, 0.0 };
        if (lane < XPH * XH) lds->X[(lane & 7) * XS + (lane >> 3)] = zero;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            for (int i = lane; i < 46 + NVX_PREROLL_U; i += 64) lds->U[c][i] = zero;
            if constexpr (F3IN) for (int i = lane; i < 70 + NVX_PREROLL_Y2; i += 64) lds->Y2[c][i] = zero;
        }
    }
    // ... and back; returns this lane's share of the fold over what it STORED
    __device__ __forceinline__ unsigned long long state_out(double2 *st)
    {
        unsigned long long f = 0;
        if (lane < 36) {
            const int v = lane + 4;
            const double2 x = lds->X[(v & 7) * XS + (v >> 3)];
            f ^= seal_pair<0>(x);
            state_store(st + lane, x);
        }
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const int ch = (NCH == 1) ? chain_of_slot0 : c;
            double2 *su = st + 36 + ch * (46 + 70);
            if (lane < 46) { const double2 u = lds->U[c][lane]; f ^= (c ? seal_pair<4>(u) : seal_pair<1>(u)); state_store(su + lane, u); }
            if constexpr (F3IN) {
                const double2 y = lds->Y2[c][lane];
                f ^= (c ? seal_pair<5>(y) : seal_pair<2>(y));
                state_store(su + 46 + lane, y);
                if (lane < 6) { const double2 t = lds->Y2[c][64 + lane]; f ^= (c ? seal_pair<6>(t) : seal_pair<3>(t)); state_store(su + 46 + 64 + lane, t); }
            }
        }
        return f;
    }

    // One pass: the window holds 256 new samples (entries XH .. XH+31 of every phase) behind its history.
    __device__ __forceinline__ void compute_pass()
    {
        // ---- FIR1: y1[o] = sum_i h1[i] * x[4o+3-i], outputs o = 2*half and 2*half+1 of component comp.
        // With s_j = x[8*half + 7 - j]:  y1[2*half+1] = sum_j h1[j] * s_j (j = 0..36),
        //                               y1[2*half]   = sum_j h1[j-4] * s_j (j = 4..40): both in tap order.
        // The reads run NVX_F1_AHEAD groups ahead of the arithmetic, so the LDS latency is covered by the wave's own
        // fp64 work.  volatile: each read stays a ds_read_b64 (512 contiguous bytes per wave, 2 LDS cycles); merged
        // into ds_read2_b64 a pair would cost 8 (MI355X_MICROARCH.md, LDS table).
        // In front of them go the two reads whose results are only needed after FIR1 -- the mixer's table entries and
        // the tail of the new samples that becomes the next pass's history -- so that neither costs a round trip
        // through the LDS with the wave idle.
        // (the lane's two outputs use table entries m and m + 1, the next pass's m + 1 and m + 2: one new entry per pass)
        const nvx_d2 c0 = mix_next, c1 = mixrow[mixbase + 1];
        mix_next = c1;
        nvx_d2 tail = { 0.0, 0.0 };
        if (lane < XPH * XH) tail = *(const lds_vd2 *)&lds->X[(lane & 7) * XS + 32 + (lane >> 3)];
        double xs[NVX_T1 + 4];
        constexpr int F1N = NVX_T1 + 4, F1NG = (F1N + NVX_F1_GROUP - 1) / NVX_F1_GROUP;
#pragma unroll
        for (int j = 0; j < NVX_F1_AHEAD * NVX_F1_GROUP; j++) xs[j] = xrv[f1_offset(j)];
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int g = 0; g < F1NG; g++) {
            // reads of group g + AHEAD go out behind the arithmetic of group g - 1 ...
            if (g + NVX_F1_AHEAD < F1NG) {
                NVX_PIN_AFTER(a1);
#pragma unroll
                for (int j = (g + NVX_F1_AHEAD) * NVX_F1_GROUP; j < (g + NVX_F1_AHEAD + 1) * NVX_F1_GROUP && j < F1N; j++) xs[j] = xrv[f1_offset(j)];
            }
            // ... and one wait covers a whole group (LDS reads return in order): its first value "depends" on its last
            const int lo = g * NVX_F1_GROUP, hi = (lo + NVX_F1_GROUP < F1N ? lo + NVX_F1_GROUP : F1N) - 1;
            if (hi > lo) asm volatile("" : "+v"(xs[lo]) : "v"(xs[hi]));
#pragma unroll
            for (int j = lo; j <= hi; j++) {
                if (j < NVX_T1) a1 += h1v[NVX_H1_FIRST.first[j]] * xs[j];
                if (j >= 4) a0 += h1v[NVX_H1_FIRST.first[j - 4]] * xs[j];
            }
        }
        // ---- mixer, table index (o mod 9), o counted from the frame start
        // (a frame is 20160 = 9 * 2240 FIR1 outputs, so that equals o from stream start).
        // 518 chain (fir2cpp.C:116-117): (I*cr - Q*ci, I*ci + Q*cr); 490 chain (:122-123): (I*cr + Q*ci, -I*ci + Q*cr).
        // A lane owns one component of its two outputs and gets the other from its partner lane (DPP pair swap).
        // With A = mine*cr and B = other*ci, both chains' results are A + B or A - B (a sum commutes exactly and
        // (-I)*ci = -(I*ci) exactly): 518 -> I lane A - B, Q lane A + B; 490 -> I lane A + B, Q lane A - B.
        {
            const double o0 = dpp_swap_pairs_f64(a0), o1 = dpp_swap_pairs_f64(a1);
            const double A0 = a0 * c0.x, A1 = a1 * c1.x;
            // the lane's table row holds +-ci: B carries the sign of the 518 chain (of the unit's only chain when
            // NCH == 1); with two chains the 490 one takes the opposite sign
            const double B0 = o0 * c0.y, B1 = o1 * c1.y;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int ch = (NCH == 1) ? chain_of_slot0 : c;
                double *uw = (double *)&lds->U[c][46 + n_u + 2 * half] + comp;
                uw[0] = (NCH == 1 || ch == 0) ? A0 + B0 : A0 - B0;
                uw[2] = (NCH == 1 || ch == 0) ? A1 + B1 : A1 - B1;
            }
        }
        n_u += 64;
        mixbase += 1; if (mixbase == NVX_MIX_N) mixbase = 0;      // 64 mod 9 == 1
        // ---- the 5-deep history of each phase moves to the front
        NVX_WAVE_LDS_FENCE();
        if (lane < XPH * XH) *(lds_vd2 *)&lds->X[(lane & 7) * XS + (lane >> 3)] = tail;
        NVX_WAVE_LDS_FENCE();

        // ---- FIR2 when a batch of mixer outputs is pending
        constexpr int U_RUN = Geo<NCH>::U_RUN, Y2_PER_RUN = Geo<NCH>::Y2_PER_RUN;
        constexpr int Y2_RUN = Geo<NCH>::Y2_RUN, Y3_PER_RUN = Geo<NCH>::Y3_PER_RUN;
        // lane -> (chain slot, output, component): one chain uses all 64 lanes for 32 outputs,
        // two chains put chain 0 on lanes 0-31 and chain 1 on lanes 32-63 (16 outputs each)
        const int f2c = (NCH == 2) ? (lane >> 5) : 0;
        const int f2o = (NCH == 2) ? ((lane >> 1) & 15) : half;
        constexpr int UT1 = 46 + Geo<NCH>::U_LEFT_MAX - 64;        // entries of the second row that can survive a run
        while (n_u >= U_RUN) {
            // the pending entries behind this run move to the front afterwards: read them now, write them after the FIR
            nvx_d2 ut0[NCH], ut1[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                ut0[c] = *(const lds_vd2 *)&lds->U[c][U_RUN + lane];
                ut1[c] = *(const lds_vd2 *)&lds->U[c][U_RUN + 64 + (lane < UT1 ? lane : UT1 - 1)];
            }
            {
                const lds_vdouble *ub = (const lds_vdouble *)((const double *)&lds->U[f2c][7 * f2o] + comp);
                double xs2[NVX_T2], acc = 0.0;                // reads run ahead of the arithmetic, as in FIR1
#pragma unroll
                for (int i = 0; i < NVX_F23_AHEAD; i++) xs2[i] = ub[2 * (52 - i)];
                nvx_static_for<0, NVX_T2>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (i + NVX_F23_AHEAD < NVX_T2) { NVX_PIN_AFTER(acc); xs2[i + NVX_F23_AHEAD] = ub[2 * (52 - (i + NVX_F23_AHEAD))]; }
                    acc += NVX_H2[i] * xs2[i];
                });
                if constexpr (F3IN) {
                    if (NCH == 1 || ((mask >> f2c) & 1u)) ((double *)&lds->Y2[f2c][70 + n_y2 + f2o])[comp] = acc;
                } else {
                    // the 9 kS/s output leaves the wave here: 32 (16 per chain) consecutive {I,Q} pairs, 512 contiguous bytes
                    if (emit && y2_lane) ((double *)(y2_lane + n2_done + f2o))[comp] = acc;
                }
            }
            NVX_WAVE_LDS_FENCE();
            // drop the consumed inputs: keep 46 history + pending (<= 46 + U_LEFT_MAX entries)
            const int keep = 46 + n_u - U_RUN;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (lane < keep) *(lds_vd2 *)&lds->U[c][lane] = ut0[c];
                if (lane + 64 < keep) *(lds_vd2 *)&lds->U[c][64 + lane] = ut1[c];
            }
            NVX_WAVE_LDS_FENCE();
            n_u -= U_RUN;
            if constexpr (!F3IN) { if (emit) n2_done += Y2_PER_RUN; continue; }
            n_y2 += Y2_PER_RUN;

            // ---- FIR3 when a batch of FIR2 outputs is pending
            if constexpr (F3IN) if (n_y2 >= Y2_RUN) {
                // lanes 0 .. 2*Y3_PER_RUN-1 hold chain 0 (output, component); with two chains the
                // next 2*Y3_PER_RUN lanes hold chain 1
                const int f3c = (NCH == 2) ? ((lane >> 4) & 1) : 0;
                const int f3o = half & (Y3_PER_RUN - 1);
                const bool f3live = lane < 2 * Y3_PER_RUN * NCH;
                nvx_d2 yt0[NCH], yt1[NCH];
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    yt0[c] = *(const lds_vd2 *)&lds->Y2[c][Y2_RUN + lane];
                    yt1[c] = *(const lds_vd2 *)&lds->Y2[c][Y2_RUN + 64 + (lane < 6 ? lane : 5)];      // 70 - 64 entries survive
                }
                {
                    const int ch = (NCH == 1) ? chain_of_slot0 : f3c;
                    const lds_vdouble *yb = (const lds_vdouble *)((const double *)&lds->Y2[f3c][10 * f3o] + comp);
                    double xs3[NVX_T3], acc = 0.0;
#pragma unroll
                    for (int i = 0; i < NVX_F23_AHEAD; i++) xs3[i] = yb[2 * (79 - i)];
                    nvx_static_for<0, NVX_T3>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        if constexpr (i + NVX_F23_AHEAD < NVX_T3) { NVX_PIN_AFTER(acc); xs3[i + NVX_F23_AHEAD] = yb[2 * (79 - (i + NVX_F23_AHEAD))]; }
                        acc += NVX_H3[i] * xs3[i];
                    });
                    if (emit && f3live && (NCH == 1 || ((mask >> f3c) & 1u))) {
                        double *out = (double *)(y3 + (y3_row0 + (size_t)ch * y3_cap + n3_done + f3o));
                        out[comp] = acc;
                    }
                }
                NVX_WAVE_LDS_FENCE();
                const int keep3 = 70 + n_y2 - Y2_RUN;           // = 70: FIR3 runs the moment its batch is complete
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    if (lane < keep3) *(lds_vd2 *)&lds->Y2[c][lane] = yt0[c];
                    if (lane + 64 < keep3) *(lds_vd2 *)&lds->Y2[c][64 + lane] = yt1[c];
                }
                NVX_WAVE_LDS_FENCE();
                n_y2 -= Y2_RUN;
                n3_done += Y3_PER_RUN;
            }
        }
    }
};

#endif
