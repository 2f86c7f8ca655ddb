// nvx_kernels.hip -- gfx950 kernels of the NAVTEX receive path.
//
//   nvx_fir_cascade<RAW, NCH, PFD, NT>   int16 IQ in HBM -> 900 S/s complex fp64 per chain
//        stage 0 (/8 integer, build-owned, RAW only)
//        FIR1 37 taps /4        receiver/fir1cpp.C:80-136
//        mixer +-14 kHz         receiver/fir2cpp.C:112-128
//        FIR2 47 taps /7        receiver/fir2cpp.C:131-215
//        FIR3 71 taps /10       receiver/fir3cpp.C:22-60
//   nvx_demod_front + nvx_demod_fsm      900 S/s -> 'B'/'Y' bits
//        discriminator          receiver/decoder.C:42-59
//        bit-timing filter      receiver/decoder.C:142-255
//        mark/space decision    receiver/decoder.C:73-137
//   nvx_channelise              wideband front-end: 2.016 MS/s -> 8 x 252 kS/s (no reference counterpart)
//   nvx_synth_kernel            deterministic CPFSK test source (no reference counterpart)
//
// Arithmetic contract (what makes results bit-identical to the reference's
// x86-64 build): every FIR output is accumulated by ONE lane, acc = 0.0 then
// acc = acc + h[i]*x in tap order, product and sum rounded separately (this
// file is compiled with -ffp-contract=off; the only v_fma_f64 in the ISA are
// the explicit error-free transformations of nvx_atan2 and the expansion of
// IEEE division), I and Q independently, fp64 throughout.
//
// Design of the cascade kernel (HBM-read bound; no MFMA -- 1-D decimating
// convolutions):
//   * a persistent grid of single-wave workgroups pulls work units (one frame
//     of one stream) from an atomic queue; every input byte is read from HBM
//     exactly once, no halo is re-read; filter histories live in LDS between
//     passes and travel between units / launches through a state block in HBM
//     (agent-scope atomic accesses, no cache-wide fences; or, with few streams,
//     are rebuilt by every unit from a nine-pass pre-roll -- see the comments
//     above the kernel and in nvx_kernels.h);
//   * a pass = 64 FIR1 outputs = 256 samples @252 kS/s = 2048 raw samples =
//     8 KiB: eight fully coalesced 1-KiB global_load_dwordx4 per wave, issued
//     one pass ahead into registers (prefetch) so HBM latency hides behind the
//     fp64 work of the current pass;
//   * stage 0 sums 8 raw samples with SDWA half-word pair adds (sign-extend +
//     add of two samples in one op) and one DPP lane-pair exchange; each lane
//     converts one component to fp64 and writes it to the LDS window;
//   * the 252 kS/s window is kept polyphase-split (4 arrays of {I,Q} doubles)
//     so that lane k's tap reads are consecutive 16-byte words: every
//     ds_read_b128 / ds_write_b64 of FIR1 / stage 0 is bank-conflict free;
//   * FIR2 and FIR3 run on batches of pending outputs sized to fill the wave
//     (struct Geo below); a frame of 32 bit periods = 315 passes is a whole
//     number of every batch, after which every decimation counter, the mixer
//     index and all LDS fill levels are back at zero: the carried state is
//     just the three filter histories.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "nvx_tables.h"
#include "nvx_atan2.h"
#include "nvx_fsm.h"
#include "nvx_synth.h"
#include "nvx_kernels.h"

// A single wave owns all LDS it touches; LDS instructions of one wave execute
// in program order, so cross-lane hand-offs need no s_barrier and no waitcnt --
// only the compiler must be kept from reordering the accesses.
#define NVX_WAVE_LDS_FENCE() asm volatile("" ::: "memory")

typedef short nvx_short2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: nontemporal builtin needs it

// ------------------------------------------------------------------ LDS map
// X: four polyphase arrays of 74 double2 (9 history + 64 new + 1 pad; the pad
//    makes the array stride = 8 banks mod 32 so the stage-0 writes spread)
// U[c]:  mixer output buffer, 46 history + pending (batch + up to 63)
// Y2[c]: FIR2 output buffer, 70 history + pending (batch + one FIR2 run - 1)
// MIX:   9 + 9 doubles
#define XS 74
#define X_ENTRIES (4 * XS)
#ifndef NVX_Y2_RUN
#define NVX_Y2_RUN 160                    /* single-chain kernel: FIR2 outputs per FIR3 run, 160 (16 outputs) or 80 */
#endif

// Batching geometry.  One chain: FIR2 runs on 224 pending mixer outputs (32 outputs x {I,Q} = 64
// lanes), FIR3 on NVX_Y2_RUN pending FIR2 outputs.  Two chains: both chains share a run
// (lane = chain x output x component), so half the batch fills the wave and the pending
// buffers -- and with them the LDS footprint -- halve: 17.4 KB instead of 24 KB, 9 instead of 6
// waves per CU.  Either way a frame (20160 / 2880 outputs) is a whole number of runs.
template <int NCH> struct Geo;
template <> struct Geo<1> { static constexpr int U_RUN = 224, Y2_PER_RUN = 32, Y2_RUN = NVX_Y2_RUN, Y3_PER_RUN = NVX_Y2_RUN / 10; };
template <> struct Geo<2> { static constexpr int U_RUN = 112, Y2_PER_RUN = 16, Y2_RUN = 80, Y3_PER_RUN = 8; };
template <int NCH> struct GeoSizes {
    static constexpr int U_ENTRIES = ((46 + Geo<NCH>::U_RUN + 63) + 7) / 8 * 8;
    static constexpr int Y2_ENTRIES = ((70 + Geo<NCH>::Y2_RUN + Geo<NCH>::Y2_PER_RUN - 1) + 7) / 8 * 8;
};

template <int NCH>
struct CascadeLds {
    double2 X[X_ENTRIES];
    double2 U[NCH][GeoSizes<NCH>::U_ENTRIES];
    double2 Y2[NCH][GeoSizes<NCH>::Y2_ENTRIES];
    double  mix[2 * NVX_MIX_N];
};

// NB: __builtin_bit_cast applied directly to a vector-element expression (v.x)
// reads element 0 for every component with this compiler; go through a by-value
// scalar instead.
__device__ __forceinline__ nvx_short2 as_short2(unsigned w) { return __builtin_bit_cast(nvx_short2, w); }

__device__ __forceinline__ int dpp_swap_pairs(int v)
{
    // quad_perm [1,0,3,2]: every lane reads its lane^1 neighbour
    return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
}

// stage 0 for one 1-KiB load: lane l holds raw samples 4l..4l+3 of the KiB;
// lanes (2i, 2i+1) together hold the 8 samples of output i.  Even lanes produce
// the I sum, odd lanes the Q sum.  Each lane adds up both components of its own
// four samples -- two SDWA adds take the sign-extended low (I) or high (Q)
// halves of two words at once, a third add joins the pairs -- keeps its own
// component, hands the other one to its partner, and one DPP pair-swap add
// completes both sums.  11 VALU instructions per load, ~34 issue cycles
// (tools/valu_probe3.hip: SDWA and v_dot2c both issue in 4 cycles, plain VOP2
// in 2; the earlier form, eight v_dot2c with per-lane selector registers, took
// ~44).  -DNVX_STAGE0_DOT2C builds that earlier form for A/B runs.
#ifndef NVX_STAGE0_DOT2C
__device__ __forceinline__ int add_low_halves(unsigned a, unsigned b)
{
    int r;
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0"
        : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int add_high_halves(unsigned a, unsigned b)
{
    int r;
    asm("v_add_u32_sdwa %0, sext(%1), sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1"
        : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double stage0_component(u32x4 v, bool odd)
{
    const int sI = add_low_halves(v.x, v.y) + add_low_halves(v.z, v.w);
    const int sQ = add_high_halves(v.x, v.y) + add_high_halves(v.z, v.w);
    int mine = (odd ? sQ : sI) + 4;                           // + 4: round half up
    const int other = odd ? sI : sQ;
    asm("" : "+v"(mine));                                     // keeps the two VOP2 adds (2 cycles each, the second with the
    const int tot = mine + dpp_swap_pairs(other);             // DPP operand) from being merged into a v_mov_dpp + 4-cycle v_add3
    return (double)(tot >> 3);                // arithmetic shift = floor((sum+4)/8)
}
#else
__device__ __forceinline__ double stage0_component(u32x4 v, bool odd)
{
    const nvx_short2 selI = { 1, 0 }, selQ = { 0, 1 };          // (1,0) picks I, (0,1) picks Q
    const nvx_short2 sel_mine = odd ? selQ : selI, sel_other = odd ? selI : selQ;
    int mine = 2, other = 2;                  // 2 + 2 = the +4 of round-half-up
    mine  = __builtin_amdgcn_sdot2(as_short2(v.x), sel_mine,  mine,  false);
    other = __builtin_amdgcn_sdot2(as_short2(v.x), sel_other, other, false);
    mine  = __builtin_amdgcn_sdot2(as_short2(v.y), sel_mine,  mine,  false);
    other = __builtin_amdgcn_sdot2(as_short2(v.y), sel_other, other, false);
    mine  = __builtin_amdgcn_sdot2(as_short2(v.z), sel_mine,  mine,  false);
    other = __builtin_amdgcn_sdot2(as_short2(v.z), sel_other, other, false);
    mine  = __builtin_amdgcn_sdot2(as_short2(v.w), sel_mine,  mine,  false);
    other = __builtin_amdgcn_sdot2(as_short2(v.w), sel_other, other, false);
    const int tot = mine + dpp_swap_pairs(other);
    return (double)(tot >> 3);                // arithmetic shift = floor((sum+4)/8)
}
#endif

template <bool RAW, bool NT>
__device__ __forceinline__ void load_pass(u32x4 (&pf)[RAW ? 8 : 1], const u32x4 *src)
{
    // src already points at this lane's first 16 bytes of the pass
    if (RAW) {
#pragma unroll
        for (int j = 0; j < 8; j++) pf[j] = NT ? __builtin_nontemporal_load(src + 64 * j) : src[64 * j];
    } else {
        pf[0] = NT ? __builtin_nontemporal_load(src) : src[0];
    }
}

// Work distribution: a persistent grid (as many single-wave workgroups as fit
// on the chip) pulls units u = part * n_streams + stream from an atomic
// counter.  A unit is one frame (315 passes, 2.5 MiB raw) of one stream (or a
// third of one, NVX_UNIT_SPLIT); the FIR histories travel from unit (stream, p)
// to (stream, p+1) through the per-stream state block in HBM: the producer
// writes it with agent-scope atomic stores, drains them (vmcnt 0) and then sets
// done[stream]; the consumer polls done[stream] and reads the block with
// agent-scope atomic loads (the two units usually run on different XCDs).
// Units are handed out part-major, so a unit's predecessor was taken n_streams
// units earlier by a workgroup that is already running: the wait cannot
// deadlock whatever the residency, and every spin is bounded anyway.
// Why: LDS limits residency to 11 waves per CU (2816), so 4096 equal-length
// per-stream jobs would run as a VALU-saturated first round and a
// latency-bound tail of 1280; frame-sized units keep every CU full to the end.
#define NVX_SPIN_LIMIT (1 << 22)

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

template <bool RAW, int NCH, int PFD, bool NT>
__global__ __launch_bounds__(64) void nvx_fir_cascade(nvx_cascade_args a)
{
    __shared__ CascadeLds<NCH> lds;
    const int lane = threadIdx.x;

    if (lane < NVX_MIX_N) {
        // constant-index selects keep the tables out of scratch
        double cr = 0.0, ci = 0.0;
#pragma unroll
        for (int j = 0; j < NVX_MIX_N; j++) if (lane == j) { cr = NVX_MIX_CR[j]; ci = NVX_MIX_CI[j]; }
        lds.mix[lane] = cr; lds.mix[NVX_MIX_N + lane] = ci;
    }

    // ------------------------------------------------------ lane constants
    const size_t pass_words = RAW ? 2048 : 256;    // 32-bit IQ words per pass
    const size_t pass_stride = pass_words / 4;     // in 16-byte units
    constexpr int NPF = RAW ? 8 : 1;
    // stage-0 write slot of this lane (RAW): output m = 32j + (lane>>1):
    // phase r = m & 3, index k' = m >> 2 = 8j + (lane>>3), component = lane & 1
    double *xw = (double *)&lds.X[((lane >> 1) & 3) * XS + 9 + (lane >> 3)] + (lane & 1);
    const bool odd = lane & 1;
    // FIR1 read base of this lane: X[r*XS + 9 + lane - q]
    const double2 *xr = &lds.X[9 + lane];
    // FIR2 / FIR3: lane = 2*output + component
    const int half = lane >> 1, comp = lane & 1;
    const int lane_mod9 = lane % 9;
    static_assert(NVX_UNIT_SPLIT == 1 || NVX_UNIT_SPLIT == 3, "a unit must end with all pending buffers empty");
    const int n_units = a.n_streams * a.n_frames * NVX_UNIT_SPLIT;

    for (;;) {
        // ------------------------------------------------------ next unit
        int u = 0;
        if (lane == 0) u = __hip_atomic_fetch_add(a.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= n_units) break;
        const int part = u / a.n_streams;              // index of this unit in its stream: frame * NVX_UNIT_SPLIT + third
        const int stream = u - part * a.n_streams;

        const unsigned mask = a.chain_masks[stream];
        // NCH == 1: the single active chain; NCH == 2: chain slot c is chain c
        const int chain_of_slot0 = (NCH == 1) ? ((mask & 1u) ? 0 : 1) : 0;

        // independent units: rebuild the histories from the nine passes in front of the unit (nvx_kernels.h)
        const bool preroll = a.independent && part > 0;
        const int pre = preroll ? NVX_PREROLL_PASSES : 0;
        const int n_pass = pre + NVX_UNIT_PASSES;
        // the input does not depend on the predecessor: request the first pass(es) now
        const u32x4 *src = (const u32x4 *)(a.iq + ((size_t)stream * a.pitch + a.first_sample)) +
                           ((size_t)part * NVX_UNIT_PASSES - (size_t)pre) * pass_stride + lane;
        u32x4 pfA[NPF], pfB[NPF];
        load_pass<RAW, NT>(pfA, src);
        if (PFD == 2) load_pass<RAW, NT>(pfB, src + pass_stride);
        const u32x4 *nxt = src + PFD * pass_stride;    // first pass not yet requested

        // ------------------------------------------------------ wait for (stream, part-1)
        if (part > 0 && !a.independent) {
            int spins = 0, ok = 0;
            do {
                int d = 0;
                if (lane == 0) d = __hip_atomic_load(a.done + stream, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                d = __builtin_amdgcn_readfirstlane(d);
                ok = d >= part;
                if (!ok) __builtin_amdgcn_s_sleep(32);
            } while (!ok && ++spins < NVX_SPIN_LIMIT);
            if (spins > 0 && lane == 0) {                // instrumentation: how often, and how long, a hand-over was waited for
                __hip_atomic_fetch_add(a.status + 1, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(a.status + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!ok) {                                   // give up loudly rather than hang the GPU
                if (lane == 0) __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
#ifdef NVX_HANDOFF_FENCES
            // one agent-scope acquire per unit: the state lines may sit stale in this CU's L1
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            asm volatile("" ::: "memory");               // the state loads below stay below the flag poll
#endif
        }

        // ------------------------------------------------------ state in
        // A stream's first unit of a launch reads the block the previous launch left (state_in); every unit
        // writes state_out, which the host swaps with state_in between launches -- so a launch never reads
        // and writes the same block through different units (the independent units run in any order).
        double2 *st = (double2 *)(a.state_out + (size_t)stream * NVX_CASCADE_STATE_BYTES);
        const double2 *st_in = (part == 0) ? (const double2 *)(a.state_in + (size_t)stream * NVX_CASCADE_STATE_BYTES) : st;
        NVX_WAVE_LDS_FENCE();
        if (!preroll) {
            if (lane < 36) {                               // 36 newest 252 kS/s samples, oldest first
                int e = lane >> 2, r = lane & 3;           // sample -36+lane = 4*(e-9) + r
                lds.X[r * XS + e] = state_load(st_in + lane);
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int ch = (NCH == 1) ? chain_of_slot0 : c;
                const double2 *su = st_in + 36 + ch * (46 + 70);
                if (lane < 46) lds.U[c][lane] = state_load(su + lane);
                lds.Y2[c][lane] = state_load(su + 46 + lane);
                if (lane < 6) lds.Y2[c][64 + lane] = state_load(su + 46 + 64 + lane);
            }
        } else {
            const double2 zero = { 0.0, 0.0 };
            if (lane < 36) lds.X[(lane & 3) * XS + (lane >> 2)] = zero;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                for (int i = lane; i < 46 + NVX_PREROLL_U; i += 64) lds.U[c][i] = zero;
                for (int i = lane; i < 70 + NVX_PREROLL_Y2; i += 64) lds.Y2[c][i] = zero;
            }
        }
        NVX_WAVE_LDS_FENCE();

        // mixer index of the unit's first FIR1 output: 6720 * third mod 9 (0 at every frame start)
        // (the pre-roll starts 576 = 0 mod 9 outputs earlier: same index)
        int n_u = preroll ? NVX_PREROLL_U : 0, n_y2 = preroll ? NVX_PREROLL_Y2 : 0, n3_done = 0;
        int mixbase = ((part % NVX_UNIT_SPLIT) * (NVX_UNIT_PASSES * 64)) % NVX_MIX_N;
        bool emit = !preroll;                            // FIR3 outputs of the pre-roll are not written
        const size_t y3_row0 = (size_t)(stream * 2) * a.y3_cap + a.y3_base + (size_t)part * NVX_UNIT_Y3;

        auto body = [&](u32x4 (&pf)[NPF], const int pass) {
            // ---- 1. new 252 kS/s samples into the polyphase window ----------
            if (RAW) {
#pragma unroll
                for (int j = 0; j < 8; j++) xw[j * 16] = stage0_component(pf[j], odd);   // +8 double2 entries per load
            } else {
                // lane holds samples 4*lane .. 4*lane+3 = phases 0..3 of index k' = lane
                const uint32_t w[4] = { pf[0].x, pf[0].y, pf[0].z, pf[0].w };
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double2 v;
                    v.x = (double)(int)(short)(w[r] & 0xffffu);       // capt_sched.c:511 (double) of each short
                    v.y = (double)((int)w[r] >> 16);
                    lds.X[r * XS + 9 + lane] = v;
                }
            }
            // ---- 2. prefetch pass + PFD into the buffer just consumed --------------
            if (pass + PFD < n_pass) load_pass<RAW, NT>(pf, nxt);
            nxt += pass_stride;
            NVX_WAVE_LDS_FENCE();

            // ---- 3. FIR1: y1[k] = sum_i h1[i] * x[4k+3-i] ----------------------
            double aI = 0.0, aQ = 0.0;
#pragma unroll
            for (int i = 0; i < NVX_T1; i++) {
                const int q = i >> 2, r = 3 - (i & 3);
                double2 x = xr[r * XS - q];
                aI += NVX_H1[i] * x.x;
                aQ += NVX_H1[i] * x.y;
            }
            // ---- 4. mixer, table index (k mod 9), k counted from the frame start
            // (a frame is 20160 = 9 * 2240 FIR1 outputs, so that equals k from stream start)
            int j9 = mixbase + lane_mod9; if (j9 >= NVX_MIX_N) j9 -= NVX_MIX_N;
            const double cr = lds.mix[j9], ci = lds.mix[NVX_MIX_N + j9];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int ch = (NCH == 1) ? chain_of_slot0 : c;
                double2 uu;
                if (ch == 0) {                 // 518 chain, fir2cpp.C:116-117
                    uu.x = aI * cr - aQ * ci;
                    uu.y = aI * ci + aQ * cr;
                } else {                       // 490 chain, fir2cpp.C:122-123
                    uu.x = aI * cr + aQ * ci;
                    uu.y = -aI * ci + aQ * cr;
                }
                lds.U[c][46 + n_u + lane] = uu;
            }
            n_u += 64;
            mixbase += 1; if (mixbase == NVX_MIX_N) mixbase = 0;      // 64 mod 9 == 1
            // ---- 5. slide the 9-deep history of each phase to the front --------
            NVX_WAVE_LDS_FENCE();
            if (lane < 36) {
                int e = lane >> 2, r = lane & 3;
                double2 t = lds.X[r * XS + 64 + e];
                lds.X[r * XS + e] = t;
            }
            NVX_WAVE_LDS_FENCE();

            // ---- 6. FIR2 when a batch of mixer outputs is pending -----------------
            constexpr int U_RUN = Geo<NCH>::U_RUN, Y2_PER_RUN = Geo<NCH>::Y2_PER_RUN;
            constexpr int Y2_RUN = Geo<NCH>::Y2_RUN, Y3_PER_RUN = Geo<NCH>::Y3_PER_RUN;
            // lane -> (chain slot, output, component): one chain uses all 64 lanes for 32 outputs,
            // two chains put chain 0 on lanes 0-31 and chain 1 on lanes 32-63 (16 outputs each)
            const int f2c = (NCH == 2) ? (lane >> 5) : 0;
            const int f2o = (NCH == 2) ? ((lane >> 1) & 15) : half;
            while (n_u >= U_RUN) {
                {
                    const double *ub = (const double *)&lds.U[f2c][7 * f2o] + comp;
                    double acc = 0.0;
#pragma unroll
                    for (int i = 0; i < NVX_T2; i++) acc += NVX_H2[i] * ub[2 * (52 - i)];
                    if (NCH == 1 || ((mask >> f2c) & 1u)) ((double *)&lds.Y2[f2c][70 + n_y2 + f2o])[comp] = acc;
                }
                NVX_WAVE_LDS_FENCE();
                // drop the consumed inputs: keep 46 history + pending (<= 109 entries)
                const int keep = 46 + n_u - U_RUN;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    double2 t0 = lds.U[c][U_RUN + lane];
                    double2 t1 = lds.U[c][U_RUN + 64 + ((lane < 45) ? lane : 44)];
                    NVX_WAVE_LDS_FENCE();
                    if (lane < keep) lds.U[c][lane] = t0;
                    if (lane + 64 < keep) lds.U[c][64 + lane] = t1;
                }
                NVX_WAVE_LDS_FENCE();
                n_u -= U_RUN;
                n_y2 += Y2_PER_RUN;

                // ---- 7. FIR3 when a batch of FIR2 outputs is pending ----------------
                if (n_y2 >= Y2_RUN) {
                    // lanes 0 .. 2*Y3_PER_RUN-1 hold chain 0 (output, component); with two chains the
                    // next 2*Y3_PER_RUN lanes hold chain 1
                    const int f3c = (NCH == 2) ? ((lane >> 4) & 1) : 0;
                    const int f3o = half & (Y3_PER_RUN - 1);
                    const bool f3live = lane < 2 * Y3_PER_RUN * NCH;
                    {
                        const int ch = (NCH == 1) ? chain_of_slot0 : f3c;
                        const double *yb = (const double *)&lds.Y2[f3c][10 * f3o] + comp;
                        double acc = 0.0;
#pragma unroll
                        for (int i = 0; i < NVX_T3; i++) acc += NVX_H3[i] * yb[2 * (79 - i)];
                        if (emit && f3live && (NCH == 1 || ((mask >> f3c) & 1u))) {
                            double *out = (double *)(a.y3 + (y3_row0 + (size_t)ch * a.y3_cap + n3_done + f3o));
                            out[comp] = acc;
                        }
                    }
                    NVX_WAVE_LDS_FENCE();
                    const int keep3 = 70 + n_y2 - Y2_RUN;           // <= 101 (one chain) / 85 (two chains)
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        double2 t0 = lds.Y2[c][Y2_RUN + lane];
                        double2 t1 = lds.Y2[c][Y2_RUN + 64 + ((lane < 37) ? lane : 36) * (NCH == 1) + ((lane < 21) ? lane : 20) * (NCH == 2)];
                        NVX_WAVE_LDS_FENCE();
                        if (lane < keep3) lds.Y2[c][lane] = t0;
                        if (lane + 64 < keep3) lds.Y2[c][64 + lane] = t1;
                    }
                    NVX_WAVE_LDS_FENCE();
                    n_y2 -= Y2_RUN;
                    n3_done += Y3_PER_RUN;
                }
            }
        };

        if (PFD == 2) {
            for (int pass = 0; pass < n_pass; pass += 2) {
                if (pass == pre) { emit = true; n3_done = 0; }
                body(pfA, pass);
                if (pass + 1 == pre) { emit = true; n3_done = 0; }
                if (pass + 1 < n_pass) body(pfB, pass + 1);
            }
        } else {
            for (int pass = 0; pass < n_pass; pass++) {
                if (pass == pre) { emit = true; n3_done = 0; }
                body(pfA, pass);
            }
        }

        // ------------------------------------------------------ state out
        // (independent units: only the stream's last unit of the launch carries state into the next launch)
        NVX_WAVE_LDS_FENCE();
        if (!a.independent || part == a.n_frames * NVX_UNIT_SPLIT - 1) {
            if (lane < 36) {
                int e = lane >> 2, r = lane & 3;
                state_store(st + lane, lds.X[r * XS + e]);
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int ch = (NCH == 1) ? chain_of_slot0 : c;
                double2 *su = st + 36 + ch * (46 + 70);
                if (lane < 46) state_store(su + lane, lds.U[c][lane]);
                state_store(su + 46 + lane, lds.Y2[c][lane]);
                if (lane < 6) state_store(su + 46 + 64 + lane, lds.Y2[c][64 + lane]);
            }
        }
        // publish: the state stores (write-through, sc1) have completed at device level once vmcnt is 0; then the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef NVX_HANDOFF_FENCES
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        if (lane == 0 && !a.independent) __hip_atomic_store(a.done + stream, part + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

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

#ifndef NVX_FRONT_THREADS
#define NVX_FRONT_THREADS 256
#endif
#ifndef DTL
#define DTL 1152                         // time tile: 4 frames of 900 S/s samples (multiple of 9)
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
    const int slot = blockIdx.x, tid = threadIdx.x;
    if (!a.slot_active[slot]) return;                    // uniform over the block

    double *st = a.dstate + (size_t)slot * NVX_DEMOD_DOUBLES;
    const double2 *y3 = a.y3 + (size_t)slot * a.y3_cap + a.y3_base;
    double *dphi_out = a.dphi ? a.dphi + (size_t)slot * a.y3_cap + a.y3_base : nullptr;
    const double *hist = st + DS_Y3;                     // read in place: only the first 4 samples need it
    if (tid < 8) { s_dphi[tid] = st[DS_DPHI + tid]; s_S[tid] = st[DS_S + tid]; }
    for (int i = tid; i < 567; i += NVX_FRONT_THREADS) s_C[i] = st[DS_C + i];
    __syncthreads();

    for (int ta = 0; ta < a.n3; ta += DTL) {
        const int tl = min(DTL, a.n3 - ta);
        const unsigned long long gt = a.g0 + (unsigned long long)ta;     // g of L = 0
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
                double temp_max = -1.0;
                max_index = 0;
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const double v = s_S[L + i];
                    if (v > temp_max) { temp_max = v; max_index = i; }
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
    const int slot = blockIdx.x * 64 + threadIdx.x;
    const int nc = a.n_slots;
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

// ===========================================================================
// launchers (C linkage, called from the host runtime)
// ===========================================================================
// tuning switches for A/B runs (defaults are the shipped configuration)
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }

template <bool RAW, int NCH, int PFD, bool NT>
static hipError_t launch_cascade_as(const nvx_cascade_args *a, hipStream_t s)
{
    // persistent grid: as many single-wave workgroups as the chip holds at once
    static int n_cus = 0, fit_per_cu = 0;
    if (!n_cus) {
        int dev = 0, cus = 0, per_cu = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nvx_fir_cascade<RAW, NCH, PFD, NT>, 64, 0);
        if (e != hipSuccess) return e;
        const int cap = env_int("NVX_WAVES_PER_CU", 0);
        if (cap > 0 && cap < per_cu) per_cu = cap;
        fit_per_cu = per_cu > 0 ? per_cu : 1;
        n_cus = cus;
    }
    int per_cu = fit_per_cu;
    if (a->max_waves_per_cu > 0 && a->max_waves_per_cu < per_cu) per_cu = a->max_waves_per_cu;
    const int resident = n_cus * per_cu;
    const long long units = (long long)a->n_streams * a->n_frames * NVX_UNIT_SPLIT;
    const unsigned grid = (unsigned)(units < resident ? units : resident);
    // Fewer streams than resident waves: the units of one stream would run one after the other and most of the
    // chip would idle.  Then every unit rebuilds its filter histories from the nine passes in front of it
    // (+2.9 % input) and all of them run at once.  NVX_INDEPENDENT=0/1 forces the choice (tests, A/B runs).
    nvx_cascade_args args = *a;
    static const int force = env_int("NVX_INDEPENDENT", -1);
    args.independent = force >= 0 ? force : (a->n_streams < resident && a->n_frames * NVX_UNIT_SPLIT > 1);
    hipLaunchKernelGGL((nvx_fir_cascade<RAW, NCH, PFD, NT>), dim3(grid), dim3(64), 0, s, args);
    return hipGetLastError();
}

extern "C" hipError_t nvx_launch_cascade(const nvx_cascade_args *a, int raw, int nch, hipStream_t s)
{
    static const int pfd = env_int("NVX_PREFETCH", 1) == 2 ? 2 : 1;
    static const int nt = env_int("NVX_NT", 1) != 0;
    // queue counter, status word and per-stream completion counts start at zero every launch
    hipError_t e = hipMemsetAsync(a->queue, 0, (size_t)(NVX_CASCADE_CTRL_INTS + a->n_streams) * sizeof(int), s);
    if (e != hipSuccess) return e;
#define NVX_CASE(R, C) ( \
        pfd == 2 ? (nt ? launch_cascade_as<R, C, 2, true>(a, s) : launch_cascade_as<R, C, 2, false>(a, s)) \
                 : (nt ? launch_cascade_as<R, C, 1, true>(a, s) : launch_cascade_as<R, C, 1, false>(a, s)))
    if (raw) return nch == 1 ? NVX_CASE(true, 1) : NVX_CASE(true, 2);
    return nch == 1 ? NVX_CASE(false, 1) : NVX_CASE(false, 2);
#undef NVX_CASE
}

extern "C" hipError_t nvx_launch_channelise(const nvx_channelise_args *a, hipStream_t s)
{
    const size_t n_chunks = a->n_out / 64;
    const unsigned bx = (unsigned)((n_chunks + a->chunks_per_block - 1) / a->chunks_per_block);
    hipLaunchKernelGGL(nvx_channelise, dim3(bx, (unsigned)a->n_wide), dim3(64), 0, s, *a);
    return hipGetLastError();
}

extern "C" hipError_t nvx_launch_demod_front(const nvx_demod_args *a, hipStream_t s)
{
    hipLaunchKernelGGL(nvx_demod_front, dim3((unsigned)a->n_slots), dim3(NVX_FRONT_THREADS), 0, s, *a);
    return hipGetLastError();
}

extern "C" hipError_t nvx_launch_demod_fsm(const nvx_demod_args *a, hipStream_t s)
{
    hipLaunchKernelGGL(nvx_demod_fsm, dim3((unsigned)((a->n_slots + 63) / 64)), dim3(64), 0, s, *a);
    return hipGetLastError();
}

extern "C" hipError_t nvx_launch_synth(const nvx_synth_args *a, int n_streams, hipStream_t s)
{
    const size_t quads = (a->n + 3) / 4;
    dim3 grid((unsigned)((quads + 255) / 256), (unsigned)n_streams), block(256);
    hipLaunchKernelGGL(nvx_synth_kernel, grid, block, 0, s, *a);
    return hipGetLastError();
}

// host-callable copy of the device atan2, for tests (tests/test_atan2.py)
extern "C" __attribute__((visibility("default"))) double nvx_atan2_host(double y, double x) { return nvx_atan2(y, x); }
