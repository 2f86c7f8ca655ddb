// nvx_cascade.hip -- the roofline kernel of the NAVTEX receive path (gfx950).  The other kernels:
// nvx_demod.hip, nvx_channelise.hip, nvx_synth.hip.
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
//     just the three filter histories.#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "nvx_tables.h"
#include "nvx_kernels.h"
#include "nvx_device.h"

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
// launcher (C linkage, called from the host runtime)
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
