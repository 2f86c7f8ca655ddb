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
//   * FIR1 is split by COMPONENT: lane = (output pair k, component), each lane
//     accumulates outputs 2k and 2k+1 of its own component.  The two outputs
//     share 37 of their 41 input samples, so a lane reads 41 x 8 bytes from LDS
//     for two outputs instead of 2 x 37 x 16 for one complex output each: 45 % of
//     the LDS bytes per output (the 252 kS/s kernels were LDS-bandwidth bound:
//     profiles/r02/a0_*).  The window is polyphase-split by 8 ({I,Q} doubles) so
//     that the 64 lanes of every read touch 512 consecutive bytes: all
//     ds_read_b64 / ds_write_b64 / ds_write_b128 of FIR1 / stage 0 / the input
//     conversion are bank-conflict free;
//   * FIR2 and FIR3 run on batches of pending outputs sized to fill the wave
//     (struct Geo below); a frame of 32 bit periods = 315 passes is a whole
//     number of every batch, after which every decimation counter, the mixer
//     index and all LDS fill levels are back at zero: the carried state is
//     just the three filter histories.#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>

#include "nvx_tables.h"
#include "nvx_kernels.h"
#include "nvx_device.h"

// ------------------------------------------------------------------ LDS map
// X: eight polyphase arrays P_r[e] of 37 double2: entry e of phase r holds sample 8*(e - 5) + r of the pass
//    (5 history entries: a lane reaches back 33 samples; 32 new ones).  37 is odd and = 5 mod 8, which spreads
//    both write patterns (stage 0: ds_write_b64, lane pairs walk the phases; 252 kS/s input: ds_write_b128,
//    lane parity picks the phase quartet) over all banks.
// U[c]:  mixer output buffer, 46 history + pending (batch + up to 63)
// Y2[c]: FIR2 output buffer, 70 history + pending (batch + one FIR2 run - 1)
// MIX:   2 signs x 2 periods x 9 x (cos, -+sin): index mixbase + (o mod 9) + 1 <= 17 needs no wrap, and the lane's
//        sign of the cross product (step 4) is part of its table address
#define XPH 8
#define XH 5
#define XS 37
#define X_ENTRIES (XPH * XS)
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
    double2 mix[2][2 * NVX_MIX_N];      // [sign of the cross term][two periods of (cos, -+sin)]
};

__device__ __forceinline__ int dpp_swap_pairs(int v)
{
    // quad_perm [1,0,3,2]: every lane reads its lane^1 neighbour
    return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
}

// FIR1 window: s_j = x[8*half + 7 - j] is component comp of X[r * XS + XH + half + fl] with 7 - j = 8 * fl + r;
// offset in doubles from the lane's base pointer
#ifndef NVX_F1_GROUP
#define NVX_F1_GROUP 4                    /* LDS reads per wait */
#endif
#ifndef NVX_F1_AHEAD
#define NVX_F1_AHEAD 3                    /* groups in flight ahead of the arithmetic */
#endif
#define NVX_F23_AHEAD 12
// timing probes (wrong results on purpose; never shipped): 1 = FIR1 without its LDS reads, 2 = FIR1 without its fp64
// arithmetic, 3 = no FIR2 / FIR3
#ifndef NVX_PROBE
#define NVX_PROBE 0
#endif
// NVX_PROBE == 9: diagnostic build with s_memtime stamps at the phase boundaries of a pass (the stamps wait for the
// LDS queue, so the build runs slower; it shows where a wave's time goes, nothing else).  Prints per-wave totals.
#if NVX_PROBE == 9
#define NVX_STAMP(k) do { asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define NVX_STAMP(k) do { } while (0)
#endif
#if NVX_PROBE == 1 || NVX_PROBE == 7
#define NVX_PROBE_READ(expr, j) ((double)(lane + (j)))
#else
#define NVX_PROBE_READ(expr, j) (expr)
#endif                  /* FIR2 / FIR3: taps read ahead */
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

__device__ __forceinline__ double dpp_swap_pairs_f64(double v)
{
    return __hiloint2double(dpp_swap_pairs(__double2hiint(v)), dpp_swap_pairs(__double2loint(v)));
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

    if (lane < 4 * NVX_MIX_N) {
        // constant-index selects keep the tables out of scratch
        const int j9 = lane % NVX_MIX_N;
        double cr = 0.0, ci = 0.0;
#pragma unroll
        for (int j = 0; j < NVX_MIX_N; j++) if (j9 == j) { cr = NVX_MIX_CR[j]; ci = NVX_MIX_CI[j]; }
        lds.mix[0][lane % (2 * NVX_MIX_N)] = double2{ cr, ci };      // both copies are written by two lanes each: same value
        lds.mix[1][lane % (2 * NVX_MIX_N)] = double2{ cr, -ci };
    }

    // ------------------------------------------------------ lane constants
    const size_t pass_words = RAW ? 2048 : 256;    // 32-bit IQ words per pass
    const size_t pass_stride = pass_words / 4;     // in 16-byte units
    constexpr int NPF = RAW ? 8 : 1;
    // lane = 2 * (pair / output index) + component, in stage 0, FIR1, the mixer, FIR2 and FIR3 alike
    const int half = lane >> 1, comp = lane & 1;
    const bool odd = lane & 1;
    // stage-0 write slot of this lane (RAW): 252 kS/s sample m = 32j + half of the pass, own component:
    // phase r = m & 7 = half & 7, entry XH + (m >> 3) = XH + 4j + (lane >> 4)
    double *xw = (double *)&lds.X[(half & 7) * XS + XH + (lane >> 4)] + comp;
    // 252 kS/s input: the lane holds samples 4*lane .. 4*lane+3 = phases 4*(lane & 1) + s of entry XH + (lane >> 1)
    double2 *xw4 = &lds.X[(lane & 1) * 4 * XS + XH + half];
    // FIR1 read base of this lane: sample 8*half + t is component comp of X[(t & 7) * XS + XH + half + floor(t / 8)]
    const double *xr = (const double *)&lds.X[XH + half] + comp;
    const lds_vdouble *xrv = (const lds_vdouble *)xr;
    const int lane_mod9 = (2 * half) % 9;
    // mixer: the 518 chain's I lanes and the 490 chain's Q lanes subtract the cross product (see step 4)
    static_assert(NVX_UNIT_SPLIT == 1 || NVX_UNIT_SPLIT == 3, "a unit must end with all pending buffers empty");
    const int n_units = a.n_streams * a.n_frames * NVX_UNIT_SPLIT;
    double h1v[NVX_T1];
#pragma unroll
    for (int i = 0; i < NVX_T1; i++)
        if (NVX_H1_FIRST.first[i] == i) { h1v[i] = NVX_H1[i]; asm volatile("" : "+v"(h1v[i])); }

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

        // mixer table row of this lane: its cross term carries the sign of the 518 chain (I lanes negated) -- or, when the
        // unit's only chain is the 490 one, of that chain (Q lanes negated); see step 4
        const lds_vd2 *mixrow = (const lds_vd2 *)&lds.mix[(comp ^ (NCH == 1 ? chain_of_slot0 : 0)) ? 0 : 1][lane_mod9];

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
                const int v = lane + 4;                    // sample -36+lane = 8*((v>>3) - XH) + (v&7)
                lds.X[(v & 7) * XS + (v >> 3)] = state_load(st_in + lane);
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
            if (lane < XPH * XH) lds.X[(lane & 7) * XS + (lane >> 3)] = zero;
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

#if NVX_PROBE == 9
        unsigned long long ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t_prev = __builtin_amdgcn_s_memtime();
#endif
        auto body = [&](u32x4 (&pf)[NPF], const int pass) {
            NVX_STAMP(0);                                  // loop control between passes
#if NVX_PROBE == 9
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            NVX_STAMP(1);                                  // waiting for this pass's input
#endif
            // ---- 1. new 252 kS/s samples into the polyphase window ----------
            if (RAW) {
#pragma unroll
                for (int j = 0; j < 8; j++) xw[j * 8] = stage0_component(pf[j], odd);   // +4 double2 entries per load
            } else {
                const uint32_t w[4] = { pf[0].x, pf[0].y, pf[0].z, pf[0].w };
#if NVX_PROBE == 4
                if ((w[0] ^ w[1] ^ w[2] ^ w[3]) == 0x12345678u) xw4[0] = double2{ 1.0, 1.0 };
#else
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double2 v;
                    v.x = (double)(int)(short)(w[r] & 0xffffu);       // capt_sched.c:511 (double) of each short
                    v.y = (double)((int)w[r] >> 16);
                    xw4[r * XS] = v;
                }
#endif
            }
            // ---- 2. prefetch pass + PFD into the buffer just consumed --------------
            if (pass + PFD < n_pass) load_pass<RAW, NT>(pf, nxt);
            nxt += pass_stride;
            NVX_WAVE_LDS_FENCE();

            NVX_STAMP(2);                                  // input conversion + window write + next load issued
            // ---- 3. FIR1: y1[o] = sum_i h1[i] * x[4o+3-i], outputs o = 2*half and 2*half+1 of component comp.
            // With s_j = x[8*half + 7 - j]:  y1[2*half+1] = sum_j h1[j] * s_j (j = 0..36),
            //                               y1[2*half]   = sum_j h1[j-4] * s_j (j = 4..40): both in tap order.
            // The reads run NVX_F1_AHEAD samples ahead of the arithmetic, so the LDS latency is covered by the wave's own
            // fp64 work.  volatile: each read stays a ds_read_b64 (512 contiguous bytes per wave, 2 LDS cycles); merged
            // into ds_read2_b64 a pair would cost 8 (MI355X_MICROARCH.md, LDS table).
            // In front of them go the two reads whose results are only needed after FIR1 -- the mixer's table entries and
            // the tail of the new samples that becomes the next pass's history -- so that neither costs a round trip
            // through the LDS with the wave idle.
            const nvx_d2 c0 = mixrow[mixbase], c1 = mixrow[mixbase + 1];
            nvx_d2 tail = { 0.0, 0.0 };
            if (lane < XPH * XH) tail = *(const lds_vd2 *)&lds.X[(lane & 7) * XS + 32 + (lane >> 3)];
            double xs[NVX_T1 + 4];
            constexpr int F1N = NVX_T1 + 4, F1NG = (F1N + NVX_F1_GROUP - 1) / NVX_F1_GROUP;
#pragma unroll
            for (int j = 0; j < NVX_F1_AHEAD * NVX_F1_GROUP; j++) xs[j] = NVX_PROBE_READ(xrv[f1_offset(j)], j);
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int g = 0; g < F1NG; g++) {
                // reads of group g + AHEAD go out behind the arithmetic of group g - 1 ...
                if (g + NVX_F1_AHEAD < F1NG) {
                    NVX_PIN_AFTER(a1);
#pragma unroll
                    for (int j = (g + NVX_F1_AHEAD) * NVX_F1_GROUP; j < (g + NVX_F1_AHEAD + 1) * NVX_F1_GROUP && j < F1N; j++) xs[j] = NVX_PROBE_READ(xrv[f1_offset(j)], j);
                }
                // ... and one wait covers a whole group (LDS reads return in order): its first value "depends" on its last
                const int lo = g * NVX_F1_GROUP, hi = (lo + NVX_F1_GROUP < F1N ? lo + NVX_F1_GROUP : F1N) - 1;
                if (hi > lo) asm volatile("" : "+v"(xs[lo]) : "v"(xs[hi]));
#pragma unroll
                for (int j = lo; j <= hi; j++) {
#if NVX_PROBE == 2 || NVX_PROBE == 7
                    a1 = __hiloint2double(__double2hiint(a1) ^ __double2hiint(xs[j]), __double2loint(a1));
#else
                    if (j < NVX_T1) a1 += h1v[NVX_H1_FIRST.first[j]] * xs[j];
                    if (j >= 4) a0 += h1v[NVX_H1_FIRST.first[j - 4]] * xs[j];
#endif
                }
            }
            NVX_STAMP(3);                                  // FIR1
            // ---- 4. mixer, table index (o mod 9), o counted from the frame start
            // (a frame is 20160 = 9 * 2240 FIR1 outputs, so that equals o from stream start).
            // 518 chain (fir2cpp.C:116-117): (I*cr - Q*ci, I*ci + Q*cr); 490 chain (:122-123): (I*cr + Q*ci, -I*ci + Q*cr).
            // A lane owns one component of its two outputs and gets the other from its partner lane (DPP pair swap).
            // With A = mine*cr and B = other*ci, both chains' results are A + B or A - B (a sum commutes exactly and
            // (-I)*ci = -(I*ci) exactly): 518 -> I lane A - B, Q lane A + B; 490 -> I lane A + B, Q lane A - B.
#if NVX_PROBE == 5
            if (__double2hiint(a0) == 0x12345678 && __double2hiint(a1) == 0x12345678 && c0.x == 3.0 && c1.x == 3.0) lds.U[0][0] = double2{ 1.0, 1.0 };
#else
            {
                const double o0 = dpp_swap_pairs_f64(a0), o1 = dpp_swap_pairs_f64(a1);
                const double A0 = a0 * c0.x, A1 = a1 * c1.x;
                // the lane's table row holds +-ci: B carries the sign of the 518 chain (of the unit's only chain when
                // NCH == 1); with two chains the 490 one takes the opposite sign
                const double B0 = o0 * c0.y, B1 = o1 * c1.y;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const int ch = (NCH == 1) ? chain_of_slot0 : c;
                    double *uw = (double *)&lds.U[c][46 + n_u + 2 * half] + comp;
                    uw[0] = (NCH == 1 || ch == 0) ? A0 + B0 : A0 - B0;
                    uw[2] = (NCH == 1 || ch == 0) ? A1 + B1 : A1 - B1;
                }
            }
#endif
            n_u += 64;
            mixbase += 1; if (mixbase == NVX_MIX_N) mixbase = 0;      // 64 mod 9 == 1
            // ---- 5. slide the 5-deep history of each phase to the front --------
            NVX_WAVE_LDS_FENCE();
#if NVX_PROBE == 6
            if (lane < XPH * XH && tail.x == 1.2345) *(lds_vd2 *)&lds.X[(lane & 7) * XS + (lane >> 3)] = tail;
#else
            if (lane < XPH * XH) *(lds_vd2 *)&lds.X[(lane & 7) * XS + (lane >> 3)] = tail;
#endif
            NVX_WAVE_LDS_FENCE();

            NVX_STAMP(4);                                  // mixer + slide
            // ---- 6. FIR2 when a batch of mixer outputs is pending -----------------
            constexpr int U_RUN = Geo<NCH>::U_RUN, Y2_PER_RUN = Geo<NCH>::Y2_PER_RUN;
            constexpr int Y2_RUN = Geo<NCH>::Y2_RUN, Y3_PER_RUN = Geo<NCH>::Y3_PER_RUN;
            // lane -> (chain slot, output, component): one chain uses all 64 lanes for 32 outputs,
            // two chains put chain 0 on lanes 0-31 and chain 1 on lanes 32-63 (16 outputs each)
            const int f2c = (NCH == 2) ? (lane >> 5) : 0;
            const int f2o = (NCH == 2) ? ((lane >> 1) & 15) : half;
#if NVX_PROBE == 3
            if (n_u >= U_RUN) { n_u -= U_RUN; n_y2 += Y2_PER_RUN; if (n_y2 >= Y2_RUN) { n_y2 -= Y2_RUN; n3_done += Y3_PER_RUN; } }
#endif
            while (n_u >= U_RUN) {
                // the pending entries behind this run move to the front afterwards: read them now, write them after the FIR
                nvx_d2 ut0[NCH], ut1[NCH];
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    ut0[c] = *(const lds_vd2 *)&lds.U[c][U_RUN + lane];
                    ut1[c] = *(const lds_vd2 *)&lds.U[c][U_RUN + 64 + ((lane < 45) ? lane : 44)];
                }
                {
                    const lds_vdouble *ub = (const lds_vdouble *)((const double *)&lds.U[f2c][7 * f2o] + comp);
                    double xs[NVX_T2], acc = 0.0;                 // reads run ahead of the arithmetic, as in FIR1
#pragma unroll
                    for (int i = 0; i < NVX_F23_AHEAD; i++) xs[i] = ub[2 * (52 - i)];
#pragma unroll
                    for (int i = 0; i < NVX_T2; i++) {
                        if (i + NVX_F23_AHEAD < NVX_T2) { NVX_PIN_AFTER(acc); xs[i + NVX_F23_AHEAD] = ub[2 * (52 - (i + NVX_F23_AHEAD))]; }
                        acc += NVX_H2[i] * xs[i];
                    }
                    if (NCH == 1 || ((mask >> f2c) & 1u)) ((double *)&lds.Y2[f2c][70 + n_y2 + f2o])[comp] = acc;
                }
                NVX_WAVE_LDS_FENCE();
                // drop the consumed inputs: keep 46 history + pending (<= 109 entries)
                const int keep = 46 + n_u - U_RUN;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    if (lane < keep) *(lds_vd2 *)&lds.U[c][lane] = ut0[c];
                    if (lane + 64 < keep) *(lds_vd2 *)&lds.U[c][64 + lane] = ut1[c];
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
                    nvx_d2 yt0[NCH], yt1[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        yt0[c] = *(const lds_vd2 *)&lds.Y2[c][Y2_RUN + lane];
                        yt1[c] = *(const lds_vd2 *)&lds.Y2[c][Y2_RUN + 64 + ((lane < 37) ? lane : 36) * (NCH == 1) + ((lane < 21) ? lane : 20) * (NCH == 2)];
                    }
                    {
                        const int ch = (NCH == 1) ? chain_of_slot0 : f3c;
                        const lds_vdouble *yb = (const lds_vdouble *)((const double *)&lds.Y2[f3c][10 * f3o] + comp);
                        double xs[NVX_T3], acc = 0.0;
#pragma unroll
                        for (int i = 0; i < NVX_F23_AHEAD; i++) xs[i] = yb[2 * (79 - i)];
#pragma unroll
                        for (int i = 0; i < NVX_T3; i++) {
                            if (i + NVX_F23_AHEAD < NVX_T3) { NVX_PIN_AFTER(acc); xs[i + NVX_F23_AHEAD] = yb[2 * (79 - (i + NVX_F23_AHEAD))]; }
                            acc += NVX_H3[i] * xs[i];
                        }
                        if (emit && f3live && (NCH == 1 || ((mask >> f3c) & 1u))) {
                            double *out = (double *)(a.y3 + (y3_row0 + (size_t)ch * a.y3_cap + n3_done + f3o));
                            out[comp] = acc;
                        }
                    }
                    NVX_WAVE_LDS_FENCE();
                    const int keep3 = 70 + n_y2 - Y2_RUN;           // <= 101 (one chain) / 85 (two chains)
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        if (lane < keep3) *(lds_vd2 *)&lds.Y2[c][lane] = yt0[c];
                        if (lane + 64 < keep3) *(lds_vd2 *)&lds.Y2[c][64 + lane] = yt1[c];
                    }
                    NVX_WAVE_LDS_FENCE();
                    n_y2 -= Y2_RUN;
                    n3_done += Y3_PER_RUN;
                }
            }
            NVX_STAMP(5);                                  // FIR2 / FIR3
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

#if NVX_PROBE == 9
        if (lane == 0 && blockIdx.x < 24 && u < 4096)
            printf("PH wg %d unit %d: loop %llu inwait %llu input %llu fir1 %llu mix %llu fir23 %llu\n", (int)blockIdx.x, u, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5]);
#endif
        // ------------------------------------------------------ state out
        // (independent units: only the stream's last unit of the launch carries state into the next launch)
        NVX_WAVE_LDS_FENCE();
        if (!a.independent || part == a.n_frames * NVX_UNIT_SPLIT - 1) {
            if (lane < 36) {
                const int v = lane + 4;
                state_store(st + lane, lds.X[(v & 7) * XS + (v >> 3)]);
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
#define NVX_MAX_DEVICES 64
// tuning switches for A/B runs (defaults are the shipped configuration; read once per process)
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }

template <bool RAW, int NCH, int PFD, bool NT>
static hipError_t launch_cascade_as(const nvx_cascade_args *a, hipStream_t s)
{
    // persistent grid: as many single-wave workgroups as the device of this launch holds at once (cached per device:
    // handles on different devices, and launches from different threads, share this function)
    static std::mutex mu;
    static int cus_of[NVX_MAX_DEVICES], fit_of[NVX_MAX_DEVICES];
    int n_cus = 0, fit_per_cu = 0;
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(mu);
        const bool cached = dev >= 0 && dev < NVX_MAX_DEVICES && cus_of[dev] > 0;
        if (!cached) {
            int cus = 0, per_cu = 0;
            e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nvx_fir_cascade<RAW, NCH, PFD, NT>, 64, 0);
            if (e != hipSuccess) return e;
            const int cap = env_int("NVX_WAVES_PER_CU", 0);
            if (cap > 0 && cap < per_cu) per_cu = cap;
            n_cus = cus; fit_per_cu = per_cu > 0 ? per_cu : 1;
            if (dev >= 0 && dev < NVX_MAX_DEVICES) { cus_of[dev] = n_cus; fit_of[dev] = fit_per_cu; }
        } else {
            n_cus = cus_of[dev]; fit_per_cu = fit_of[dev];
        }
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
