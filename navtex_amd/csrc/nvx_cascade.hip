// nvx_cascade.hip -- the roofline kernel of the NAVTEX receive path (gfx950).  The other kernels:
// nvx_demod.hip, nvx_channelise.hip, nvx_wideband_fused.hip, nvx_synth.hip.
//
//   nvx_fir_cascade<RAW, NCH>            int16 IQ in HBM -> 900 S/s complex fp64 per chain
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
//     the LDS bytes per output (in the 252 kS/s kernels, which have no HBM bound to
//     hide behind, the LDS array was 87 % busy: profiles/r02/a0_*).  The window is
//     polyphase-split by 8 ({I,Q} doubles) so
//     that the 64 lanes of every read touch 512 consecutive bytes: all
//     ds_read_b64 / ds_write_b64 / ds_write_b128 of FIR1 / stage 0 / the input
//     conversion are bank-conflict free;
//   * FIR2 and FIR3 run on batches of pending outputs sized to fill the wave
//     (struct Geo, nvx_cascade_wave.h); a frame of 32 bit periods = 315 passes is a whole
//     number of every batch, after which every decimation counter, the mixer
//     index and all LDS fill levels are back at zero: the carried state is
//     just the three filter histories.
// The per-wave cascade itself (LDS map, FIR1 .. FIR3, carried state) is nvx_cascade_wave.h, shared with the fused
// wideband kernel (nvx_wideband_fused.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>

#include "nvx_cascade_wave.h"

// stage 0 for one 1-KiB load: lane l holds raw samples 4l..4l+3 of the KiB;
// lanes (2i, 2i+1) together hold the 8 samples of output i.  Even lanes produce
// the I sum, odd lanes the Q sum.  Each lane adds up both components of its own
// four samples -- two SDWA adds take the sign-extended low (I) or high (Q)
// halves of two words at once, a third add joins the pairs -- keeps its own
// component, hands the other one to its partner, and one DPP pair-swap add
// completes both sums.  11 VALU instructions per load, ~34 issue cycles
// (tools/valu_probe3.hip: SDWA and v_dot2c both issue in 4 cycles, plain VOP2
// in 2; the first form, eight v_dot2c with per-lane selector registers, took ~44: profiles/TUNING.md).
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

// Stage 0, third-order form (nvx_config.stage0_order = 3): three cascaded 8-sample boxcars decimated by 8, i.e. the
// 22-tap filter w = 1 3 6 10 15 21 28 36 42 46 48 48 46 ... 3 1 (sum 512):  y[k] = (sum_j w[j] x[8k+7-j] + 256) >> 9.
// 76 dB of alias rejection at the NAVTEX offsets where the integrate-and-dump has 25.
// Per block b of 8 samples three weighted sums:  A_b = sum_i w[7-i] x_i,  B_b = sum_i w[15-i] x_i,  C_b = sum_i w[23-i] x_i
// (w[22] = w[23] = 0), and y[k] = A_k + B_(k-1) + C_(k-2).  Lanes as in the first-order form: the pair (2i, 2i+1) holds
// block i of the load, four samples each; a lane sums both components of its four samples (v_perm de-interleaves them
// into (x_i, x_i+1) half-word pairs -- its own component and the partner's, the selector is per lane -- and
// v_dot2_i32_i16 multiplies a pair by a weight pair): the partner's component first, which crosses to its owner by a DPP
// pair swap, and the owner's own two dot products accumulate on top of what arrived.  The two block delays are two
// rotations of the wave by one lane pair (wave_ror:1 twice); lanes 62 / 63 of the rotated value come from the previous
// load, so that its last block lands in front of this load's first:
//      t_b = B_b + C_(b-1),  y_b = A_b + t_(b-1).
// 29 VALU instructions per load as compiled (first-order form: 11; the round-2 form of step() below compiled to 34:
// profiles/r06/c0_*).  Carried between units: lanes 62 / 63 of (C, t) of the last load, four integers, entry
// NVX_CASCADE_STATE_ENTRIES - 1 of the state block.
struct Stage0Cic3 {
    unsigned sel_mine, sel_other;            // v_perm selectors: low halves (I) or high halves (Q) of two words
    unsigned wa0, wa1, wb0, wb1, wc0, wc1;   // weight pairs of this lane's samples (0..3 on even lanes, 4..7 on odd ones)
    int c_prev, t_prev;                      // C and t of the previous load
    bool last_pair;                          // lanes 62, 63

    __device__ __forceinline__ void init(int lane)
    {
        const bool odd = lane & 1;
        sel_mine = odd ? 0x07060302u : 0x05040100u;
        sel_other = odd ? 0x05040100u : 0x07060302u;
        // w[7-i], w[15-i], w[23-i] for i = 0..3 (even lanes) or 4..7 (odd lanes), packed (first | second << 16)
        wa0 = odd ? (10u | 6u << 16) : (36u | 28u << 16);  wa1 = odd ? (3u | 1u << 16) : (21u | 15u << 16);
        wb0 = odd ? (48u | 48u << 16) : (28u | 36u << 16); wb1 = odd ? (46u | 42u << 16) : (42u | 46u << 16);
        wc0 = odd ? (6u | 10u << 16) : 0u;                 wc1 = odd ? (15u | 21u << 16) : (1u | 3u << 16);
        last_pair = lane >= 62;
        c_prev = 0; t_prev = 0;
    }
    __device__ __forceinline__ static int rot1(int v) { return __builtin_amdgcn_mov_dpp(v, 0x13C, 0xF, 0xF, false); }   // wave_ror:1: lane l reads lane l-1, lane 0 lane 63
    // a dot product that STARTS a sum: the VOP3P form, whose third operand is the inline constant 0.  (The builtin with a
    // zero accumulator compiles to v_mov_b32 acc, 0 + v_dot2c_i32_i16: six moves per load -- r6, counted in the ISA.)
    __device__ __forceinline__ static int dot2_first(nvx_short2 a, unsigned w)
    {
        int r;
        asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(w));
        return r;
    }
    // r6: 29 vector instructions per load where the compiler had made 34 of the round-2 form (own sums and partner's sums
    // side by side, three adds to join them: PMC 62.9 vector instructions per load in the kernel where the design counted
    // 55.5; profiles/r06/c0_*).  Same integers, same sums mod 2^32, other grouping: the partner's three weighted sums first
    // (the rounding constant rides on the first of them); each crosses to its owner by ONE DPP move, and the owner's own two
    // dot products accumulate ON TOP of what arrived -- no separate add, no accumulator to clear.  Kernel -1.2 %.
    __device__ __forceinline__ double step(u32x4 v)
    {
        const nvx_short2 m01 = as_short2(__builtin_amdgcn_perm(v.y, v.x, sel_mine)), m23 = as_short2(__builtin_amdgcn_perm(v.w, v.z, sel_mine));
        const nvx_short2 o01 = as_short2(__builtin_amdgcn_perm(v.y, v.x, sel_other)), o23 = as_short2(__builtin_amdgcn_perm(v.w, v.z, sel_other));
        int oa = __builtin_amdgcn_sdot2(o01, as_short2(wa0), 256, false); oa = __builtin_amdgcn_sdot2(o23, as_short2(wa1), oa, false);   // + 256: round half up
        int ob = dot2_first(o01, wb0); ob = __builtin_amdgcn_sdot2(o23, as_short2(wb1), ob, false);
        int oc = dot2_first(o01, wc0); oc = __builtin_amdgcn_sdot2(o23, as_short2(wc1), oc, false);
        int A = dpp_swap_pairs(oa); A = __builtin_amdgcn_sdot2(m01, as_short2(wa0), A, false); A = __builtin_amdgcn_sdot2(m23, as_short2(wa1), A, false);
        int B = dpp_swap_pairs(ob); B = __builtin_amdgcn_sdot2(m01, as_short2(wb0), B, false); B = __builtin_amdgcn_sdot2(m23, as_short2(wb1), B, false);
        int C = dpp_swap_pairs(oc); C = __builtin_amdgcn_sdot2(m01, as_short2(wc0), C, false); C = __builtin_amdgcn_sdot2(m23, as_short2(wc1), C, false);
        const int t = B + rot1(rot1(last_pair ? c_prev : C));
        const int y = A + rot1(rot1(last_pair ? t_prev : t));
        c_prev = C; t_prev = t;
        return (double)(y >> 9);             // arithmetic shift = floor
    }
};

// One pass of input into registers: every byte is read once and never again, so the loads are non-temporal (plain
// loads measured 1.4 % slower).  src already points at this lane's first 16 bytes of the pass.
template <bool RAW>
__device__ __forceinline__ void load_pass(u32x4 (&pf)[RAW ? 8 : 1], const u32x4 *src)
{
    if (RAW) {
#pragma unroll
        for (int j = 0; j < 8; j++) pf[j] = __builtin_nontemporal_load(src + 64 * j);
    } else {
        pf[0] = __builtin_nontemporal_load(src);
    }
}

// Work distribution: a persistent grid (as many single-wave workgroups as fit
// on the chip) pulls units u = part * n_streams + stream from an atomic
// counter.  A unit is one frame (315 passes, 2.5 MiB raw) of one stream (or a
// third of one: the last frame of a launch, see the kernel); the FIR histories travel from unit (stream, p)
// to (stream, p+1) through the per-stream state block in HBM: the producer
// writes it with agent-scope atomic stores, drains them (vmcnt 0) and then sets
// done[stream]; the consumer polls done[stream] and reads the block with
// agent-scope atomic loads (the two units usually run on different XCDs).
// Units are handed out part-major, so a unit's predecessor was taken n_streams
// units earlier by a workgroup that is already running: the wait cannot
// deadlock whatever the residency, and every spin is bounded anyway.
// r2: a unit that finds its predecessor still running does not wait for it: it
// rebuilds the histories from the nine passes in front of it (the pre-roll of
// the independent-unit form, +2.9 % of a unit, identical results) and only
// PUBLISHES behind its predecessor, so the state block and done[] still advance
// in order.  1.4 % of the units at the headline size, 10 % with 252 kS/s input;
// cascade 20.86 -> 20.30 ms and 74.4-78.4 -> 73.7 ms (profiles/TUNING.md).
// NVX_DYNAMIC_PREROLL=0: such a unit waits, as in round 1.
// Why: LDS limits residency to 11 waves per CU (2816), so 4096 equal-length
// per-stream jobs would run as a VALU-saturated first round and a
// latency-bound tail of 1280; frame-sized units keep every CU full to the end.
#define NVX_SPIN_LIMIT (1 << 22)

// ARGS: how the wave sees the kernel's argument block.  By reference the compiler fetches a field from the kernarg
// segment where it is used (s_load, per unit) and the pass loop has that many more scalar registers; by value every field
// sits in a scalar register from the first instruction on.  Same results, different register allocation -- measured on
// one box, interleaved: raw-rate kernel 20.28 / 20.39 ms by reference against 20.76 / 20.86 by value; the 252 kS/s kernel
// showed nothing beyond its run-to-run spread, so its code stays as it was (profiles/TUNING.md).
// LIST: the launch names its streams (nvx_kernels.h, nvx_part: a push-mode handle whose streams have come apart in
// time).  A kernel of its own, so that the launches of every stream -- the roofline configuration -- run exactly the
// code they ran before lists existed (with the list test inside it the headline kernel was 0.8 % slower: 20.50 against
// 20.34 ms, same box, interleaved, profiles/r03).
template <bool RAW, int NCH, int S0, typename ARGS, bool LIST = false>
__device__ __forceinline__ void cascade_wave_main(ARGS a)
{
    __shared__ CascadeLds<NCH> lds;
    const int lane = threadIdx.x;
    CascadeWave<NCH> cw;
    cw.init(&lds, lane);

    // ------------------------------------------------------ lane constants
    const size_t pass_words = RAW ? 2048 : 256;    // 32-bit IQ words per pass
    const size_t pass_stride = pass_words / 4;     // in 16-byte units
    constexpr int NPF = RAW ? 8 : 1;
    const int half = lane >> 1, comp = lane & 1;
    const bool odd = lane & 1;
    // stage-0 write slot of this lane (RAW): 252 kS/s sample m = 32j + half of the pass, own component:
    // phase r = m & 7 = half & 7, entry XH + (m >> 3) = XH + 4j + (lane >> 4)
    double *xw = (double *)&lds.X[(half & 7) * XS + XH + (lane >> 4)] + comp;
    // 252 kS/s input: the lane holds samples 4*lane .. 4*lane+3 = phases 4*(lane & 1) + s of entry XH + (lane >> 1)
    double2 *xw4 = &lds.X[(lane & 1) * 4 * XS + XH + half];
    static_assert(S0 == 1 || (S0 == 3 && RAW), "the third-order stage 0 belongs to the raw-rate kernels");
    Stage0Cic3 s0;
    if (S0 == 3) s0.init(lane);
    // Units: whole frames, except that the frames from a.split_from on are handed out in thirds (105 passes; every
    // pending buffer is empty there too, nvx_kernels.h) -- the launcher asks for that in the launch's LAST frame, so
    // that the ragged end of the persistent grid is a third of a frame long instead of a whole one.  Positions in a
    // stream count in thirds (tpart); done[stream] = thirds completed.
    const int n_full = a.n_streams * a.split_from;
    const int n_units = n_full + 3 * a.n_streams * (a.n_frames - a.split_from);

    for (;;) {
        // ------------------------------------------------------ next unit
        int u = 0;
        if (lane == 0) u = __hip_atomic_fetch_add(a.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= n_units) break;
        int tpart, thirds, entry;                      // first third of the unit in its stream, thirds it covers (3 = a frame)
        if (u < n_full) { const int frame = u / a.n_streams; entry = u - frame * a.n_streams; tpart = 3 * frame; thirds = 3; }
        else { const int v = u - n_full, q = v / a.n_streams; entry = v - q * a.n_streams; tpart = 3 * a.split_from + q; thirds = 1; }
        const int part = tpart;                        // (position in the stream, in thirds)
        // which stream: every stream in index order, all reading state[0] and writing state[1] -- or (LIST) the launch's
        // list of participants (nvx_kernels.h, nvx_part), each with its own parity
        int stream = entry;
        const uint8_t *state_rd = a.state[0];
        uint8_t *state_wr = a.state[1];
        unsigned third0 = a.third0;                    // where the launch starts in the stream, in thirds of a frame since reset
        if (LIST) {
            const unsigned long long e = nvx_load_const_u64(a.part + entry);      // { stream, parity }
            stream = (int)(unsigned)e;
            if (e >> 32) { state_rd = a.state[1]; state_wr = a.state[0]; }
            third0 = (unsigned)(nvx_load_const_u64(&a.part[entry].g0) / NVX_THIRD_Y3);
        }
        int *const done = a.done + entry;              // hand-over flag of this stream within the launch
        const unsigned mask = a.chain_masks[stream];

        // independent units: rebuild the histories from the nine passes in front of the unit (nvx_kernels.h)
        bool preroll = a.independent && part > 0;
        // ... and so does a unit of a hand-over launch whose predecessor is still running (a.dynamic_preroll): nine passes
        // more (2.9 % of a unit) instead of the rest of the predecessor's run time.  Same results either way
        // (tests: the two unit forms agree bit for bit).  Such a unit owes its successor the order of the state block: it
        // publishes only behind its predecessor (the wait moves from the start of the unit, where it costs, to its end).
        bool late_publish = false;
        // the input does not depend on the predecessor: request the first pass now (of the unit itself; a unit that turns
        // out to need the pre-roll requests its real first pass again and lets this one go)
        const u32x4 *unit0 = (const u32x4 *)(a.iq + ((size_t)stream * a.pitch + a.first_sample)) + (size_t)part * NVX_THIRD_PASSES * pass_stride + lane;
        u32x4 pfA[NPF];
        if (!preroll) load_pass<RAW>(pfA, unit0);

        // ------------------------------------------------------ wait for (stream, part-1)
        if (part > 0 && !a.independent) {
            int spins = 0, ok = 0;
            do {
                int d = 0;
                if (lane == 0) d = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                d = __builtin_amdgcn_readfirstlane(d);
                ok = d >= part;
                if (!ok && a.dynamic_preroll) break;
                if (!ok) __builtin_amdgcn_s_sleep(32);
            } while (!ok && ++spins < NVX_SPIN_LIMIT);
            if (!ok && a.dynamic_preroll) {
                preroll = true; late_publish = true;
                if (lane == 0) __hip_atomic_fetch_add(a.status + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // units that found their predecessor running
            } else {
                if (spins > 0 && lane == 0) {                // instrumentation: how often, and how long, a hand-over was waited for
                    __hip_atomic_fetch_add(a.status + 1, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(a.status + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!ok) {                                   // give up loudly rather than hang the GPU
                    if (lane == 0) __hip_atomic_store(a.status, NVX_STATUS_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                asm volatile("" ::: "memory");               // the state loads below stay below the flag poll
            }
        }

        // ------------------------------------------------------ state in
        // A stream's first unit of a launch reads the block the stream's previous launch left (state[parity]); every
        // unit writes the other one, and the host flips the stream's parity behind every launch it took part in -- so
        // a launch never reads and writes the same block through different units (the independent units run in any order).
        double2 *st = (double2 *)(state_wr + (size_t)stream * NVX_CASCADE_STATE_BYTES);
        const double2 *st_in = (part == 0) ? (const double2 *)(state_rd + (size_t)stream * NVX_CASCADE_STATE_BYTES) : st;
#ifdef NVX_INJECT_STALE
        // fault-injection build (tests): every NVX_INJECT_STALE-th unit that takes over from a predecessor reads the OTHER
        // block instead -- what the stream's previous launch left there, a well-formed block of an earlier position: a
        // stale hand-over in its purest form.  The seal must catch every one of them and the results must not change.
        if (part > 0 && !preroll && (u % NVX_INJECT_STALE) == 0) st_in = (const double2 *)(state_rd + (size_t)stream * NVX_CASCADE_STATE_BYTES);
#endif
        cw.set_mask(mask);
        unsigned long long ct = 0;                     // S0 == 3: (C, t) of the predecessor's last lane pair, or silence in front of a pre-roll
        if (!preroll) {
            NVX_WAVE_LDS_FENCE();
            const unsigned long long sealed = seal_load(st_in);
            unsigned long long fold = cw.state_in(st_in);
            if (S0 == 3 && s0.last_pair) {
                ct = __hip_atomic_load((const unsigned long long *)(st_in + NVX_STATE_CIC3) + (lane & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fold ^= seal_rotc<37>(ct);
            }
            // The seal (nvx_kernels.h): what was loaded must be what the predecessor stored, whole and of the right position.
            // (a stream at position 0 has the zeros of nvx_reset in front of it: nothing was ever stored there)
            const unsigned third_in = third0 + (unsigned)part;
            if (third_in != 0 && !seal_ok(sealed, wave_fold64(fold), stream, third_in)) {
                if (part == 0) {
                    // inherited through a kernel boundary, not through the fence-free hand-over, and nothing to fall back
                    // on (the samples in front of the launch are gone): the launch is reported as failed (the unit runs
                    // on, so that its successors are not left waiting; the host discards the launch's results)
                    if (lane == 0) __hip_atomic_store(a.status, NVX_STATUS_INTEGRITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    // a stale or torn hand-over: the unit rebuilds its histories from its own input instead (the
                    // independent units' pre-roll: bit-identical by construction) and the event is counted
                    preroll = true; ct = 0;
                    if (lane == 0) __hip_atomic_fetch_add(a.status + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        const int pre = preroll ? NVX_PREROLL_PASSES : 0;
        const int n_pass = pre + thirds * NVX_THIRD_PASSES;
        const u32x4 *src = unit0 - (size_t)pre * pass_stride;
        if (preroll) load_pass<RAW>(pfA, src);
        const u32x4 *nxt = src + pass_stride;          // first pass not yet requested (one pass of prefetch; two measured null)
        // mixer index of the unit's first FIR1 output: 6720 * third mod 9 (0 at every frame start; the pre-roll starts
        // 576 = 0 mod 9 outputs earlier: same index); FIR3 outputs of the pre-roll are not written
        cw.begin_unit(mask, a.y3, (size_t)(stream * 2) * a.y3_cap + a.y3_base + (size_t)part * NVX_THIRD_Y3, a.y3_cap,
                      ((part % 3) * (NVX_THIRD_PASSES * 64)) % NVX_MIX_N,
                      preroll ? NVX_PREROLL_U : 0, preroll ? NVX_PREROLL_Y2 : 0, !preroll);
        NVX_WAVE_LDS_FENCE();
        if (preroll) cw.state_zero();
        NVX_WAVE_LDS_FENCE();
        if (S0 == 3) { s0.c_prev = (int)(unsigned)ct; s0.t_prev = (int)(unsigned)(ct >> 32); }

        auto body = [&](u32x4 (&pf)[NPF], const int pass) {
            // ---- 1. new 252 kS/s samples into the polyphase window ----------
            if (RAW) {
#pragma unroll
                for (int j = 0; j < 8; j++) xw[j * 8] = (S0 == 3) ? s0.step(pf[j]) : stage0_component(pf[j], odd);   // +4 double2 entries per load
            } else {
                const uint32_t w[4] = { pf[0].x, pf[0].y, pf[0].z, pf[0].w };
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double2 v;
                    v.x = (double)(int)(short)(w[r] & 0xffffu);       // capt_sched.c:511 (double) of each short
                    v.y = (double)((int)w[r] >> 16);
                    xw4[r * XS] = v;
                }
            }
            // ---- 2. prefetch the next pass into the buffer just consumed --------------
            if (pass + 1 < n_pass) load_pass<RAW>(pf, nxt);
            nxt += pass_stride;
            NVX_WAVE_LDS_FENCE();
            // ---- 3.-7. FIR1, mixer, history slide, FIR2 / FIR3 when their batches are full
            cw.compute_pass();
        };

        for (int pass = 0; pass < n_pass; pass++) {
            if (pass == pre) { cw.emit = true; cw.n3_done = 0; }
            body(pfA, pass);
        }

        // ------------------------------------------------------ state out
        // (independent units: only the stream's last unit of the launch carries state into the next launch)
        NVX_WAVE_LDS_FENCE();
        if (late_publish) {
            // the predecessor has to have published (state block, then done[]) before this unit's state goes on top of it
            int spins = 0, ok = 0;
            do {
                int d = 0;
                if (lane == 0) d = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                d = __builtin_amdgcn_readfirstlane(d);
                ok = d >= part;
                if (!ok) __builtin_amdgcn_s_sleep(32);
            } while (!ok && ++spins < NVX_SPIN_LIMIT);
            if (spins > 0 && lane == 0) __hip_atomic_fetch_add(a.status + 1, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!ok) {
                if (lane == 0) __hip_atomic_store(a.status, NVX_STATUS_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            asm volatile("" ::: "memory");
        }
        if (!a.independent || part + thirds == 3 * a.n_frames) {
            unsigned long long fold = cw.state_out(st);
            if (S0 == 3 && s0.last_pair) {
                const unsigned long long ct_out = (unsigned long long)(unsigned)s0.c_prev | ((unsigned long long)(unsigned)s0.t_prev << 32);
                __hip_atomic_store((unsigned long long *)(st + NVX_STATE_CIC3) + (lane & 1), ct_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fold ^= seal_rotc<37>(ct_out);
            }
            fold = wave_fold64(fold);
            if (lane == 0) seal_store(st, fold, stream, third0 + (unsigned)(part + thirds));
        }
        // publish: the state stores (write-through, sc1) have completed at device level once vmcnt is 0; then the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && !a.independent) __hip_atomic_store(done, part + thirds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool RAW, int NCH>
__global__ __launch_bounds__(64) void nvx_fir_cascade(nvx_cascade_args a)
{
    if (RAW) cascade_wave_main<RAW, NCH, 1, const nvx_cascade_args &>(a);
    else cascade_wave_main<RAW, NCH, 1, const nvx_cascade_args>(a);
}

// The raw-rate kernels with the third-order stage 0.  The single-chain one is held to 168 VGPRs (it would take 171: two
// waves per SIMD instead of the three that make up the 11 per CU its LDS allows); the compiler finds the three registers
// without spilling.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void nvx_fir_cascade_cic3_1(nvx_cascade_args a) { cascade_wave_main<true, 1, 3, const nvx_cascade_args &>(a); }
__global__ __launch_bounds__(64) void nvx_fir_cascade_cic3_2(nvx_cascade_args a) { cascade_wave_main<true, 2, 3, const nvx_cascade_args &>(a); }

// The same kernels for launches that name their streams (LIST).
template <bool RAW, int NCH, int S0>
__global__ __launch_bounds__(64) void nvx_fir_cascade_list(nvx_cascade_args a)
{
    if (RAW) cascade_wave_main<RAW, NCH, S0, const nvx_cascade_args &, true>(a);
    else cascade_wave_main<RAW, NCH, S0, const nvx_cascade_args, true>(a);
}

// Twelve kernels in all, every one launched by the GPU suite (tests/test_isa.py lists them): {252 kS/s, raw rate} x {one,
// two chains} x {every stream, list}, and for raw-rate input each of those once more with the third-order stage 0.
template <bool RAW, int NCH, int S0, bool LIST> struct CascadeKernel { static constexpr auto fn = nvx_fir_cascade_list<RAW, NCH, S0>; };
template <bool RAW, int NCH> struct CascadeKernel<RAW, NCH, 1, false> { static constexpr auto fn = nvx_fir_cascade<RAW, NCH>; };
template <> struct CascadeKernel<true, 1, 3, false> { static constexpr auto fn = nvx_fir_cascade_cic3_1; };
template <> struct CascadeKernel<true, 2, 3, false> { static constexpr auto fn = nvx_fir_cascade_cic3_2; };

// ===========================================================================
// launcher (C linkage, called from the host runtime)
// ===========================================================================
#define NVX_MAX_DEVICES 64
// test switches: both unit forms ship and the launcher picks one per launch; the suite forces each (read once per process)
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }

template <bool RAW, int NCH, int S0 = 1, bool LIST = false>
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
            if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, CascadeKernel<RAW, NCH, S0, LIST>::fn, 64, 0);
            if (e != hipSuccess) return e;
            n_cus = cus; fit_per_cu = per_cu > 0 ? per_cu : 1;
            if (dev >= 0 && dev < NVX_MAX_DEVICES) { cus_of[dev] = n_cus; fit_of[dev] = fit_per_cu; }
        } else {
            n_cus = cus_of[dev]; fit_per_cu = fit_of[dev];
        }
    }
    int per_cu = fit_per_cu;
    if (a->max_waves_per_cu > 0 && a->max_waves_per_cu < per_cu) per_cu = a->max_waves_per_cu;
    if (a->max_waves_per_cu < 0 && per_cu + a->max_waves_per_cu >= 4) per_cu += a->max_waves_per_cu;     // "so many fewer than fit"
    const int resident = n_cus * per_cu;
    nvx_cascade_args args = *a;
    // Fewer streams than resident waves: the units of one stream would run one after the other and most of the
    // chip would idle.  Then every unit rebuilds its filter histories from the nine passes in front of it
    // (+2.9 % input) and all of them run at once.  NVX_INDEPENDENT=0/1 forces the choice (tests, A/B runs).
    static const int force = env_int("NVX_INDEPENDENT", -1);
    args.independent = force >= 0 ? force : (a->n_streams < resident && a->n_frames > 1);
    // The last frame of a launch goes out in thirds when the launch is longer than one round of the grid: the waves that
    // get no unit in the last round idle for a third of a frame instead of a whole one (4096 x 12: 4.1 % of all wave
    // time was that idle tail).  r3: a launch of independent units that leaves two thirds of the chip idle even so -- one
    // or a few channels replayed from a recording -- goes out in thirds THROUGHOUT: three times the units, each with its
    // own nine-pass pre-roll (+8.6 % input), a third of the time.
    const long long whole_units = (long long)a->n_streams * a->n_frames;
    const int split_frames = whole_units > resident ? 1 : ((args.independent && 3 * whole_units <= resident) ? a->n_frames : 0);
    args.split_from = a->n_frames - split_frames;
    const long long units = (long long)a->n_streams * (args.split_from + 3LL * split_frames);
    const unsigned grid = (unsigned)(units < resident ? units : resident);
    // hand-over launches: a unit whose predecessor is still running rebuilds its histories instead of waiting for it
    // (NVX_DYNAMIC_PREROLL=0: it waits, as in round 1)
    static const int dynamic = env_int("NVX_DYNAMIC_PREROLL", 1);
    args.dynamic_preroll = dynamic != 0;
    hipLaunchKernelGGL((CascadeKernel<RAW, NCH, S0, LIST>::fn), dim3(grid), dim3(64), 0, s, args);
    return hipGetLastError();
}

extern "C" hipError_t nvx_launch_cascade(const nvx_cascade_args *a, int raw, int nch, hipStream_t s)
{
    // queue counter, status word and per-stream completion counts start at zero every launch
    hipError_t e = hipMemsetAsync(a->queue, 0, (size_t)(NVX_CASCADE_CTRL_INTS + a->n_streams) * sizeof(int), s);
    if (e != hipSuccess) return e;
    const bool cic3 = raw && a->stage0_order == 3;
    if (a->part) {
        if (cic3) return nch == 1 ? launch_cascade_as<true, 1, 3, true>(a, s) : launch_cascade_as<true, 2, 3, true>(a, s);
        if (raw) return nch == 1 ? launch_cascade_as<true, 1, 1, true>(a, s) : launch_cascade_as<true, 2, 1, true>(a, s);
        return nch == 1 ? launch_cascade_as<false, 1, 1, true>(a, s) : launch_cascade_as<false, 2, 1, true>(a, s);
    }
    if (cic3) return nch == 1 ? launch_cascade_as<true, 1, 3>(a, s) : launch_cascade_as<true, 2, 3>(a, s);
    if (raw) return nch == 1 ? launch_cascade_as<true, 1>(a, s) : launch_cascade_as<true, 2>(a, s);
    return nch == 1 ? launch_cascade_as<false, 1>(a, s) : launch_cascade_as<false, 2>(a, s);
}
