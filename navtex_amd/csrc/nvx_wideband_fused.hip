// nvx_wideband_fused.hip -- the wideband receive path as ONE kernel (gfx950): a 2.016 MS/s stream is read from HBM once,
// channelised into eight 252 kS/s sub-bands IN LDS, and every sub-band runs the two-chain cascade (FIR1 -> mixers ->
// FIR2 -> FIR3) from there: 16 NAVTEX carriers per input stream, no sub-band round trip through HBM.
// (The stand-alone channeliser of header section G, nvx_channelise.hip, followed by nvx_fir_cascade<false, 2> does the same
// through a [8 x n_wide][samples] buffer in HBM: 3 x the bytes.  A wideband handle runs this kernel only.)
//
// No reference counterpart: the reference tunes ONE 252 kS/s slice (receiver/capt_sched.c:356-417); the cascade part is
// the reference's arithmetic (nvx_cascade_wave.h), the channeliser is build-owned integer arithmetic (nvx_pfb.h).
//
// Mapping.  A work unit = one frame (315 passes) of one wideband stream, pulled from the same kind of atomic queue as
// the cascade kernel's; a workgroup = 8 waves, wave k owns sub-band k (its polyphase window, mixer / FIR2 buffers and
// histories: one CascadeLds<2, false> each, 8 x 12.0 KB -- r4: the waves end at FIR2, FIR3 is nvx_fir3.hip) and the workgroup
// shares one raw window (40 halo + 2048 samples).
// A pass = 256 channeliser output instants = 2048 raw samples = 8 KiB:
//   1. every wave stores its prefetched 1-KiB piece into the raw window and requests the next pass's piece   | barrier
//   2. every wave channelises 32 instants (a lane pair per instant, one component each) and writes the eight
//      sub-band samples of every instant, as fp64, into the eight windows                                     | barrier
//   3. every wave runs one cascade pass on its own window (CascadeWave<2, false>::compute_pass: FIR1, mixers, FIR2 -> y2 row in HBM)
// Two workgroup barriers per pass.  Filter histories and the 40-sample halo travel from unit (w, f) to (w, f+1)
// through HBM exactly as in the cascade kernel: sc1 accesses, every storing wave drains (vmcnt 0), workgroup
// barrier, one lane sets done[w]; the consumer's lane 0 polls, workgroup barrier, sc1 loads (MI355X_MICROARCH.md,
// "Valid forms", first row).  The grid is one workgroup per CU (LDS: 104 KB; 187 VGPRs: two waves per SIMD).
// Few streams (r3): with fewer wideband streams than resident workgroups -- the physically real case is ONE RSP's
// 2.016 MS/s capture replayed from a recording, receiver/capt_sched.c:356-417 -- the frames of a stream would run one
// after the other on 1 of 256 CUs.  The launcher then makes the units INDEPENDENT, as the cascade kernel does
// (nvx_kernels.h): a unit that is not the first of its stream in the launch starts nine passes early, from silence,
// with the real 40-sample channeliser halo in front of those passes; by its first real pass every filter history holds
// exactly what the hand-over would have delivered (the horizon argument of nvx_kernels.h: 2153 < 2304 sub-band
// samples, every batch counter back at 0), so the results are bit-identical; +2.9 % input.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>

#include "nvx_cascade_wave.h"
#include "nvx_pfb.h"

#define WB_SPIN_LIMIT (1 << 22)
#define WB_PASS_WORDS 2048

struct WidebandLds {
    CascadeLds<2, false> sub[NVX_WB_SUBBANDS_K];           // the waves end at FIR2 (nvx_kernels.h: FIR3 is nvx_fir3.hip)
    __attribute__((aligned(16))) unsigned raw[40 + WB_PASS_WORDS];
    int unit, ok, bad;
};

// (the argument block by reference: fields are fetched from the kernarg segment where they are used instead of sitting in
// scalar registers from the start -- 18.54 against 18.67 ms, same box, interleaved; see nvx_cascade.hip, cascade_wave_main)
__device__ __forceinline__ void wideband_main(const nvx_wideband_args &a)
{
    __shared__ WidebandLds L;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    CascadeWave<2, false> cw;
    cw.init(&L.sub[wave], lane);
    // a unit = a frame, or (a.thirds: independent units only) a third of one -- 105 passes, every pending buffer empty there
    // too (nvx_kernels.h), three times the units for launches that would leave most of the chip idle
    const int per_frame = a.thirds ? 3 : 1, unit_passes = NVX_PASSES_PER_FRAME / per_frame, unit_y3 = NVX_Y3_PER_FRAME / per_frame;
    const int n_units = a.n_wide * a.n_frames * per_frame;
    // channeliser: lane = (instant, component); instant m = 32 * wave + (lane >> 1) of the pass -> phase m & 7,
    // entry XH + (m >> 3) of every sub-band's window, component lane & 1
    const int xslot = ((lane >> 1) & 7) * XS + XH + 4 * wave + (lane >> 4);

    for (;;) {
        // One lane dequeues for the workgroup.  The value goes through readfirstlane and is stored by the WHOLE of wave 0:
        // with a lane-divergent store in front of the barrier the compiler threads the lanes that skip it straight to the
        // barrier of the next iteration, wave 0's lane 0 then never gets to dequeue again (seen: every unit but the first
        // re-ran unit 0 forever).
        int next = 0;
        if (tid == 0) next = __hip_atomic_fetch_add(a.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        next = __builtin_amdgcn_readfirstlane(next);
        if (wave == 0) { L.unit = next; L.bad = 0; }
        __syncthreads();
        const int u = L.unit;
        if (u >= n_units) break;
        const int part = u / a.n_wide;                  // frame (or third of a frame) of the launch
        const int entry = u - part * a.n_wide;          // position in the launch's list of participants (nvx_kernels.h, nvx_part)
        int w = entry, parity = 0;                      // wideband stream, and which of its state blocks it reads
        unsigned third0 = a.third0;                     // where the launch starts in the stream, in thirds of a frame since reset
        if (a.part) {
            const unsigned long long e = nvx_load_const_u64(a.part + entry); w = (int)(unsigned)e; parity = (int)(e >> 32);                 // { stream, parity }
            third0 = (unsigned)(nvx_load_const_u64(&a.part[entry].g0) / NVX_THIRD_Y3);
        }
        int *const done = a.done + entry;
        const int s = NVX_WB_SUBBANDS_K * w + wave;     // decoded 252 kS/s stream of this wave
        const unsigned mask = a.chain_masks[s];

        // independent units: rebuild the histories from the nine passes in front of the unit (above)
        bool preroll = a.independent && part > 0;
        int pre = preroll ? NVX_PREROLL_PASSES : 0;
        // the input does not depend on the predecessor: request this wave's piece of the first pass now
        const uint32_t *unit0 = a.raw + ((size_t)w * a.pitch + a.first_sample) + ((size_t)part * unit_passes - (size_t)pre) * WB_PASS_WORDS;
        const u32x4 *src = (const u32x4 *)unit0 + wave * 64 + lane;
        u32x4 pf = __builtin_nontemporal_load(src);

        // ------------------------------------------------------ wait for (w, part-1)
        if (part > 0 && !a.independent) {
            if (wave == 0) {                            // wave-uniform; lane 0 polls, the whole wave agrees on the answer
                int spins = 0, ok = 0;
                do {
                    int d = 0;
                    if (lane == 0) d = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    d = __builtin_amdgcn_readfirstlane(d);
                    ok = d >= part;
                    if (!ok) __builtin_amdgcn_s_sleep(32);
                } while (!ok && ++spins < WB_SPIN_LIMIT);
                if (spins > 0 && lane == 0) {
                    __hip_atomic_fetch_add(a.status + 1, spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(a.status + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!ok && lane == 0) __hip_atomic_store(a.status, NVX_STATUS_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // give up loudly rather than hang
                L.ok = ok;
            }
            __syncthreads();                            // the other waves load the state only behind the poll
            if (!L.ok) break;
        }

        // ------------------------------------------------------ state in (sc1 loads)
        double2 *st = (double2 *)((parity ? a.state[0] : a.state[1]) + (size_t)s * NVX_CASCADE_STATE_BYTES);
        const double2 *st_in = (part == 0) ? (const double2 *)((parity ? a.state[1] : a.state[0]) + (size_t)s * NVX_CASCADE_STATE_BYTES) : st;
        const uint32_t *hin = (((part == 0) == (parity == 0)) ? a.hist[0] : a.hist[1]) + (size_t)w * 40;
#ifdef NVX_INJECT_STALE
        // fault-injection build (tests; see nvx_cascade.hip): every NVX_INJECT_STALE-th hand-over reads the blocks the
        // stream's PREVIOUS launch left (sub-band 3's wave only, or the halo only, or everything, by turns)
        if (part > 0 && !preroll && (u % NVX_INJECT_STALE) == 0) {
            const int how = (u / NVX_INJECT_STALE) % 3;
            if (how == 0 ? wave == 3 : how == 2) st_in = (const double2 *)((parity ? a.state[1] : a.state[0]) + (size_t)s * NVX_CASCADE_STATE_BYTES);
            if (how >= 1) hin = (parity ? a.hist[1] : a.hist[0]) + (size_t)w * 40;
        }
#endif
        cw.set_mask(mask);
        if (!preroll) {
            NVX_WAVE_LDS_FENCE();
            const unsigned long long sealed = seal_load(st_in);
            unsigned long long fold = cw.state_in(st_in);
            // the 40-sample channeliser halo travels with sub-band 7's block: wave 7 stores it, loads it and seals it
            if (wave == 7 && lane < 40) {
                const uint32_t x = __hip_atomic_load(hin + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                L.raw[lane] = x;
                fold ^= seal_rotc<41>((unsigned long long)x);
            }
            // The seal (nvx_kernels.h): what was loaded must be what the predecessor stored, whole and of the right position
            // (position 0: the zeros of nvx_reset, never stored by a unit)
            const unsigned third_in = third0 + (unsigned)part * (a.thirds ? 1u : 3u);
            if (third_in != 0 && !seal_ok(sealed, wave_fold64(fold), s, third_in)) L.bad = 1;
        }
        __syncthreads();                                // (every wave's verdict is in; wave 7's halo is in the raw window)
        if (L.bad) {
            if (part == 0) {
                // inherited through a kernel boundary: nothing to fall back on; the launch is reported as failed (and runs
                // on, so that nobody is left waiting; the host discards its results)
                if (tid == 0) __hip_atomic_store(a.status, NVX_STATUS_INTEGRITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // a stale or torn hand-over in at least one of the nine blocks: the whole workgroup takes the independent
                // units' path -- nine passes early from silence, the real samples in front of them as the halo --
                // bit-identical by construction; counted
                if (tid == 0) __hip_atomic_fetch_add(a.status + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                preroll = true; pre = NVX_PREROLL_PASSES;
                unit0 -= (size_t)pre * WB_PASS_WORDS;
                src = (const u32x4 *)unit0 + wave * 64 + lane;
                pf = __builtin_nontemporal_load(src);
            }
        }
        // (a frame starts at mixer index 0, a third at 6720 * third mod 9, and a pre-roll 9 * 64 = 0 mod 9 outputs earlier at
        // the same index; FIR3 outputs of the pre-roll are not written)
        cw.begin_unit(mask, a.y3, (size_t)(s * 2) * a.y3_cap + a.y3_base + (size_t)part * unit_y3, a.y3_cap,
                      a.thirds ? ((part % 3) * (NVX_THIRD_PASSES * 64)) % NVX_MIX_N : 0,
                      preroll ? NVX_PREROLL_U : 0, preroll ? NVX_PREROLL_Y2 : 0, !preroll);
        {
            const unsigned long long rows = nvx_load_const_u64(a.y2_row + 2 * s);             // the rows of this sub-band's two chains
            cw.begin_unit_y2(parity ? a.y2[1] : a.y2[0], a.y2_pitch, (int)(unsigned)rows, (int)(unsigned)(rows >> 32), a.y3_base * 10 + (size_t)part * (NVX_Y2_PER_FRAME / per_frame));
        }
        NVX_WAVE_LDS_FENCE();
        if (preroll) {
            cw.state_zero();
            if (wave == 7 && lane < 40) L.raw[lane] = unit0[lane - 40];       // the real samples in front of the pre-roll (this launch's own input)
        }
        NVX_WAVE_LDS_FENCE();

        const int n_pass = pre + unit_passes;
        for (int pass = 0; pass < n_pass; pass++) {
            if (pass == pre) { cw.emit = true; cw.n3_done = 0; }
            // ---- 1. this wave's 1 KiB of the pass into the raw window; next pass's piece requested
            *(u32x4 *)&L.raw[40 + 256 * wave + 4 * lane] = pf;
            src += WB_PASS_WORDS / 4;
            if (pass + 1 < n_pass) pf = __builtin_nontemporal_load(src);
            __syncthreads();                            // raw window complete; every wave is done with the last pass's windows
            // ---- 2. channeliser: every wave 32 instants, a lane pair per instant (one component each); instant m's
            //         48-word window is raw[8m .. 8m+47]
            {
                int y[8];
                nvx_pfb_instant_split(&L.raw[8 * (32 * wave + (lane >> 1))], lane & 1, y);
#pragma unroll
                for (int k = 0; k < NVX_WB_SUBBANDS_K; k++)
                    ((double *)&L.sub[k].X[xslot])[lane & 1] = (double)y[k];            // capt_sched.c:511: (double) of each short
            }
            __syncthreads();                            // windows filled; raw window consumed
            // ---- 3. the newest 40 raw samples are the halo of the next pass (wave 7 owns that end of the window: its
            //         own next store in step 1 follows this copy in its LDS queue); then the cascade pass
            if (wave == 7 && lane < 40) { const unsigned t = L.raw[WB_PASS_WORDS + lane]; NVX_WAVE_LDS_FENCE(); L.raw[lane] = t; }
            cw.compute_pass();
        }

        // ------------------------------------------------------ state out (sc1 stores), publish
        // (independent units: only the stream's last unit of the launch carries state into the next launch)
        NVX_WAVE_LDS_FENCE();
        if (!a.independent || part + 1 == a.n_frames * per_frame) {
            unsigned long long fold = cw.state_out(st);
            if (wave == 7 && lane < 40) {
                const uint32_t x = L.raw[lane];
                __hip_atomic_store((parity ? a.hist[0] : a.hist[1]) + (size_t)w * 40 + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                fold ^= seal_rotc<41>((unsigned long long)x);
            }
            fold = wave_fold64(fold);
            if (lane == 0) seal_store(st, fold, s, third0 + (unsigned)(part + 1) * (a.thirds ? 1u : 3u));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains ...
        __syncthreads();                                            // ... before the one lane that signals for all of them
        // (the whole of wave 0 stores the same word: a lane-divergent store here is merged with the dequeue at the top of
        // the loop into one "lane 0" region around the back-edge -- the deadlock described there)
        if (wave == 0 && !a.independent) __hip_atomic_store(done, part + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(512) void nvx_wideband_fused(nvx_wideband_args a) { wideband_main(a); }

extern "C" hipError_t nvx_launch_wideband_fused(const nvx_wideband_args *a, hipStream_t s)
{
    // persistent grid: as many 8-wave workgroups as the device of this launch holds at once (LDS: one per CU)
    static std::mutex mu;
    static int resident_of[64];
    int resident = 0;
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(mu);
        if (dev >= 0 && dev < 64 && resident_of[dev] > 0) resident = resident_of[dev];
        else {
            int cus = 0, per_cu = 0;
            e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nvx_wideband_fused, 512, 0);
            if (e != hipSuccess) return e;
            resident = cus * (per_cu > 0 ? per_cu : 1);
            if (dev >= 0 && dev < 64) resident_of[dev] = resident;
        }
    }
    hipError_t e = hipMemsetAsync(a->queue, 0, (size_t)(NVX_CASCADE_CTRL_INTS + a->n_wide) * sizeof(int), s);
    if (e != hipSuccess) return e;
    const long long units = (long long)a->n_wide * a->n_frames;
    // Fewer streams than resident workgroups: independent units (pre-roll instead of hand-over), all frames at once.
    // NVX_INDEPENDENT=0/1 forces the choice (tests, A/B runs), as for the cascade kernel.
    nvx_wideband_args args = *a;
    static const int force = getenv("NVX_INDEPENDENT") ? atoi(getenv("NVX_INDEPENDENT")) : -1;
    args.independent = force >= 0 ? force : (a->n_wide < resident && a->n_frames > 1);
    // ... and in thirds when even that leaves two thirds of the chip idle (one RSP capture replayed)
    args.thirds = args.independent && 3 * units <= resident;
    const long long all_units = units * (args.thirds ? 3 : 1);
    const unsigned grid = (unsigned)(all_units < resident ? all_units : resident);
    hipLaunchKernelGGL(nvx_wideband_fused, dim3(grid), dim3(512), 0, s, args);
    return hipGetLastError();
}
