/* nvx_kernels.h -- argument blocks of the gfx950 kernels (internal). */
#ifndef NVX_KERNELS_H
#define NVX_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nvx_synth.h"

#define NVX_PASSES_PER_FRAME 315          /* 20160 FIR1 outputs per frame / 64 per pass */
#define NVX_Y3_PER_FRAME 288
/* A work unit of the cascade is a frame or a third of one.  A third of a frame (105 passes =
 * 6720 FIR1 outputs = 960 FIR2 outputs = 96 FIR3 outputs) still ends with every pending buffer
 * empty (6720 = 30 * 224 = 60 * 112, 960 = 6 * 160 = 12 * 80); only the mixer index does not
 * return to 0 (6720 mod 9 = 6).  Thirds for EVERY frame lose (round 1: 20.58 vs 20.39-20.53 ms; round 2, with the
 * dynamic pre-roll: 21.0 vs 20.6): a switch of unit costs more than its state traffic.  Thirds for the LAST frame of a
 * launch shorten the ragged end of the persistent grid from a frame to a third (nvx_cascade.hip, profiles/TUNING.md). */
#define NVX_THIRD_PASSES (NVX_PASSES_PER_FRAME / 3)
#define NVX_THIRD_Y3 (NVX_Y3_PER_FRAME / 3)
/* r4: the waves of the fused wideband kernel (nvx_wideband_fused) end at FIR2: their 9 kS/s outputs go to HBM and FIR3
 * (71 taps, /10, receiver/fir3cpp.C:22-60) is a kernel of its own (nvx_fir3.hip) in front of the demodulator, beside the
 * next launch.  Inside the wave FIR3 ran on half of the lanes; out of it the fused kernel is 4.8 % faster and so is the
 * step.  The single-wave cascade kernels keep FIR3 inside (252 kS/s input: 2.4 % in the kernel, nothing in the step;
 * raw-rate input: HBM-bound, the extra traffic would cost more) -- nvx_fir3.hip, profiles/TUNING.md.
 * y2 rows: one per ACTIVE chain (y2_row[slot], -1 = none), NVX_Y2_PREFIX entries in front of the launch's own outputs;
 * entries PREFIX-70 .. PREFIX-1 hold the last 70 outputs of the chain's previous launch (FIR3's history).  Two buffers:
 * a stream of parity p writes the outputs of its launch into [p]; nvx_fir3 reads [p] and leaves the tail in [p ^ 1]'s
 * prefix for the stream's next launch -- while the next cascade launch already fills [p ^ 1]'s output region.        */
#define NVX_Y2_PER_FRAME 2880
#define NVX_THIRD_Y2 (NVX_Y2_PER_FRAME / 3)
#define NVX_Y2_PREFIX 80
/* Independent units (launches with fewer streams than resident waves): a unit that is not the first of its
 * stream in the launch rebuilds the filter histories from the input instead of waiting for its predecessor.
 * Every value a real output uses must come from real samples in the reference's operation order:
 *   y3[k >= 0] uses y2[>= -61]; the carried history needs y2[-70..-1]; y2[-70] uses u[-530..-484];
 *   u[m] uses x[4m-33 .. 4m+3]  ->  x from -2153 on (252 kS/s samples before the unit).
 * Nine passes (2304 samples) cover that, and with 96 mixer outputs and 64 FIR2 outputs "pending" at their start
 * (zeros, consumed by runs whose outputs lie before the horizon) every batch counter is back at 0 exactly at the
 * unit's first real pass: 96 + 9*64 = 672 = 3*224 = 6*112, 64 + 96 = 160 = 2*80; 576 = 0 mod 9 for the mixer. */
#define NVX_PREROLL_PASSES 9
#define NVX_PREROLL_U 96
#define NVX_PREROLL_Y2 64
#define NVX_CASCADE_CTRL_INTS 8           /* [0] work-queue counter, [1] status (1 = spin timeout, 2 = the state a launch   */
                                         /* inherited failed its integrity word), [2] polls spent waiting for a            */
                                         /* predecessor, [3] units that waited, [4] hand-overs whose state block failed its */
                                         /* integrity word and were repaired by a pre-roll (r4), [5..7] unused              */
#define NVX_STATUS_TIMEOUT 1
#define NVX_STATUS_INTEGRITY 2
#define NVX_STATUS_INTS 4                 /* ints [1..4] travel to the host behind every launch                             */
/* carried FIR state per stream: 36 x {I,Q} @252 kS/s, then per chain 46 mixer
 * outputs and 70 FIR2 outputs, all fp64 pairs; entry NVX_STATE_CIC3: the third-order stage 0's
 * two blocks of history as four integers (C_I | t_I << 32, C_Q | t_Q << 32);
 * entry NVX_STATE_SEAL (r4): the block's integrity word and its tag in clear (below); padded to
 * whole 128-byte lines so that the blocks of neighbouring streams share no cache line         */
#define NVX_STATE_DATA_ENTRIES (36 + 2 * (46 + 70))
#define NVX_STATE_CIC3 NVX_STATE_DATA_ENTRIES
#define NVX_STATE_SEAL (NVX_STATE_DATA_ENTRIES + 1)
#ifndef NVX_CASCADE_STATE_ENTRIES
#define NVX_CASCADE_STATE_ENTRIES 272
#endif
#define NVX_CASCADE_STATE_BYTES   (NVX_CASCADE_STATE_ENTRIES * 16)
/* The seal.  The state block is the one piece of memory a unit writes and ANOTHER unit -- usually on another XCD, behind
 * another L2 -- reads within a launch, and the hand-over rests on per-instruction device coherence (sc1) instead of
 * cache-wide fences: measured behaviour of gfx950, not an architectural guarantee (MI355X_MICROARCH.md).  So a block
 * carries a 64-bit word over everything the unit stored (every 8-byte pattern rotated by its position and XOR-folded)
 * mixed with the block's TAG = (stream, the stream's position since reset in thirds of a frame -- which is also the
 * launch sequence: no two blocks a stream ever writes carry the same tag).  The consumer recomputes it over what it
 * LOADED; on a mismatch a hand-over unit takes the pre-roll path (bit-identical by construction) and counts the event
 * (status int [4]); the first unit of a launch, whose predecessor state came through a kernel boundary, reports
 * NVX_STATUS_INTEGRITY instead.  What the carried state restates: receiver/fir1cpp.C:51-60, receiver/fir2cpp.C:74-83,
 * receiver/fir3cpp.h:90-95 (the reference keeps it in statics).                                                      */
#define NVX_DEMOD_DOUBLES (8 + 8 + 8 + 567) /* per slot, contiguous: last 4 IQ, dphi, class sums, |corr| */
#define NVX_DEMOD_INTS    5
#define NVX_DI_PHASE       3            /* bit-FSM phase, resets to -1 (waiting)            */
#define NVX_DI_PREV_OFFSET 4            /* resets to -1 (decoder.C:30)                      */

/* Which streams of a handle a launch covers.  The chains share no state (receiver/decoder.h:31-60,
 * receiver/nav_b_sm.h:92-114; FIR1 / mixer per stream: receiver/fir1cpp.C:57-60, receiver/fir2cpp.C:74-83), so nothing ties
 * the streams of one handle to a common clock: a launch names the streams that HAVE a frame (a radio that stalls or is
 * unplugged -- receiver/capt_sched.c:210-212 only prints sdrplay_api_DeviceRemoved -- must not freeze the others).
 * One entry per participating stream; a NULL list = every stream, in index order, with the launch-wide parity and g0.
 * Rows of every per-stream buffer (input, state blocks, y3, words, bits) are indexed by `stream`; the work queue's
 * done[] and the unit numbering by the entry's position in the list.                                              */
typedef struct {
    int stream;                /* stream index (input stream: a wideband handle's 2.016 MS/s stream)             */
    int parity;                /* the stream reads state block [parity] and writes [parity ^ 1]                  */
    unsigned long long g0;     /* 900 S/s samples the stream has been through since reset (multiple of 288)      */
    int n3;                    /* 900 S/s samples of this launch the stream's REAL input produces: frames * 288, or fewer */
                               /* in the launch that ends the stream (nvx_finish: the demodulator stops there; the        */
                               /* reference's loop stops with its last sample, receiver/capt_sched.c:509-513)              */
    int reserved;
} nvx_part;

typedef struct {
    const uint32_t *iq;        /* [all streams][pitch] packed int16 I | Q<<16          */
    size_t pitch;              /* complex samples between streams                      */
    size_t first_sample;       /* first complex sample of this launch in every stream  */
    int n_frames;
    int n_streams;             /* streams taking part in this launch                   */
    const nvx_part *part;      /* [n_streams], or NULL: streams 0 .. n_streams-1, all of parity 0 */
    const uint8_t *chain_masks;
    uint8_t *state[2];         /* [all streams][NVX_CASCADE_STATE_BYTES] each: a stream of parity p reads [p] (left by ITS previous */
                               /* launch) and writes [p ^ 1]; hand-over inside the launch goes through the written one.  Without a  */
                               /* list the host passes the block to read as [0] and the one to write as [1]                        */
    double2 *y3;               /* [all streams*2][y3_cap]                              */
    size_t y3_cap, y3_base;
    int *queue;                /* NVX_CASCADE_CTRL_INTS control ints followed by ...   */
    int *status;               /* = queue + 1                                          */
    int *done;                 /* = queue + 2: frames completed per stream             */
    int independent;           /* set by the launcher: units do not wait for their predecessor (pre-roll instead) */
    int max_waves_per_cu;      /* 0 = as many as fit; >0 caps the persistent grid; <0 = that many fewer than fit */
    int stage0_order;          /* raw-rate input: 3 = third-order stage 0, anything else = integrate-and-dump     */
    int dynamic_preroll;       /* set by the launcher: a unit whose predecessor is still running pre-rolls instead of waiting */
    int split_from;            /* set by the launcher: frames from this one on are handed out in thirds (n_frames = none) */
    unsigned third0;           /* position of the launch in its streams, in thirds of a frame since reset (g0 / 96), for the    */
                               /* state block's tag; with a list every entry carries its own g0                                */
} nvx_cascade_args;

/* FIR3 as a kernel of its own (wideband handles, fused form): y2 rows -> y3, and the rows' tails into the other
 * buffer's prefix */
typedef struct {
    double2 *y2[2];            /* FIR2 output buffers [rows][y2_pitch] (above); a stream of parity p wrote [p]; without a   */
    size_t y2_pitch;           /* list the host passes the buffer the launch wrote as [0], the other as [1]                  */
    const int *y2_row;         /* [all decoded streams * 2]: row of the chain, or -1                                         */
    double2 *y3; size_t y3_cap, y3_base;                    /* [all streams * 2][y3_cap]                        */
    int n_frames;
    int n_slots;                                            /* all streams * 2                                  */
    const nvx_part *part; int n_part, per_part;             /* participants, as nvx_demod_args (NULL = all)      */
} nvx_fir3_args;

/* How close the bit-timing arg-max (receiver/decoder.C:202-215, strict '>') comes to a tie.  The class sums it compares
 * are built from delta-phi values that may differ from glibc's atan2 in the last bit; such a difference
 * moves a sum by ~1e-16 relative, so it can only change the decision where best and runner-up are closer than that.
 * A timing evaluation counts as a NEAR TIE when best > 0 and (best - runner_up) < best * 2^-40.                          */
typedef struct {
    unsigned long long near_ties;      /* evaluations with a margin below 2^-40 relative                               */
    unsigned long long evaluations;    /* evaluations with best > 0 (silence -- all sums exactly 0 -- has no margin)    */
    unsigned int min_margin_bits;      /* smallest (best - runner_up) / best seen, as float bits (0x7f800000 = none yet) */
    unsigned int pad;
} nvx_tie_stats;

typedef struct {
    const double2 *y3;
    size_t y3_cap, y3_base;
    int n3;                    /* 900 S/s samples in this launch (whole frames); part != NULL: per entry, and the last period may be ragged */
    int n_slots;               /* all streams * 2                                      */
    const uint8_t *slot_active;
    unsigned long long g0;     /* 900 S/s samples processed since reset (multiple of 288); part != NULL: per entry */
    const nvx_part *part;      /* participating streams of this launch, or NULL = all  */
    int n_part;                /* entries of part                                      */
    int per_part;              /* decoded streams per entry: 8 on a wideband handle (stream 8 * w + k), else 1 */
    double *dstate[2];         /* [n_slots][NVX_DEMOD_DOUBLES] each: a chain of parity p reads [p] and writes [p ^ 1], as the    */
                               /* cascade state; without a list the host passes (read, write) as ([0], [1])                      */
    int *state_i;              /* [field][n_slots]                                     */
    const uint32_t *fsm_table; /* NVX_FSM_TABLE_ALLOC entries (nvx_fsm.h), 16-byte aligned */
    unsigned short *words;     /* [n_slots][y3_cap/9] per-bit-period hand-over, front -> fsm (rows 16-byte aligned) */
    uint8_t *bits; int bits_cap; int *nbits;   /* bits: packed, B = 1, LSB first; bits_cap bytes (multiple of 4) per slot */
    double *dphi;              /* optional debug tap, same layout as y3 (or NULL)      */
    nvx_tie_stats *ties;       /* cumulative arg-max margin statistics (never NULL)    */
} nvx_demod_args;

typedef struct {
    const nvx_synth_desc *desc;
    const nvx_period *pool;
    uint32_t *out; size_t pitch; size_t n; uint32_t spb;
} nvx_synth_args;

typedef struct {
    const uint32_t *raw;       /* [n_wide][pitch_raw] packed IQ at 2.016 MS/s                 */
    size_t pitch_raw, first_sample;
    int n_wide;
    size_t n_out;              /* outputs per sub-band in this launch (multiple of 64)         */
    const uint32_t *hist_in;   /* [n_wide][40] raw words preceding first_sample, or NULL (= 0) */
    uint32_t *hist_out;        /* [n_wide][40] last raw words of this launch, or NULL          */
    uint32_t *sub;             /* [n_wide*8][pitch_sub] packed IQ at 252 kS/s                  */
    size_t pitch_sub, sub_first;
    int chunks_per_block;
} nvx_channelise_args;

/* fused wideband kernel (nvx_wideband_fused.hip): n_wide streams at 2.016 MS/s -> 8 * n_wide decoded 252 kS/s streams */
#define NVX_WB_SUBBANDS_K 8
typedef struct {
    const uint32_t *raw;       /* [n_wide][pitch] packed IQ at 2.016 MS/s                                    */
    size_t pitch, first_sample;
    int n_wide, n_frames;      /* n_wide: wideband streams taking part in this launch                         */
    const nvx_part *part;      /* [n_wide], or NULL: streams 0 .. n_wide-1, all of parity 0                   */
    const uint8_t *chain_masks;/* [8 * all wideband streams]                                                  */
    uint8_t *state[2];         /* cascade state blocks of the decoded streams, as nvx_cascade_args            */
    uint32_t *hist[2];         /* [all wideband streams][40] raw words in front of the launch: read [parity], */
                               /* written [parity ^ 1] (hand-over inside this launch and to the next one);    */
                               /* without a list the host passes (read, write) as ([0], [1]), like state      */
    double2 *y3; size_t y3_cap, y3_base;
    int *queue, *status, *done;/* as nvx_cascade_args; done[n_wide]                                          */
    int independent;           /* set by the launcher: units pre-roll instead of waiting for their predecessor */
    int thirds;                /* set by the launcher (independent units only): a unit is a third of a frame   */
    unsigned third0;           /* as nvx_cascade_args                                                          */
    double2 *y2[2]; size_t y2_pitch; const int *y2_row;     /* as nvx_fir3_args (rows by decoded stream * 2 + chain)     */
} nvx_wideband_args;

#ifdef __cplusplus
extern "C" {
#endif
hipError_t nvx_launch_wideband_fused(const nvx_wideband_args *a, hipStream_t s);
hipError_t nvx_launch_channelise(const nvx_channelise_args *a, hipStream_t s);
hipError_t nvx_launch_cascade(const nvx_cascade_args *a, int raw, int nch, hipStream_t s);
hipError_t nvx_launch_fir3(const nvx_fir3_args *a, hipStream_t s);
hipError_t nvx_launch_demod_front(const nvx_demod_args *a, hipStream_t s);
hipError_t nvx_launch_demod_fsm(const nvx_demod_args *a, hipStream_t s);
hipError_t nvx_launch_synth(const nvx_synth_args *a, int n_streams, hipStream_t s);
double nvx_atan2_host(double y, double x);
#ifdef __cplusplus
}
#endif
#endif
