/* nvx_fsm.h -- the two small state machines of the reference's decoder, per
 * 900 S/s sample, and the per-bit-period transition table the FSM kernel uses.
 *
 *   timing slew limiter   receiver/decoder.C:202-249  (nvx_fsm_timing)
 *   mark/space bit FSM    receiver/decoder.C:62-137   (nvx_fsm_bit_step)
 *
 * The reference's (status, burn_count, samplecount) triple only ever walks one
 * fixed path, so it is kept as a single phase counter:
 *   phase -1        STATUS_SYNCED_WAIT (or STATUS_INIT while not synced)
 *   phase 0, 1      the two burned samples (decoder.C:91-110; the sample that
 *                   matches the sync offset is itself the first of them)
 *   phase 2         the sample that flips to RECEIVING and is not used
 *   phase 3..7      the five accumulated samples; the decision falls on 7
 *
 * A bit period is nine samples; sample k of a period has bd_seq_nbr mod 9 ==
 * (k + 1) mod 9 (launches start on a multiple of 9) and the timing decision
 * falls on k == 6 (582 mod 9), BEFORE that sample's bit-FSM step.  Over one
 * period the bit FSM is a function of four small integers
 *     (phase + 1, sync_off or 9 = not synced yet, next_sync_off before the
 *      timing decision, next_sync_off after it)
 * so the kernel steps a whole period with one table lookup (and a second, 100-entry
 * one for the slew limiter).  Both tables are GENERATED from nvx_fsm_timing /
 * nvx_fsm_bit_step below -- the per-sample rule stays the only statement of the
 * behaviour -- and nvx_fsm_selftest() replays random input through both.
 */
#ifndef NVX_FSM_H
#define NVX_FSM_H

#include <stdint.h>

#if defined(__HIPCC__)
#  define NVX_FSM_HD __host__ __device__ static inline
#else
#  define NVX_FSM_HD static inline
#endif

#define NVX_FSM_TIMING_SAMPLE 6                 /* G_CSA mod 9 */
#define NVX_FSM_UNSYNCED 9                      /* sync_off code: no timing decision yet */
#define NVX_FSM_TABLE_SIZE (9 * 10 * 9 * 9)     /* bit FSM: one entry per (phase1, so, nso_old, nso_new)         */
#define NVX_FSM_TIMING_BASE NVX_FSM_TABLE_SIZE  /* slew limiter: 10 x 10 entries (prev_offset + 1, raw or 9 = none) */
#define NVX_FSM_TABLE_ALLOC 7392                /* 7290 + 100, rounded up to whole 16-byte words                 */

/* table entry: bits 0-3 phase+1 after the period, 4-7 sync_off after it (never 9),
 * 8-9 number of decisions (0..2), 10-13 / 14-17 the samples k they fall on        */
#define NVX_FSM_KEY(phase1, so, nso_old, nso_new) ((((phase1) * 10 + (so)) * 9 + (nso_old)) * 9 + (nso_new))
#define NVX_FSM_E_PHASE1(e) ((int)((e) & 15u))
#define NVX_FSM_E_SO(e)     ((int)(((e) >> 4) & 15u))
#define NVX_FSM_E_N(e)      ((int)(((e) >> 8) & 3u))
#define NVX_FSM_E_K1(e)     ((int)(((e) >> 10) & 15u))
#define NVX_FSM_E_K2(e)     ((int)(((e) >> 14) & 15u))

/* Timing decision of one bit period (decoder.C:202-249).  raw = arg-max of the nine class
 * sums (first maximum wins), 15 = class sums not primed yet.  Returns 1 and the new sync
 * offset when a decision was made.  With diff = (raw - prev) mod 9 the reference's four-way
 * branch is: diff 1..4 -> prev + 1, diff 5..8 -> prev - 1 (circular, one step per bit).      */
NVX_FSM_HD int nvx_fsm_timing(int raw, int *prev_offset, int *offset)
{
    const int have = raw != 15;
    const int prev = *prev_offset;
    const int diff = (raw - prev + 9) % 9;
    const int slew = (prev != -1) && (diff != 0);
    const int up = (prev + 1) % 9, dn = (prev + 8) % 9;
    const int mi = slew ? ((diff <= 4) ? up : dn) : raw;
    *offset = (mi + 5) % 9;                                   /* decoder.C:249 */
    *prev_offset = have ? mi : prev;
    return have;
}

/* Slew-limiter table entry: bits 0-3 prev_offset + 1 afterwards, 4-7 the new sync offset, 8 = decision made */
NVX_FSM_HD uint32_t nvx_fsm_timing_entry(int prev1, int rawc)
{
    int prev = prev1 - 1, offset = 0;
    const int have = nvx_fsm_timing(rawc == 9 ? 15 : rawc, &prev, &offset);
    return (uint32_t)(prev + 1) | ((uint32_t)offset << 4) | ((uint32_t)have << 8);
}

/* One sample of the bit FSM (decoder.C:73-137); k = index of the sample in its bit period.
 * Returns 1 when the mark/space decision of a bit falls on this sample.                     */
NVX_FSM_HD int nvx_fsm_bit_step(int k, int synced, int *phase, int *sync_off, int next_sync_off)
{
    const int start = (*phase < 0) && synced && (((k + 1) % 9) == *sync_off);
    *phase = start ? 0 : ((*phase >= 0) ? *phase + 1 : *phase);
    const int decide = *phase == 7;
    *phase = decide ? -1 : *phase;
    *sync_off = decide ? next_sync_off : *sync_off;
    return decide;
}

/* One table entry: nine samples from (phase1 - 1, so, nso_old), with the timing decision on
 * sample 6 setting next_sync_off = nso_new (and, if not synced yet, sync_off too: decoder.C:62-70). */
NVX_FSM_HD uint32_t nvx_fsm_table_entry(int phase1, int so, int nso_old, int nso_new)
{
    int synced = so != NVX_FSM_UNSYNCED;
    int phase = phase1 - 1, sync_off = synced ? so : 0, next_sync_off = nso_old;
    int n = 0, kk[2] = { 0, 0 };
    for (int k = 0; k < 9; k++) {
        if (k == NVX_FSM_TIMING_SAMPLE) {
            sync_off = synced ? sync_off : nso_new;
            next_sync_off = nso_new;
            synced = 1;
        }
        if (nvx_fsm_bit_step(k, synced, &phase, &sync_off, next_sync_off)) { if (n < 2) kk[n] = k; n++; }
    }
    return (uint32_t)(phase + 1) | ((uint32_t)sync_off << 4) | ((uint32_t)n << 8) | ((uint32_t)kk[0] << 10) | ((uint32_t)kk[1] << 14);
}

/* The registers one chain carries from bit period to bit period. */
typedef struct nvx_fsm_regs {
    int phase1;          /* phase + 1: 0 = waiting, 1..7 = inside a bit                */
    int so;              /* sync_off, NVX_FSM_UNSYNCED before the first timing decision */
    int nso;             /* next_sync_off                                              */
    int prev_offset;     /* slew limiter state, -1 before the first decision           */
} nvx_fsm_regs;

/* One bit period.  w = the front kernel's word: bits 0..8 the mark/space decision if a window
 * ended on sample k ('B' = 1), bits 12..15 the arg-max of the timing evaluation (15 = none).
 * Returns the decided bits in the low *n_out (0..2) positions, first decision lowest.       */
NVX_FSM_HD unsigned nvx_fsm_period(const uint32_t *tab, unsigned w, nvx_fsm_regs *r, int *n_out)
{
    const unsigned raw = w >> 12;
    const uint32_t t = tab[NVX_FSM_TIMING_BASE + (r->prev_offset + 1) * 10 + (int)(raw > 9u ? 9u : raw)];
    const int have = (int)((t >> 8) & 1u);
    r->prev_offset = (int)(t & 15u) - 1;
    const int nso_new = have ? (int)((t >> 4) & 15u) : r->nso;
    /* unsynced and no decision this period either: nothing can happen */
    const int live = have | (r->so != NVX_FSM_UNSYNCED);
    const uint32_t e = tab[NVX_FSM_KEY(r->phase1, r->so, r->nso, nso_new)];
    r->phase1 = live ? NVX_FSM_E_PHASE1(e) : r->phase1;
    r->so = live ? NVX_FSM_E_SO(e) : r->so;
    r->nso = nso_new;
    *n_out = live ? NVX_FSM_E_N(e) : 0;
    return ((w >> NVX_FSM_E_K1(e)) & 1u) | (((w >> NVX_FSM_E_K2(e)) & 1u) << 1);
}

/* The first `rem` (1..8) samples of a bit period -- the period in which a stream's input ends: the per-sample rule
 * itself, sample by sample (the reference's decoder stops with its last sample, receiver/capt_sched.c:509-513).  w as
 * above; its timing arg-max is only looked at when the period's sample 6 exists.  Returns the decided bits, *n_out of them. */
NVX_FSM_HD unsigned nvx_fsm_partial_period(unsigned w, int rem, nvx_fsm_regs *r, int *n_out)
{
    int synced = r->so != NVX_FSM_UNSYNCED;
    int phase = r->phase1 - 1, sync_off = synced ? r->so : 0, next_sync_off = r->nso, prev_offset = r->prev_offset;
    unsigned bits = 0;
    int n = 0;
    for (int k = 0; k < rem; k++) {
        if (k == NVX_FSM_TIMING_SAMPLE) {
            int offset = 0;
            if (nvx_fsm_timing((int)(w >> 12), &prev_offset, &offset)) {          /* decoder.C:62-70: bd_in_bit_sync */
                sync_off = synced ? sync_off : offset;
                next_sync_off = offset;
                synced = 1;
            }
        }
        if (nvx_fsm_bit_step(k, synced, &phase, &sync_off, next_sync_off)) { bits |= ((w >> k) & 1u) << n; n++; }
    }
    r->phase1 = phase + 1; r->so = synced ? sync_off : NVX_FSM_UNSYNCED; r->nso = next_sync_off; r->prev_offset = prev_offset;
    *n_out = n;
    return bits;
}

#endif
