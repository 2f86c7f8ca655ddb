/* TEST INFRASTRUCTURE ONLY -- see nvx_oracle_tables.h for the status header.
 *
 * CPU restatement ("oracle") of the reference receive path
 *     int16 IQ @252 kS/s -> FIR1 /4 -> +-14 kHz mixer -> FIR2 /7 -> FIR3 /10
 *     -> FSK discriminator + bit sync + mark/space decision -> 'B'/'Y' bits
 *     -> SITOR-B character layer -> add_message(bbbb, text, freq)
 * plus the build-owned integer stage 0 (/8 from 2.016 MS/s) that has no
 * reference counterpart (the SDRplay API does it inside a closed library,
 * receiver/capt_sched.c:412-413).
 *
 * Two forms are provided:
 *   - whole-array stage functions with zero initial state (seam parity);
 *   - nvxo_pipe: a streaming, block-based pipeline with carried state
 *     (chunk-invariance tests and the timed CPU baseline).
 */
#ifndef NVX_ORACLE_H
#define NVX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVXO_T1 37
#define NVXO_T2 47
#define NVXO_T3 71
#define NVXO_D0 8
#define NVXO_D1 4
#define NVXO_D2 7
#define NVXO_D3 10
#define NVXO_MIX_N 9
#define NVXO_SPB 9                 /* 900 S/s samples per bit (decoder.h:21)      */
#define NVXO_CORR_BITS 63          /* decoder.h:23                               */
#define NVXO_CORR_N (NVXO_CORR_BITS * NVXO_SPB)

/* ---- build-owned stage 0: integrate-and-dump /8 with round-half-up --------
 * out[m] = (sum_{j<8} raw[8m+j] + 4) >> 3, per component, arithmetic shift.  */
void nvxo_stage0(const int16_t *raw_iq, size_t n_out, int16_t *out_iq);
/* ---- build-owned stage 0, third-order form (three cascaded 8-sample boxcars, decimated by 8):
 * out[m] = (sum_{j<22} w[j] * raw[8m+7-j] + 256) >> 9,  w = (1,1,..,1)^*3 = 1 3 6 10 15 21 28 36 42 46 48 48 46 ... 3 1
 * (sum 512), per component, floor shift; raw[n<0] = hist14 (the 14 samples in front, oldest first; NULL = silence),
 * which the call replaces by the last 14 samples it saw.  Alias rejection at the NAVTEX offsets (+-14 kHz +- 85 Hz
 * around multiples of 252 kHz): 3 x the boxcar's 25.4 dB = 76 dB.                                                    */
void nvxo_stage0_cic3(const int16_t *raw_iq, size_t n_out, int16_t *hist14, int16_t *out_iq);

/* ---- build-owned wideband front-end (no reference counterpart):
 * 8-channel maximally decimated polyphase channeliser, integer arithmetic only.
 * Sub-band k (0..7) is the 252 kHz-wide slice centred at k * 252 kHz (k >= 4: negative
 * frequencies), delivered at 252 kS/s.  Exact definition, with x[n<0] = hist (40 samples,
 * oldest first) or zero:
 *   u[p] = ( sum_{j = p (mod 8), j < 48} h[47-j] * x[8m - 40 + j]  + 16 ) >> 5      (int32, per component)
 *   Y[k] = 8-point DFT of u (radix-2 DIT below; 45-degree twiddles as 23170/2^15, floor shifts)
 *   out[k][m] = clamp_int16( (Y[k] + 4096) >> 13 )
 * out is [8][n_out] interleaved I,Q.                                                   */
void nvxo_channelise(const int16_t *raw_iq, size_t n_out, const int16_t *hist40, int16_t *out);

/* ---- whole-array stage functions, zero history (fir1cpp.C:80-136 etc.) ---- */
size_t nvxo_fir1(const int16_t *iq, size_t n, double *y1);          /* returns n/4  */
void   nvxo_mixer_table(double cr[NVXO_MIX_N], double ci[NVXO_MIX_N]);
void   nvxo_mix(const double *y1, size_t n1, int chain, double *u); /* chain 0=518, 1=490 */
size_t nvxo_fir2(const double *u, size_t n, double *y2);            /* returns n/7  */
size_t nvxo_fir3(const double *y2, size_t n, double *y3);           /* returns n/10 */

/* ---- decoder (decoder.h:26-88, decoder.C) --------------------------------- */
typedef struct nvxo_dec {
    int    bs_seq_nbr, status;
    float  BR, BI, YR, YI;
    int    samplecount, bit_sync_offset, next_bit_sync_offset, burn_count;
    float  fR[5], fI[5];
    double prevI, prevQ;
    double dab[NVXO_SPB], cb[NVXO_CORR_N], csa[NVXO_SPB];
    int    dab_index, cb_index, csa_index, dab_primed, cb_primed, csa_primed;
    int    prev_offset, bd_seq_nbr;
    /* taps for tests (not reference state) */
    double last_dphi;
    int    last_sync;              /* offset passed to bd_in_bit_sync this sample, or -1 */
} nvxo_dec;
void nvxo_dec_init(nvxo_dec *d);
/* returns 0 (no bit), 'B' or 'Y' */
int  nvxo_dec_push(nvxo_dec *d, double I, double Q);
/* whole-array convenience: bits_out must hold n3 chars; returns number of bits */
size_t nvxo_decode(const double *y3, size_t n3, char *bits_out, double *dphi_out);
void nvxo_bitfilter_table(float fR[5], float fI[5]);
/* test probe: nvxo_decode with bd_seq_nbr (decoder.h:60) set to `value` in front of sample `at`; *bd_seq_nbr_at (optional)
 * receives the value it had there                                                                                    */
size_t nvxo_decode_inject(const double *y3, size_t n3, char *bits_out, size_t at, int value, int *bd_seq_nbr_at);
/* test hook: same as nvxo_decode but with the discriminator's atan2 supplied by the caller
 * (NULL = libm).  Used to measure whether replacing glibc's atan2 by another one that
 * differs in the last bit ever changes a decoded bit.                                    */
typedef double (*nvxo_atan2_fn)(double, double);
size_t nvxo_decode_with(const double *y3, size_t n3, char *bits_out, nvxo_atan2_fn fn, size_t *dphi_mismatch);

/* ---- SITOR-B character layer (nav_b_sm.h / nav_b_sm.C) -------------------- */
typedef void (*nvxo_msg_cb)(void *user, const char *bbbb, const char *message, int freq);
typedef struct nvxo_sm nvxo_sm;
nvxo_sm *nvxo_sm_new(int freq, nvxo_msg_cb cb, void *user);
void     nvxo_sm_free(nvxo_sm *s);
void     nvxo_sm_bit(nvxo_sm *s, char bit);
/* everything the reference would have printf'ed, byte for byte */
const char *nvxo_sm_trace(nvxo_sm *s, size_t *len);

/* ---- streaming pipeline ---------------------------------------------------- */
typedef struct nvxo_pipe nvxo_pipe;
/* chain_mask bit0 = 518 chain (down-mix), bit1 = 490 chain (up-mix); freq
 * labels are what the character layer reports to the message callback.       */
nvxo_pipe *nvxo_pipe_new(int chain_mask, int freq0, int freq1, nvxo_msg_cb cb, void *user);
void       nvxo_pipe_free(nvxo_pipe *p);
/* push n complex samples at 252 kS/s; any n */
void       nvxo_pipe_push(nvxo_pipe *p, const int16_t *iq252, size_t n);
/* push n_out*8 complex samples at 2.016 MS/s through stage 0 first */
void       nvxo_pipe_push_raw(nvxo_pipe *p, const int16_t *raw_iq, size_t n_out);
/* which stage 0 push_raw applies: 1 (default) = integrate-and-dump, 3 = third-order form (history carried in the pipe) */
void       nvxo_pipe_set_stage0(nvxo_pipe *p, int order);
/* bits decoded so far on chain c (NUL-terminated, grows) */
const char *nvxo_pipe_bits(nvxo_pipe *p, int chain, size_t *n);
/* optional seam taps: append every y3 sample of chain c to a caller buffer */
void       nvxo_pipe_tap_y3(nvxo_pipe *p, int chain, double *buf, size_t cap_pairs, size_t *count);
/* test probe: the reference's init functions called again in mid-stream (which & 1: init_fir_filter1, & 2: init_fir2_wrapper) */
void       nvxo_pipe_reinit(nvxo_pipe *p, int which);
/* when set, the character layer is skipped (bit-level runs) */
void       nvxo_pipe_set_charlayer(nvxo_pipe *p, int enabled);

/* ---- timed CPU baseline: nstreams independent streams, OpenMP over streams.
 * raw=1: iq is [nstreams][n*8] at 2.016 MS/s (stage 0 included; raw=3: its third-order form), else
 * [nstreams][n] at 252 kS/s.  The whole sample is processed `repeat` times
 * (fresh state each time) so a bounded sample can fill a timing window.
 * Returns seconds; fills bits_out[nstreams][cap]
 * with NUL-terminated strings of chain 0 when non-NULL.                       */
double nvxo_bench(const int16_t *iq, size_t nstreams, size_t n, int raw, int chain_mask,
                  int nthreads, int repeat, char *bits_out, size_t cap);
/* wideband CPU baseline: raw is [nwide][n_out*8] at 2.016 MS/s; each stream is channelised
 * and its 8 sub-bands run through 252 kS/s pipes with both chains.  bits_out (optional):
 * [nwide*16][cap] NUL-terminated, index (w*8 + k)*2 + chain.                             */
double nvxo_bench_wide(const int16_t *raw, size_t nwide, size_t n_out, int nthreads, int repeat, char *bits_out, size_t cap);
int    nvxo_max_threads(void);


/* What a benchmark loop over a resident batch computes: the same samples pushed `loops` times into ONE pipe per stream,
 * state carried from repeat to repeat.  bits_out: [nstreams][chains][cap] (chains = 2 for chain_mask 3, else 1);
 * nvxo_replay_wide: [nwide * 8][2][cap], channeliser history carried too.  Return the seconds spent.                 */
double nvxo_replay(const int16_t *iq, size_t nstreams, size_t n, int raw, int chain_mask, int nthreads, int loops, char *bits_out, size_t cap);
double nvxo_replay_wide(const int16_t *raw, size_t nwide, size_t n_out, int nthreads, int loops, char *bits_out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
