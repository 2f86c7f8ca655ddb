/* TEST INFRASTRUCTURE ONLY -- see nvx_oracle_tables.h for the status header.
 *
 * CPU restatement of the reference path.  Own code, block based; every
 * function cites the reference lines whose arithmetic it reproduces.  The
 * arithmetic contract that makes results bit-identical to the reference's
 * x86-64 build (-O2, SSE2, no FMA):
 *   - FIR sums: acc = 0.0; for i = 0..T-1 in order: acc = acc + h[i] * x[newest - i]
 *     (one rounding for the product, one for the sum), I and Q independently;
 *   - zero-initialised history (the reference's statics / constructors);
 *   - libm calls exactly where the reference makes them (atan2, cos, sin,
 *     cosf, sinf) -- same glibc, same results.
 * Built with -ffp-contract=off so no compiler may fuse the mul/add pairs.
 */
#define _GNU_SOURCE
#include "nvx_oracle.h"
#include "nvx_oracle_tables.h"

#include <math.h>
#include <regex.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double v2d __attribute__((vector_size(16)));   /* {I, Q}: lanes never interact */

/* ========================================================================== */
/* stage 0 (build-owned, no reference counterpart)                            */
/* ========================================================================== */
void nvxo_stage0(const int16_t *raw, size_t n_out, int16_t *out)
{
    for (size_t m = 0; m < n_out; m++) {
        int32_t si = 0, sq = 0;
        for (int j = 0; j < NVXO_D0; j++) {
            si += raw[2 * (NVXO_D0 * m + j)];
            sq += raw[2 * (NVXO_D0 * m + j) + 1];
        }
        /* arithmetic shift of a negative int32 is floor division by 8 on every
         * compiler this builds with; written with an explicit floor to be safe */
        si += 4; sq += 4;
        out[2 * m]     = (int16_t)((si >= 0) ? (si >> 3) : -((-si + 7) >> 3));
        out[2 * m + 1] = (int16_t)((sq >= 0) ? (sq >> 3) : -((-sq + 7) >> 3));
    }
}

/* Third-order form: three cascaded 8-sample boxcars, decimated by 8 (a CIC^3 written as the 22-tap FIR it equals).
 * hist14: the 14 raw samples in front of raw[0] (oldest first; NULL = silence), replaced by the last 14 of this call. */
static const int32_t NVXO_CIC3_W[22] = { 1, 3, 6, 10, 15, 21, 28, 36, 42, 46, 48, 48, 46, 42, 36, 28, 21, 15, 10, 6, 3, 1 };

void nvxo_stage0_cic3(const int16_t *raw, size_t n_out, int16_t *hist14, int16_t *out)
{
    for (size_t m = 0; m < n_out; m++) {
        int32_t si = 256, sq = 256;                              /* + 256: round half up before >> 9 */
        for (int j = 0; j < 22; j++) {
            const long n = (long)(NVXO_D0 * m) + 7 - j;          /* newest sample first, as the definition reads */
            int32_t xi = 0, xq = 0;
            if (n >= 0) { xi = raw[2 * n]; xq = raw[2 * n + 1]; }
            else if (hist14) { xi = hist14[2 * (14 + n)]; xq = hist14[2 * (14 + n) + 1]; }
            si += NVXO_CIC3_W[j] * xi;
            sq += NVXO_CIC3_W[j] * xq;
        }
        out[2 * m]     = (int16_t)((si >= 0) ? (si >> 9) : -((-si + 511) >> 9));      /* floor */
        out[2 * m + 1] = (int16_t)((sq >= 0) ? (sq >> 9) : -((-sq + 511) >> 9));
    }
    if (hist14) {
        const size_t n_raw = n_out * NVXO_D0;
        int16_t keep[28];
        for (int i = 0; i < 14; i++) {
            const long n = (long)n_raw - 14 + i;
            if (n >= 0) { keep[2 * i] = raw[2 * n]; keep[2 * i + 1] = raw[2 * n + 1]; }
            else { keep[2 * i] = hist14[2 * (14 + n)]; keep[2 * i + 1] = hist14[2 * (14 + n) + 1]; }
        }
        memcpy(hist14, keep, sizeof keep);
    }
}

/* ========================================================================== */
/* wideband channeliser (build-owned, no reference counterpart)               */
/* ========================================================================== */
#define NVXO_PFB_TABLE static const
#include "nvx_oracle_pfb_taps.h"

static inline int32_t mulc45(int32_t t) { return (int32_t)(((int64_t)t * NVXO_PFB_C45) >> 15); }
static inline int16_t clamp16(int32_t v) { return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }

void nvxo_channelise(const int16_t *raw, size_t n_out, const int16_t *hist40, int16_t *out)
{
    for (size_t m = 0; m < n_out; m++) {
        int32_t ur[8], ui[8];
        for (int p = 0; p < 8; p++) {
            int32_t re = 0, im = 0;
            for (int j = p; j < NVXO_PFB_T; j += 8) {
                const long n = (long)(8 * m) - 40 + j;              /* sample index of window slot j */
                int32_t xi, xq;
                if (n >= 0) { xi = raw[2 * n]; xq = raw[2 * n + 1]; }
                else if (hist40) { xi = hist40[2 * (40 + n)]; xq = hist40[2 * (40 + n) + 1]; }
                else { xi = 0; xq = 0; }
                re += NVXO_PFB_H[47 - j] * xi;
                im += NVXO_PFB_H[47 - j] * xq;
            }
            ur[p] = (re + 16) >> 5;
            ui[p] = (im + 16) >> 5;
        }
        /* radix-2 decimation-in-time, forward transform e^{-j 2 pi k p / 8} */
        int32_t ar[8], ai[8], br[8], bi[8], yr[8], yi[8];
        ar[0] = ur[0] + ur[4]; ai[0] = ui[0] + ui[4];  ar[1] = ur[0] - ur[4]; ai[1] = ui[0] - ui[4];
        ar[2] = ur[2] + ur[6]; ai[2] = ui[2] + ui[6];  ar[3] = ur[2] - ur[6]; ai[3] = ui[2] - ui[6];
        ar[4] = ur[1] + ur[5]; ai[4] = ui[1] + ui[5];  ar[5] = ur[1] - ur[5]; ai[5] = ui[1] - ui[5];
        ar[6] = ur[3] + ur[7]; ai[6] = ui[3] + ui[7];  ar[7] = ur[3] - ur[7]; ai[7] = ui[3] - ui[7];
        /* (-j)(x + jy) = y - jx */
        br[0] = ar[0] + ar[2]; bi[0] = ai[0] + ai[2];  br[2] = ar[0] - ar[2]; bi[2] = ai[0] - ai[2];
        br[1] = ar[1] + ai[3]; bi[1] = ai[1] - ar[3];  br[3] = ar[1] - ai[3]; bi[3] = ai[1] + ar[3];
        br[4] = ar[4] + ar[6]; bi[4] = ai[4] + ai[6];  br[6] = ar[4] - ar[6]; bi[6] = ai[4] - ai[6];
        br[5] = ar[5] + ai[7]; bi[5] = ai[5] - ar[7];  br[7] = ar[5] - ai[7]; bi[7] = ai[5] + ar[7];
        /* W1 = (1 - j)/sqrt2: W1 (x+jy) = ((x+y) + j(y-x))/sqrt2 ; W3 = (-1 - j)/sqrt2: ((y-x) + j(-x-y))/sqrt2 */
        const int32_t w1r = mulc45(br[5] + bi[5]), w1i = mulc45(bi[5] - br[5]);
        const int32_t w3r = mulc45(bi[7] - br[7]), w3i = mulc45(-br[7] - bi[7]);
        yr[0] = br[0] + br[4]; yi[0] = bi[0] + bi[4];  yr[4] = br[0] - br[4]; yi[4] = bi[0] - bi[4];
        yr[1] = br[1] + w1r;   yi[1] = bi[1] + w1i;    yr[5] = br[1] - w1r;   yi[5] = bi[1] - w1i;
        yr[2] = br[2] + bi[6]; yi[2] = bi[2] - br[6];  yr[6] = br[2] - bi[6]; yi[6] = bi[2] + br[6];
        yr[3] = br[3] + w3r;   yi[3] = bi[3] + w3i;    yr[7] = br[3] - w3r;   yi[7] = bi[3] - w3i;
        for (int k = 0; k < 8; k++) {
            out[2 * (k * n_out + m)]     = clamp16((yr[k] + 4096) >> 13);
            out[2 * (k * n_out + m) + 1] = clamp16((yi[k] + 4096) >> 13);
        }
    }
}

/* ========================================================================== */
/* decimating FIR, direct form, reference tap order                           */
/* ========================================================================== */
typedef struct {
    int T, D;
    const double *h;
    int phase;          /* samples since the last output, 0..D-1               */
    v2d *work;          /* [T-1 history | block]                               */
    size_t cap;         /* capacity of the block part                          */
} fir_stage;

static void stage_init(fir_stage *s, int T, int D, const double *h)
{
    s->T = T; s->D = D; s->h = h; s->phase = 0; s->cap = 0;
    s->work = calloc((size_t)(T - 1), sizeof(v2d));     /* zero history */
}
static void stage_free(fir_stage *s) { free(s->work); s->work = NULL; }

static v2d *stage_block(fir_stage *s, size_t n)      /* where the caller writes n new samples */
{
    if (n > s->cap) {
        s->work = realloc(s->work, ((size_t)(s->T - 1) + n) * sizeof(v2d));
        s->cap = n;
    }
    return s->work + (s->T - 1);
}

/* fir1cpp.C:94-131, fir2cpp.C:143-167, fir3cpp.C:33-56: an output is produced
 * by the sample that makes the decimation counter reach D; it is the dot
 * product of h[0..T-1] with the newest T samples, newest first.              */
static size_t stage_run(fir_stage *s, size_t n, v2d *out)
{
    const int T = s->T, D = s->D;
    const double *h = s->h;
    v2d *x = s->work + (T - 1);
    size_t nout = 0;
    size_t j = (size_t)(D - 1 - s->phase);                /* first firing sample */
    for (; j < n; j += (size_t)D) {
        const v2d *p = x + j;
        v2d acc = { 0.0, 0.0 };
        for (int i = 0; i < T; i++) {
            v2d c = { h[i], h[i] };
            acc = acc + c * p[-i];
        }
        out[nout++] = acc;
    }
    s->phase = (int)((s->phase + n) % (size_t)D);
    /* keep the newest T-1 samples as history */
    memmove(s->work, s->work + n, (size_t)(T - 1) * sizeof(v2d));
    return nout;
}

/* ---- whole-array forms ---------------------------------------------------- */
size_t nvxo_fir1(const int16_t *iq, size_t n, double *y1)
{
    fir_stage s; stage_init(&s, NVXO_T1, NVXO_D1, NVXO_H1);
    v2d *x = stage_block(&s, n);
    for (size_t k = 0; k < n; k++) { v2d v = { (double)iq[2 * k], (double)iq[2 * k + 1] }; x[k] = v; }
    size_t r = stage_run(&s, n, (v2d *)y1);
    stage_free(&s);
    return r;
}
static size_t fir_dd(const double *in, size_t n, double *out, int T, int D, const double *h)
{
    fir_stage s; stage_init(&s, T, D, h);
    v2d *x = stage_block(&s, n);
    memcpy(x, in, n * sizeof(v2d));
    size_t r = stage_run(&s, n, (v2d *)out);
    stage_free(&s);
    return r;
}
size_t nvxo_fir2(const double *u, size_t n, double *y2) { return fir_dd(u, n, y2, NVXO_T2, NVXO_D2, NVXO_H2); }
size_t nvxo_fir3(const double *y2, size_t n, double *y3) { return fir_dd(y2, n, y3, NVXO_T3, NVXO_D3, NVXO_H3); }

/* ========================================================================== */
/* mixer (fir2cpp.C:104-107, 112-128)                                         */
/* ========================================================================== */
void nvxo_mixer_table(double cr[NVXO_MIX_N], double ci[NVXO_MIX_N])
{
    for (int i = 0; i < NVXO_MIX_N; i++) {
        cr[i] = cos((2 * M_PI * i * 14000) / 63000);      /* fir2cpp.C:105 */
        ci[i] = -sin((2 * M_PI * i * 14000) / 63000);     /* fir2cpp.C:106 */
    }
}

static inline v2d mix_one(v2d s, double cr, double ci, int chain)
{
    double I = s[0], Q = s[1];
    v2d r;
    if (chain == 0) {           /* 518: fir2cpp.C:116-117 */
        r[0] = I * cr - Q * ci;
        r[1] = I * ci + Q * cr;
    } else {                    /* 490: fir2cpp.C:122-123 */
        r[0] = I * cr + Q * ci;
        r[1] = -I * ci + Q * cr;
    }
    return r;
}

void nvxo_mix(const double *y1, size_t n1, int chain, double *u)
{
    double cr[NVXO_MIX_N], ci[NVXO_MIX_N];
    nvxo_mixer_table(cr, ci);
    const v2d *in = (const v2d *)y1; v2d *out = (v2d *)u;
    for (size_t k = 0; k < n1; k++) out[k] = mix_one(in[k], cr[k % NVXO_MIX_N], ci[k % NVXO_MIX_N], chain);
}

/* ========================================================================== */
/* decoder (decoder.C)                                                        */
/* ========================================================================== */
enum { ST_INIT = 0, ST_SYNCED_WAIT = 1, ST_BIT_START = 2, ST_RECEIVING = 3 };   /* decoder.h:16-19 */

void nvxo_bitfilter_table(float fR[5], float fI[5])
{
    for (int i = 0; i < 5; i++) {                          /* decoder.C:23-28 */
        float angle = (i * 2 * 3.1415 * 85) / 900;         /* double expression -> float */
        fR[i] = cosf(angle);                               /* C++ float overloads => cosf/sinf */
        fI[i] = sinf(angle);
    }
}

void nvxo_dec_init(nvxo_dec *d)                            /* decoder.C:6-39 */
{
    memset(d, 0, sizeof *d);
    d->status = ST_INIT;
    nvxo_bitfilter_table(d->fR, d->fI);
    d->prev_offset = -1;
    d->last_sync = -1;
}

static void bd_in_bit_sync(nvxo_dec *d, int offset)        /* decoder.C:62-70 */
{
    if (d->status == ST_INIT) { d->status = ST_SYNCED_WAIT; d->bit_sync_offset = offset; }
    d->next_bit_sync_offset = offset;
    d->last_sync = offset;
}

static void bs_sample(nvxo_dec *d, double ds)              /* decoder.C:142-255 */
{
    d->dab[d->dab_index] = ds;
    d->dab_index++;
    if (d->dab_index == NVXO_SPB) { d->dab_index = 0; d->dab_primed = 1; }

    if (d->dab_primed) {
        double temp = 0.0;
        int j = d->dab_index;
        for (int i = 0; i < NVXO_SPB; i++) {
            temp += NVXO_CORR_MASK[i] * d->dab[j];         /* int * double, in i order */
            j++; j %= NVXO_SPB;
        }
        d->cb[d->cb_index] = fabs(temp);
        d->cb_index++;
        if (d->cb_index == NVXO_CORR_N) { d->cb_index = 0; d->cb_primed = 1; }
    }

    if (d->cb_primed) {
        double temp = 0.0;
        for (int i = d->csa_index; i < NVXO_CORR_N; i += NVXO_SPB) temp += d->cb[i];
        d->csa[d->csa_index] = temp;
        d->csa_index++;
        if (d->csa_index == NVXO_SPB) { d->csa_index = 0; d->csa_primed = 1; }
    }

    if (d->csa_primed) {
        if ((d->bs_seq_nbr % NVXO_SPB) == 0) {
            double temp_max = -1.0;
            int max_index = 0;   /* reference leaves it uninitialised; sums are >= 0 > -1 so it is always set */
            for (int i = 0; i < NVXO_SPB; i++)
                if (d->csa[i] > temp_max) { temp_max = d->csa[i]; max_index = i; }
            if (!(d->prev_offset == -1 || max_index == d->prev_offset)) {
                if (max_index > d->prev_offset) {
                    if (max_index - d->prev_offset > 4) max_index = (d->prev_offset - 1 + NVXO_SPB) % NVXO_SPB;
                    else                                max_index = (d->prev_offset + 1) % NVXO_SPB;
                } else {
                    if (d->prev_offset - max_index > 4) max_index = (d->prev_offset + 1) % NVXO_SPB;
                    else                                max_index = (d->prev_offset - 1 + NVXO_SPB) % NVXO_SPB;
                }
            }
            d->prev_offset = max_index;
            bd_in_bit_sync(d, (max_index + 5) % NVXO_SPB);
        }
        d->bs_seq_nbr = (d->bs_seq_nbr + 1) % NVXO_SPB;
    }
}

static int bd_sample(nvxo_dec *d, double sampleR, double sampleI)   /* decoder.C:73-137 */
{
    /* decoder.C:75 `bd_seq_nbr ++;` on the `int` of decoder.h:60, for ever: after 2^31 samples at 900 S/s -- 27.6 days -- it
     * passes INT_MAX.  That is undefined behaviour in the reference; what its x86-64 build DOES is wrap to INT_MIN (an
     * `add` on a memory operand), and that is stated here without the undefined behaviour.  From then on the remainders
     * of decoder.C:85 are -8..0, which equal a sync offset (0..8) only when both are 0: the reference's decoder falls
     * nearly silent for the next 27.6 days.  The PRODUCT does not copy this ("Deviations"): its bit phase
     * is a function of the sample's position.  tests/test_deviations.py pins this line against the compiled reference. */
    d->bd_seq_nbr = (int)((unsigned)d->bd_seq_nbr + 1u);
    if (d->status == ST_INIT) return 0;
    if (d->status == ST_SYNCED_WAIT) {
        if ((d->bd_seq_nbr % NVXO_SPB) == d->bit_sync_offset) { d->status = ST_BIT_START; d->burn_count = 0; }
    }
    if (d->status == ST_BIT_START) {
        if (d->burn_count == 2) {
            d->status = ST_RECEIVING;
            d->samplecount = 0;
            d->BR = 0.0f; d->BI = 0.0f; d->YR = 0.0f; d->YI = 0.0f;
            return 0;
        }
        d->burn_count++;
        return 0;
    }
    if (d->status == ST_RECEIVING) {
        const float *bit_filterR = d->fR, *bit_filterI = d->fI;
        int samplecount = d->samplecount;
        /* decoder.C:115-118, expression text kept: the cast binds to sampleR
         * only, so the first product is float*float and the second is
         * double*float; the compound assignment rounds back to float.       */
        d->YR += (float) sampleR*bit_filterR[samplecount]-sampleI*bit_filterI[samplecount];
        d->YI += (float) sampleR*bit_filterI[samplecount]+sampleI*bit_filterR[samplecount];
        d->BR += (float) sampleR*bit_filterR[samplecount]+sampleI*bit_filterI[samplecount];
        d->BI += (float) -sampleR*bit_filterI[samplecount]+sampleI*bit_filterR[samplecount];
        d->samplecount++;
        if (d->samplecount == 5) {
            float Brot = d->BR * d->BR + d->BI * d->BI;
            float Yrot = d->YR * d->YR + d->YI * d->YI;
            d->status = ST_SYNCED_WAIT;
            d->bit_sync_offset = d->next_bit_sync_offset;
            return (Brot > Yrot) ? 'B' : 'Y';
        }
    }
    return 0;
}

int nvxo_dec_push(nvxo_dec *d, double sampleI, double sampleQ)      /* decoder.C:42-59 */
{
    double prodReal = sampleI * d->prevI + sampleQ * d->prevQ;
    double prodImg  = sampleQ * d->prevI - sampleI * d->prevQ;
    double result = atan2(prodImg, prodReal);
    d->prevI = sampleI; d->prevQ = sampleQ;
    d->last_dphi = result;
    d->last_sync = -1;
    bs_sample(d, result);
    return bd_sample(d, sampleI, sampleQ);
}

size_t nvxo_decode_with(const double *y3, size_t n3, char *bits_out, nvxo_atan2_fn fn, size_t *dphi_mismatch)
{
    nvxo_dec d; nvxo_dec_init(&d);
    size_t nb = 0, mism = 0;
    for (size_t k = 0; k < n3; k++) {
        const double sampleI = y3[2 * k], sampleQ = y3[2 * k + 1];
        const double prodReal = sampleI * d.prevI + sampleQ * d.prevQ;       /* decoder.C:48-49 */
        const double prodImg  = sampleQ * d.prevI - sampleI * d.prevQ;
        const double ref = atan2(prodImg, prodReal);
        const double result = fn ? fn(prodImg, prodReal) : ref;
        if (memcmp(&ref, &result, 8)) mism++;
        d.prevI = sampleI; d.prevQ = sampleQ;
        bs_sample(&d, result);
        int b = bd_sample(&d, sampleI, sampleQ);
        if (b) bits_out[nb++] = (char)b;
    }
    if (dphi_mismatch) *dphi_mismatch = mism;
    return nb;
}

/* test probe (tests/test_deviations.py): nvxo_decode with the decoder's bd_seq_nbr set to `value` in front of sample
 * `at` -- the state 27.6 days of running reach by themselves (see bd_sample) */
size_t nvxo_decode_inject(const double *y3, size_t n3, char *bits_out, size_t at, int value, int *bd_seq_nbr_at)
{
    nvxo_dec d; nvxo_dec_init(&d);
    size_t nb = 0;
    for (size_t k = 0; k < n3; k++) {
        if (k == at) { if (bd_seq_nbr_at) *bd_seq_nbr_at = d.bd_seq_nbr; d.bd_seq_nbr = value; }
        int b = nvxo_dec_push(&d, y3[2 * k], y3[2 * k + 1]);
        if (b) bits_out[nb++] = (char)b;
    }
    return nb;
}

size_t nvxo_decode(const double *y3, size_t n3, char *bits_out, double *dphi_out)
{
    nvxo_dec d; nvxo_dec_init(&d);
    size_t nb = 0;
    for (size_t k = 0; k < n3; k++) {
        int b = nvxo_dec_push(&d, y3[2 * k], y3[2 * k + 1]);
        if (dphi_out) dphi_out[k] = d.last_dphi;
        if (b) bits_out[nb++] = (char)b;
    }
    return nb;
}

/* ========================================================================== */
/* SITOR-B character layer (nav_b_sm.C)                                       */
/* ========================================================================== */
#define E_BUFFER_SIZE 20                                   /* nav_b_sm.h:49 */
#define ERROR_THRESHOLD 12                                 /* nav_b_sm.h:50 */
#define PHASE_DIS_TIMER (100 * 11)                         /* nav_b_sm.h:52 */
enum { S_BYTE_WAIT = 1, S_BYTE_RECEIVED_DX = 2, S_BYTE_RECEIVED_RX = 3 };
enum { MODE_LETTERS = 3, MODE_FIGURES = 4 };

/* nav_b_sm.h:60-83 -- code -> character; '_' marks an invalid code; lower
 * case letters are control pseudo-characters (l/f shifts, n LF, r CR, p/q
 * phasing).  Quirks kept on purpose: 0x5C -> ' ', 0x19 -> '-' in both.      */
static const char LTRS[129] =
    "_______p___J_WA_" "___F_YS__-D_Z___" "___C_PI__GR_L___" "_MN_H___O_______"
    "___K_QU__fE_q___" "_Xl_____B___ ___" "_V _n___T_______" "r_______________";
static const char FIGS[129] =
    "_______p___b_2-_" "___*_6'__-%_+ __" "___:_08__*4_)___" "_.,_*___9_______"
    "___(_17__f3_q___" "_/l_____?___ ___" "_= _n___5_______" "r_______________";

struct nvxo_sm {
    char  dx_buffer[3];
    char  error_buffer[E_BUFFER_SIZE];
    char  temp_byte;
    char  line_buffer[5000], message_buffer[5000], message_bbbb[10];
    int   status, byte_mode, bits_received, dx_buf_ptr, dx_buf_filled, byte_status;
    int   error_count, error_buffer_ptr, error_buffer_filled;
    int   end_of_emission_counter, previous_DX_was_alpha, phase_det_disable_timer;
    int   freq, byte_reception_enabled, message_reception_ongoing;
    nvxo_msg_cb cb; void *user;
    char *trace; size_t trace_len, trace_cap;
};

static void tr(nvxo_sm *s, const char *fmt, ...)
{
    char tmp[5200];
    va_list ap; va_start(ap, fmt);
    int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (n < 0) return;
    if ((size_t)n >= sizeof tmp) n = sizeof tmp - 1;
    if (s->trace_len + (size_t)n + 1 > s->trace_cap) {
        s->trace_cap = (s->trace_cap + (size_t)n + 1) * 2;
        s->trace = realloc(s->trace, s->trace_cap);
    }
    memcpy(s->trace + s->trace_len, tmp, (size_t)n);
    s->trace_len += (size_t)n;
    s->trace[s->trace_len] = 0;
}

/* bounded strcat: the reference's 5000-byte buffers would overflow (UB) on a
 * line-feed-free garbage stream; the oracle truncates instead.               */
static void cat5000(char *dst, const char *src)
{
    size_t l = strlen(dst), m = strlen(src);
    if (l + m > 4999) m = 4999 - l;
    memcpy(dst + l, src, m); dst[l + m] = 0;
}

static void sm_init(nvxo_sm *s)                            /* nav_b_sm.C:16-42 */
{
    s->status = 0; s->byte_status = S_BYTE_WAIT; s->byte_mode = MODE_LETTERS;
    s->bits_received = 0; s->dx_buf_ptr = 0; s->dx_buf_filled = 0;
    s->error_count = 0; s->error_buffer_ptr = 0; s->error_buffer_filled = 0;
    s->end_of_emission_counter = 0; s->previous_DX_was_alpha = 0;
    s->line_buffer[0] = 0; s->message_buffer[0] = 0; s->message_bbbb[0] = 0;
    s->phase_det_disable_timer = 0;
    s->byte_reception_enabled = 0; s->message_reception_ongoing = 0;
}

static void sm_abort(nvxo_sm *s)                           /* nav_b_sm.C:44-52 */
{
    tr(s, "message abort\n");
    if (s->message_reception_ongoing && s->cb) s->cb(s->user, s->message_bbbb, s->message_buffer, s->freq);
    sm_init(s);
}

static void sm_line_out(nvxo_sm *s)                        /* nav_b_sm.C:56-97 */
{
    regex_t re; regmatch_t pm[4];
    if (s->message_reception_ongoing) {
        cat5000(s->message_buffer, s->line_buffer);
        cat5000(s->message_buffer, "\n");
        tr(s, "line added: %s\n", s->line_buffer);
    }
    regcomp(&re, "(CZC|Z.ZC|ZC.C|ZCZ.) +([A-Z][A-Z])([0-9][0-9])", REG_EXTENDED);
    int som = regexec(&re, s->line_buffer, 4, pm, 0) == 0;
    regfree(&re);
    if (som) {
        strcpy(s->message_buffer, s->line_buffer);
        cat5000(s->message_buffer, "\n");
        strncat(s->message_bbbb, s->line_buffer + pm[2].rm_so, (size_t)(pm[2].rm_eo - pm[2].rm_so));
        strncat(s->message_bbbb, s->line_buffer + pm[3].rm_so, (size_t)(pm[3].rm_eo - pm[3].rm_so));
        s->message_bbbb[4] = 0;
        tr(s, "============START OF MESSAGE============ \n");
        s->message_reception_ongoing = 1;
    } else {
        regcomp(&re, "NNN.*|N.NN.*|NN.N.*", REG_EXTENDED);
        int eom = regexec(&re, s->line_buffer, 1, pm, 0) == 0;
        regfree(&re);
        if (eom) {
            if (s->message_reception_ongoing && s->cb) s->cb(s->user, s->message_bbbb, s->message_buffer, s->freq);
            s->message_buffer[0] = 0; s->message_bbbb[0] = 0;
            tr(s, "============ END OF MESSAGE ============\n");
            s->message_reception_ongoing = 0;
        }
    }
    s->line_buffer[0] = 0;
}

static void sm_byte_out(nvxo_sm *s, unsigned char b)       /* nav_b_sm.C:100-145 */
{
    if (b == 0) { tr(s, "*"); cat5000(s->line_buffer, "*"); }
    else if (LTRS[b] == 'l') s->byte_mode = MODE_LETTERS;
    else if (LTRS[b] == 'f') s->byte_mode = MODE_FIGURES;
    else if (LTRS[b] == 'n') sm_line_out(s);
    else if (LTRS[b] == 'r') { }
    else if (LTRS[b] == 'p') { }
    else if (LTRS[b] == 'q') { }
    else {
        char c[2] = { (s->byte_mode == MODE_LETTERS) ? LTRS[b] : FIGS[b], 0 };
        tr(s, ".");
        cat5000(s->line_buffer, c);
        tr(s, ";");
    }
}

static void sm_rxdx_byte(nvxo_sm *s, unsigned char b)      /* nav_b_sm.C:150-262 */
{
    switch (s->byte_status) {
    case S_BYTE_WAIT:
        if (b == 0x07) s->byte_status = S_BYTE_RECEIVED_RX;
        if (b == 0x4c) s->byte_status = S_BYTE_RECEIVED_DX;
        break;
    case S_BYTE_RECEIVED_RX:
        s->dx_buffer[s->dx_buf_ptr] = (char)b;
        s->dx_buf_ptr++;
        if (s->dx_buf_ptr == 3) { s->dx_buf_ptr = 0; s->dx_buf_filled = 1; }
        if (b == 0x07) {
            tr(s, "\n alpha received in DX position\n");
            if (s->previous_DX_was_alpha) {
                s->end_of_emission_counter++;
                if (s->end_of_emission_counter == 2) {
                    tr(s, "\nend of emission detected\n");
                    tr(s, "\nstopping reception\n");
                    sm_abort(s);
                    break;
                }
            }
            s->previous_DX_was_alpha = 1;
        } else {
            s->previous_DX_was_alpha = 0;
        }
        s->byte_status = S_BYTE_RECEIVED_DX;
        break;
    case S_BYTE_RECEIVED_DX:
        if (s->dx_buf_filled) {
            unsigned char dx = (unsigned char)s->dx_buffer[s->dx_buf_ptr];
            if (LTRS[b] != '_')       sm_byte_out(s, b);
            else if (LTRS[dx] != '_') sm_byte_out(s, dx);
            else                      sm_byte_out(s, 0);
        }
        s->byte_status = S_BYTE_RECEIVED_RX;
        break;
    }
    /* sliding error window, nav_b_sm.C:235-261 */
    if (s->error_buffer_filled && s->error_buffer[s->error_buffer_ptr] == '_') s->error_count--;
    s->error_buffer[s->error_buffer_ptr] = LTRS[b];
    if (s->error_buffer[s->error_buffer_ptr] == '_') s->error_count++;
    s->error_buffer_ptr++;
    if (s->error_buffer_ptr == E_BUFFER_SIZE) { s->error_buffer_ptr = 0; s->error_buffer_filled = 1; }
    if (s->error_count > ERROR_THRESHOLD) {
        sm_byte_out(s, 0);
        tr(s, "\n error th exceeded \n");
        sm_abort(s);
    }
}

/* nav_b_sm.C:301-631: 30 states, one per matched bit of this pattern; any
 * mismatch returns to state 0 WITHOUT re-examining the bit, except state 6
 * (six B's seen) which stays put on a further 'B' (nav_b_sm.C:363-372).     */
static const char PHASING[31] = "BBBBBBYYYYBBYYBBBBBBYYYYBBYYBB";

void nvxo_sm_bit(nvxo_sm *s, char bit)                     /* nav_b_sm.C:266-634 */
{
    if (s->byte_reception_enabled) {
        s->temp_byte = (char)(s->temp_byte << 1);
        if (bit == 'Y') s->temp_byte |= 0x01;
        s->bits_received += 1;
        if (s->bits_received == 7) {
            sm_rxdx_byte(s, (unsigned char)s->temp_byte);
            s->bits_received = 0;
            s->temp_byte = 0;
        }
    }
    if (s->phase_det_disable_timer != 0) {
        s->phase_det_disable_timer--;
        if (s->phase_det_disable_timer == 0) tr(s, "phase det disable timer expired\n");
    } else {
        int st = s->status;
        if (st == 29) {
            if (bit == 'B') {
                s->byte_reception_enabled = 1;
                s->bits_received = 0;
                s->temp_byte = 0;
                tr(s, "phasing detected\n");
                s->phase_det_disable_timer = PHASE_DIS_TIMER;
            }
            s->status = 0;
        } else if (bit == PHASING[st]) {
            s->status = st + 1;
        } else if (st == 6) {
            s->status = 6;
        } else {
            s->status = 0;
        }
    }
}

nvxo_sm *nvxo_sm_new(int freq, nvxo_msg_cb cb, void *user)
{
    nvxo_sm *s = calloc(1, sizeof *s);
    s->freq = freq; s->cb = cb; s->user = user;
    s->trace_cap = 256; s->trace = malloc(s->trace_cap); s->trace[0] = 0;
    sm_init(s);
    return s;
}
void nvxo_sm_free(nvxo_sm *s) { if (s) { free(s->trace); free(s); } }
const char *nvxo_sm_trace(nvxo_sm *s, size_t *len) { if (len) *len = s->trace_len; return s->trace; }

/* ========================================================================== */
/* streaming pipeline (nav_sched.C:10-22 object graph, per stream)            */
/* ========================================================================== */
typedef struct {
    fir_stage f2, f3;
    nvxo_dec dec;
    nvxo_sm *sm;
    char *bits; size_t nbits, capbits;
    double *tap; size_t tap_cap, *tap_count;
} pipe_chain;

struct nvxo_pipe {
    int chain_mask, charlayer;
    fir_stage f1;
    unsigned mix_idx;                                    /* fir2cpp.C:7 freq_shift_idx */
    double cr[NVXO_MIX_N], ci[NVXO_MIX_N];
    pipe_chain ch[2];
    v2d *y1, *y2, *y3; size_t cap1;
    int16_t *s0; size_t cap0;
    int stage0_order; int16_t hist0[28];      /* raw-rate front end: 1 = integrate-and-dump, 3 = third-order form + its history */
};

nvxo_pipe *nvxo_pipe_new(int chain_mask, int freq0, int freq1, nvxo_msg_cb cb, void *user)
{
    nvxo_pipe *p = calloc(1, sizeof *p);
    p->chain_mask = chain_mask; p->charlayer = 1;
    stage_init(&p->f1, NVXO_T1, NVXO_D1, NVXO_H1);
    nvxo_mixer_table(p->cr, p->ci);
    for (int c = 0; c < 2; c++) {
        stage_init(&p->ch[c].f2, NVXO_T2, NVXO_D2, NVXO_H2);
        stage_init(&p->ch[c].f3, NVXO_T3, NVXO_D3, NVXO_H3);
        nvxo_dec_init(&p->ch[c].dec);
        p->ch[c].sm = nvxo_sm_new(c == 0 ? freq0 : freq1, cb, user);
        p->ch[c].capbits = 1024; p->ch[c].bits = malloc(p->ch[c].capbits); p->ch[c].bits[0] = 0;
    }
    return p;
}

void nvxo_pipe_free(nvxo_pipe *p)
{
    if (!p) return;
    stage_free(&p->f1);
    for (int c = 0; c < 2; c++) {
        stage_free(&p->ch[c].f2); stage_free(&p->ch[c].f3);
        nvxo_sm_free(p->ch[c].sm); free(p->ch[c].bits);
    }
    free(p->y1); free(p->y2); free(p->y3); free(p->s0); free(p);
}

void nvxo_pipe_set_charlayer(nvxo_pipe *p, int enabled) { p->charlayer = enabled; }

/* What calling the reference's init functions AGAIN, in mid-stream, does to its statics (test probe: the product does
 * not copy this, DESIGN.md "Deviations"; pinned against the compiled reference by tests/test_deviations.py):
 *   which & 1  init_fir_filter1() (fir1cpp.C:65-77): FIR1's ring zeroed, its write pointer and decimation counter 0 --
 *              nothing else;
 *   which & 2  init_fir2_wrapper() -> init_fir_filter2() (nav_sched.C:19-22, fir2cpp.C:90-110): the 518 chain's FIR2
 *              ring, pointer and counter, and the mixer index BOTH chains share (fir2cpp.C:7) -- never the 490
 *              chain's FIR2 statics (fir2cpp.C:80-83), FIR3, the decoders or the character layers. */
static void stage_clear(fir_stage *s) { s->phase = 0; memset(s->work, 0, (size_t)(s->T - 1) * sizeof(v2d)); }
void nvxo_pipe_reinit(nvxo_pipe *p, int which)
{
    if (which & 1) stage_clear(&p->f1);
    if (which & 2) { stage_clear(&p->ch[0].f2); p->mix_idx = 0; }
}
void nvxo_pipe_tap_y3(nvxo_pipe *p, int c, double *buf, size_t cap_pairs, size_t *count)
{
    p->ch[c].tap = buf; p->ch[c].tap_cap = cap_pairs; p->ch[c].tap_count = count;
    if (count) *count = 0;
}
const char *nvxo_pipe_bits(nvxo_pipe *p, int c, size_t *n) { if (n) *n = p->ch[c].nbits; return p->ch[c].bits; }

static void pipe_block(nvxo_pipe *p, const int16_t *iq, size_t n)
{
    if (n / NVXO_D1 + 2 > p->cap1) {
        p->cap1 = n / NVXO_D1 + 2;
        p->y1 = realloc(p->y1, p->cap1 * sizeof(v2d));
        p->y2 = realloc(p->y2, (p->cap1 / NVXO_D2 + 2) * sizeof(v2d));
        p->y3 = realloc(p->y3, (p->cap1 / (NVXO_D2 * NVXO_D3) + 2) * sizeof(v2d));
    }
    v2d *x = stage_block(&p->f1, n);
    for (size_t k = 0; k < n; k++) {                      /* capt_sched.c:511: (double) of each short */
        v2d v = { (double)iq[2 * k], (double)iq[2 * k + 1] };
        x[k] = v;
    }
    size_t n1 = stage_run(&p->f1, n, p->y1);
    for (int c = 0; c < 2; c++) {
        if (!(p->chain_mask & (1 << c))) continue;
        pipe_chain *ch = &p->ch[c];
        v2d *u = stage_block(&ch->f2, n1);
        unsigned idx = p->mix_idx;
        for (size_t k = 0; k < n1; k++) {
            u[k] = mix_one(p->y1[k], p->cr[idx], p->ci[idx], c);
            idx++; idx %= NVXO_MIX_N;
        }
        size_t n2 = stage_run(&ch->f2, n1, p->y2);
        v2d *w = stage_block(&ch->f3, n2);
        memcpy(w, p->y2, n2 * sizeof(v2d));
        size_t n3 = stage_run(&ch->f3, n2, p->y3);
        for (size_t k = 0; k < n3; k++) {
            if (ch->tap && ch->tap_count && *ch->tap_count < ch->tap_cap) {
                ch->tap[2 * *ch->tap_count] = p->y3[k][0];
                ch->tap[2 * *ch->tap_count + 1] = p->y3[k][1];
                (*ch->tap_count)++;
            }
            int b = nvxo_dec_push(&ch->dec, p->y3[k][0], p->y3[k][1]);
            if (b) {
                if (ch->nbits + 2 > ch->capbits) { ch->capbits *= 2; ch->bits = realloc(ch->bits, ch->capbits); }
                ch->bits[ch->nbits++] = (char)b; ch->bits[ch->nbits] = 0;
                if (p->charlayer) nvxo_sm_bit(ch->sm, (char)b);
            }
        }
    }
    p->mix_idx = (unsigned)((p->mix_idx + n1) % NVXO_MIX_N);
}

void nvxo_pipe_push(nvxo_pipe *p, const int16_t *iq, size_t n)
{
    const size_t BLK = 2520 * 8;
    while (n) {
        size_t m = n < BLK ? n : BLK;
        pipe_block(p, iq, m);
        iq += 2 * m; n -= m;
    }
}

void nvxo_pipe_set_stage0(nvxo_pipe *p, int order) { p->stage0_order = (order == 3) ? 3 : 1; }

void nvxo_pipe_push_raw(nvxo_pipe *p, const int16_t *raw, size_t n_out)
{
    const size_t BLK = 2520 * 8;
    if (p->cap0 < BLK) { p->cap0 = BLK; p->s0 = realloc(p->s0, BLK * 2 * sizeof(int16_t)); }
    while (n_out) {
        size_t m = n_out < BLK ? n_out : BLK;
        if (p->stage0_order == 3) nvxo_stage0_cic3(raw, m, p->hist0, p->s0);
        else nvxo_stage0(raw, m, p->s0);
        pipe_block(p, p->s0, m);
        raw += 2 * m * NVXO_D0; n_out -= m;
    }
}

/* ========================================================================== */
/* timed CPU baseline                                                          */
/* ========================================================================== */
int nvxo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

double nvxo_bench(const int16_t *iq, size_t nstreams, size_t n, int raw, int chain_mask,
                  int nthreads, int repeat, char *bits_out, size_t cap)
{
    struct timespec t0, t1;
    size_t stride = (raw ? n * NVXO_D0 : n) * 2;
    if (nthreads < 1) nthreads = 1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (long job = 0; job < (long)nstreams * repeat; job++) {
        const long s = job % (long)nstreams;            /* the same sample, `repeat` times over */
        nvxo_pipe *p = nvxo_pipe_new(chain_mask, 518, 490, NULL, NULL);
        nvxo_pipe_set_charlayer(p, 0);
        if (raw) { nvxo_pipe_set_stage0(p, raw == 3 ? 3 : 1); nvxo_pipe_push_raw(p, iq + (size_t)s * stride, n); }
        else     nvxo_pipe_push(p, iq + (size_t)s * stride, n);
        if (bits_out && job < (long)nstreams) {
            size_t nb; const char *b = nvxo_pipe_bits(p, (chain_mask & 1) ? 0 : 1, &nb);
            if (nb >= cap) nb = cap - 1;
            memcpy(bits_out + (size_t)s * cap, b, nb); bits_out[(size_t)s * cap + nb] = 0;
        }
        nvxo_pipe_free(p);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

double nvxo_bench_wide(const int16_t *raw, size_t nwide, size_t n_out, int nthreads, int repeat, char *bits_out, size_t cap)
{
    struct timespec t0, t1;
    if (nthreads < 1) nthreads = 1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (long job = 0; job < (long)nwide * repeat; job++) {
        const long w = job % (long)nwide;
        int16_t *sub = malloc(8 * n_out * 2 * sizeof(int16_t));
        nvxo_channelise(raw + (size_t)w * n_out * 16, n_out, NULL, sub);
        for (int k = 0; k < 8; k++) {
            nvxo_pipe *p = nvxo_pipe_new(3, 518, 490, NULL, NULL);
            nvxo_pipe_set_charlayer(p, 0);
            nvxo_pipe_push(p, sub + (size_t)k * n_out * 2, n_out);
            if (bits_out && job < (long)nwide) {
                for (int c = 0; c < 2; c++) {
                    size_t nb; const char *b = nvxo_pipe_bits(p, c, &nb);
                    if (nb >= cap) nb = cap - 1;
                    char *dst = bits_out + ((size_t)(w * 8 + k) * 2 + c) * cap;
                    memcpy(dst, b, nb); dst[nb] = 0;
                }
            }
            nvxo_pipe_free(p);
        }
        free(sub);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* TEST INFRASTRUCTURE, as everything in this file.  What a benchmark loop over a resident batch computes: the same `n`
 * samples of every stream pushed `loops` times into ONE pipe per stream, the carried state (FIR histories: fir1cpp.C:51-60,
 * fir2cpp.C:74-83, fir3cpp.h:90-95; decoder: decoder.h:31-60) running through from repeat to repeat.  bits_out:
 * [nstreams][chains][cap], chains = 2 when chain_mask = 3 (chain 0 then chain 1), else the one chain of the mask.        */
double nvxo_replay(const int16_t *iq, size_t nstreams, size_t n, int raw, int chain_mask, int nthreads, int loops, char *bits_out, size_t cap)
{
    struct timespec t0, t1;
    const size_t stride = (raw ? n * NVXO_D0 : n) * 2;
    const int nch = chain_mask == 3 ? 2 : 1;
    if (nthreads < 1) nthreads = 1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (long s = 0; s < (long)nstreams; s++) {
        nvxo_pipe *p = nvxo_pipe_new(chain_mask, 518, 490, NULL, NULL);
        nvxo_pipe_set_charlayer(p, 0);
        if (raw) nvxo_pipe_set_stage0(p, raw == 3 ? 3 : 1);
        for (int k = 0; k < loops; k++) {
            if (raw) nvxo_pipe_push_raw(p, iq + (size_t)s * stride, n);
            else     nvxo_pipe_push(p, iq + (size_t)s * stride, n);
        }
        for (int c = 0; c < nch && bits_out; c++) {
            size_t nb; const char *b = nvxo_pipe_bits(p, nch == 2 ? c : ((chain_mask & 1) ? 0 : 1), &nb);
            char *dst = bits_out + ((size_t)s * nch + c) * cap;
            if (nb >= cap) nb = cap - 1;
            memcpy(dst, b, nb); dst[nb] = 0;
        }
        nvxo_pipe_free(p);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ... and the same for wideband streams: raw is [nwide][n_out * 8] at 2.016 MS/s, the channeliser's 40-sample history
 * carried from repeat to repeat like the pipes' state; bits_out: [nwide * 8][2][cap].                                   */
double nvxo_replay_wide(const int16_t *raw, size_t nwide, size_t n_out, int nthreads, int loops, char *bits_out, size_t cap)
{
    struct timespec t0, t1;
    if (nthreads < 1) nthreads = 1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (long w = 0; w < (long)nwide; w++) {
        const int16_t *in = raw + (size_t)w * n_out * 16;
        int16_t *sub = malloc(8 * n_out * 2 * sizeof(int16_t));
        int16_t hist[80];
        nvxo_pipe *p[8];
        for (int k = 0; k < 8; k++) { p[k] = nvxo_pipe_new(3, 518, 490, NULL, NULL); nvxo_pipe_set_charlayer(p[k], 0); }
        for (int l = 0; l < loops; l++) {
            nvxo_channelise(in, n_out, l ? hist : NULL, sub);
            memcpy(hist, in + (n_out * 8 - 40) * 2, sizeof hist);          /* the last 40 raw samples, oldest first */
            for (int k = 0; k < 8; k++) nvxo_pipe_push(p[k], sub + (size_t)k * n_out * 2, n_out);
        }
        for (int k = 0; k < 8; k++) {
            for (int c = 0; c < 2 && bits_out; c++) {
                size_t nb; const char *b = nvxo_pipe_bits(p[k], c, &nb);
                char *dst = bits_out + ((size_t)(w * 8 + k) * 2 + c) * cap;
                if (nb >= cap) nb = cap - 1;
                memcpy(dst, b, nb); dst[nb] = 0;
            }
            nvxo_pipe_free(p[k]);
        }
        free(sub);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
