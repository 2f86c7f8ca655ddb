/* nvx_synth_host.c -- host side of the deterministic synthetic source:
 * per-bit-period phase tables (shared with the device launcher) and the host
 * sample generator.  See nvx_synth.h for the arithmetic.                     */
#include "navtex_amd.h"
#include "nvx_synth.h"
#include "nvx_internal.h"

#include <stdlib.h>
#include <string.h>

/* Fill out[0..count) with the periods b = first .. first+count-1 of carrier c.
 * Period b covers samples [b*spb - bit_offset, (b+1)*spb - bit_offset); the
 * phase of period 0 is extrapolated back to its virtual first sample so one
 * formula serves every period:  phase(n) = P[b] + r * inc[b],  r = g - b*spb,
 * g = n + bit_offset.                                                        */
void nvx_synth_periods(const nvx_carrier *c, uint32_t sample_rate, uint64_t first, size_t count,
                       nvx_period *out)
{
    const uint32_t spb = sample_rate / 100;
    const uint32_t inc_b = nvx_phase_inc(c->freq_hz + c->shift_hz, sample_rate);
    const uint32_t inc_y = nvx_phase_inc(c->freq_hz - c->shift_hz, sample_rate);
    uint32_t inc0 = (c->n_bits && c->bits[0] == 'Y') ? inc_y : inc_b;
    uint32_t phase = c->phase0 - c->bit_offset * inc0;          /* mod 2^32 */
    for (uint64_t b = 0; b < first + count; b++) {
        char bit = c->n_bits ? c->bits[b % c->n_bits] : 'B';
        uint32_t inc = (bit == 'Y') ? inc_y : inc_b;
        if (b >= first) { out[b - first].phase = phase; out[b - first].inc = inc; }
        phase += spb * inc;
    }
}

int nvx_synth_host(const nvx_synth_stream *s, uint32_t sample_rate, uint64_t n0, size_t n, int16_t *out)
{
    if (!s || !out || (sample_rate != NVX_RATE_RAW && sample_rate != NVX_RATE_IN) ||
        s->n_carriers < 0 || s->n_carriers > NVX_SYNTH_MAX_CARRIERS) {
        nvx_set_error("nvx_synth_host: bad argument");
        return NVX_ERR_ARG;
    }
    const uint32_t spb = sample_rate / 100;
    nvx_period *per[NVX_SYNTH_MAX_CARRIERS] = { NULL };
    uint64_t first[NVX_SYNTH_MAX_CARRIERS] = { 0 };
    for (int c = 0; c < s->n_carriers; c++) {
        if (s->carrier[c].bit_offset >= spb) { nvx_set_error("nvx_synth_host: bit_offset >= samples per bit"); return NVX_ERR_ARG; }
        first[c] = (n0 + s->carrier[c].bit_offset) / spb;
        uint64_t last = (n0 + n + s->carrier[c].bit_offset) / spb;
        per[c] = (nvx_period *)malloc((size_t)(last - first[c] + 1) * sizeof(nvx_period));
        if (!per[c]) { for (int k = 0; k < c; k++) free(per[k]); nvx_set_error("nvx_synth_host: out of memory"); return NVX_ERR_NOMEM; }
        nvx_synth_periods(&s->carrier[c], sample_rate, first[c], (size_t)(last - first[c] + 1), per[c]);
    }
    uint32_t *o = (uint32_t *)out;
    for (size_t k = 0; k < n; k++) {
        uint64_t idx = n0 + k;
        int32_t I = 0, Q = 0;
        for (int c = 0; c < s->n_carriers; c++) {
            uint64_t g = idx + s->carrier[c].bit_offset;
            uint64_t b = g / spb;
            uint32_t r = (uint32_t)(g - b * spb);
            const nvx_period *p = &per[c][b - first[c]];
            nvx_synth_tone(p->phase + r * p->inc, s->carrier[c].amplitude, &I, &Q);
        }
        if (s->noise_amp > 0) nvx_synth_noise(s->seed, idx, s->noise_amp, &I, &Q);
        uint32_t w = nvx_synth_pack(I, Q);
        memcpy(&o[k], &w, 4);
    }
    for (int c = 0; c < s->n_carriers; c++) free(per[c]);
    return NVX_OK;
}
