/* nvx_synth.h -- integer-only CPFSK sample arithmetic shared by the host
 * generator (nvx_synth_host.c) and the device generator kernel.
 *
 * No reference counterpart: the reference has no signal source but the SDR
 * (SURVEY 7.1).  Everything is integer arithmetic so that host, device and
 * the GPU box regenerate bit-identical IQ from (seed, bits, parameters):
 *   phase      32-bit accumulator, 2^32 = one turn, +inc per sample
 *              ('B' = carrier + shift, 'Y' = carrier - shift: decoder.C:115-125)
 *   waveform   2048-entry int16 sine table, I = cos, Q = sin, amplitude scaled
 *              with an arithmetic shift (floor)
 *   noise      uniform integers from a counter-based hash of (seed, n)
 */
#ifndef NVX_SYNTH_H
#define NVX_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#  define NVX_SHD __host__ __device__ static inline
#else
#  define NVX_SHD static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#  define NVX_SIN_TABLE __device__ __constant__ static const
#else
#  define NVX_SIN_TABLE static const
#endif
#include "nvx_sin_table.h"

/* one bit period of one carrier: phase at its first sample, increment in it */
typedef struct { uint32_t phase; uint32_t inc; } nvx_period;

/* flattened per-stream description the device kernel reads */
#define NVX_SYNTH_CARRIERS 16
typedef struct {
    uint32_t seed;
    int32_t  noise_amp;
    int32_t  n_carriers;
    int32_t  amp[NVX_SYNTH_CARRIERS];
    uint32_t bit_offset[NVX_SYNTH_CARRIERS];
    uint32_t pool_off[NVX_SYNTH_CARRIERS];   /* first nvx_period of this carrier in the pool */
} nvx_synth_desc;

NVX_SHD uint32_t nvx_hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

/* phase increment per sample for f_hz at sample_rate, rounded to nearest */
NVX_SHD uint32_t nvx_phase_inc(int32_t f_hz, uint32_t sample_rate)
{
    int64_t num = (int64_t)f_hz * 4294967296ll;
    int64_t half = (int64_t)(sample_rate / 2);
    int64_t q = (num >= 0) ? (num + half) / (int64_t)sample_rate : -((-num + half) / (int64_t)sample_rate);
    return (uint32_t)q;
}

NVX_SHD void nvx_synth_tone(uint32_t phase, int32_t amp, int32_t *I, int32_t *Q)
{
    uint32_t idx = phase >> (32 - NVX_SIN_BITS);
    int32_t s = NVX_SIN_Q15[idx];
    int32_t c = NVX_SIN_Q15[(idx + NVX_SIN_N / 4) & (NVX_SIN_N - 1)];
    *I += (amp * c) >> 15;
    *Q += (amp * s) >> 15;
}

NVX_SHD void nvx_synth_noise(uint32_t seed, uint64_t n, int32_t noise_amp, int32_t *I, int32_t *Q)
{
    uint32_t h = nvx_hash32(seed ^ nvx_hash32((uint32_t)n ^ (0x9e3779b9u * (uint32_t)(n >> 32))));
    uint32_t span = (uint32_t)(2 * noise_amp + 1);
    *I += (int32_t)(((h & 0xffffu) * span) >> 16) - noise_amp;
    *Q += (int32_t)(((h >> 16) * span) >> 16) - noise_amp;
}

NVX_SHD uint32_t nvx_synth_pack(int32_t I, int32_t Q)
{
    I = I > 32767 ? 32767 : (I < -32768 ? -32768 : I);
    Q = Q > 32767 ? 32767 : (Q < -32768 ? -32768 : Q);
    return ((uint32_t)I & 0xffffu) | ((uint32_t)Q << 16);
}

#endif
