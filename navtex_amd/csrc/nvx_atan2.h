/* nvx_atan2.h -- libm-free atan2 in double-double arithmetic, one source for
 * host and gfx950 device code.
 *
 * Why it exists: the reference's FSK discriminator calls libm atan2 once per
 * 900 S/s sample (receiver/decoder.C:52).  ROCm's device atan2 is not glibc's,
 * so the device path carries its own, built only from IEEE-exact operations
 * (+ - * / fma), evaluated to ~1e-29 relative before the final rounding.  It
 * therefore returns the correctly rounded atan2 except with probability ~1e-13
 * per call; tests/test_atan2.py measures its agreement with the glibc atan2
 * the oracle uses.
 *
 * Method: a = min(|x|,|y|), b = max(|x|,|y|), k = round(64 a/b), c = k/64;
 *   atan(a/b) = atan(c) + atan(z),  z = (a - c b) / (b + c a),  |z| <= ~1/127
 * with atan(c) from a 65-entry double-double table (tools/gen_atan_table.py),
 * z from exact products, compensated sums and a quotient plus one correction
 * quotient, atan(z) from its Taylor series with the cubic term carried as a
 * (hi, lo) pair and the rest in double.  Octant fix-ups with double-double
 * pi/2 and pi.  Special cases follow C99 Annex F (what glibc
 * implements): signed zeros, infinities, NaN.
 *
 * Compile with -ffp-contract=off: the error-free transformations below must
 * not be fused or reassociated.
 */
#ifndef NVX_ATAN2_H
#define NVX_ATAN2_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#  define NVX_HD __host__ __device__ static inline
#else
#  define NVX_HD static inline
#endif
/* runtime-indexed table: constant memory in the device pass of a HIP compile,
 * an ordinary static array in the host pass and in plain C/C++ builds        */
#if defined(__HIP_DEVICE_COMPILE__)
#  define NVX_ATAN_TABLE __device__ __constant__ static const
#else
#  define NVX_ATAN_TABLE static const
#endif
#include "nvx_atan_table.h"

typedef struct { double hi, lo; } nvx_dd;

NVX_HD nvx_dd nvx_two_sum(double a, double b)
{
    double s = a + b, bb = s - a;
    nvx_dd r = { s, (a - (s - bb)) + (b - bb) };
    return r;
}
NVX_HD nvx_dd nvx_fast_two_sum(double a, double b)      /* |a| >= |b| or a == 0 */
{
    double s = a + b;
    nvx_dd r = { s, b - (s - a) };
    return r;
}
NVX_HD nvx_dd nvx_two_prod(double a, double b)
{
    double p = a * b;
    nvx_dd r = { p, __builtin_fma(a, b, -p) };
    return r;
}
NVX_HD nvx_dd nvx_dd_add(nvx_dd a, nvx_dd b)
{
    nvx_dd s = nvx_two_sum(a.hi, b.hi), t = nvx_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = nvx_fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return nvx_fast_two_sum(s.hi, s.lo);
}
NVX_HD nvx_dd nvx_dd_neg(nvx_dd a) { nvx_dd r = { -a.hi, -a.lo }; return r; }
NVX_HD uint64_t nvx_bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
NVX_HD double nvx_from_bits(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }

NVX_HD double nvx_atan2(double y, double x)
{
    const uint64_t SIGN = 0x8000000000000000ull, EXPM = 0x7ff0000000000000ull;
    uint64_t ux = nvx_bits(x), uy = nvx_bits(y);
    uint64_t sy = uy & SIGN;
    int xneg = (ux & SIGN) != 0;
    double ax = nvx_from_bits(ux & ~SIGN), ay = nvx_from_bits(uy & ~SIGN);

    if (x != x || y != y) return x + y;                                   /* NaN */
    int xinf = (ux & ~SIGN) == EXPM, yinf = (uy & ~SIGN) == EXPM;
    double r;
    if (ay == 0.0) {                       /* +-0 or +-pi, sign of y            */
        r = xneg ? NVX_PI_HI : 0.0;
        return nvx_from_bits(nvx_bits(r) | sy);
    }
    if (ax == 0.0 || (yinf && !xinf)) {    /* +-pi/2                            */
        return nvx_from_bits(nvx_bits((double)NVX_PIO2_HI) | sy);
    }
    if (xinf) {
        if (yinf) r = xneg ? 0x1.2d97c7f3321d2p+1 /* 3pi/4 */ : 0x1.921fb54442d18p-1 /* pi/4 */;
        else      r = xneg ? NVX_PI_HI : 0.0;
        return nvx_from_bits(nvx_bits(r) | sy);
    }

    /* finite, non-zero operands: scale both by the same power of two so that
     * b is in [1, 2) -- exact, and keeps every product below in range        */
    int swap = ay > ax;
    double a = swap ? ax : ay, b = swap ? ay : ax;
    int eb = (int)((nvx_bits(b) >> 52) & 0x7ff);
    if (eb == 0) {                         /* subnormal b: pre-scale by 2^200  */
        a *= 0x1p200; b *= 0x1p200;
        eb = (int)((nvx_bits(b) >> 52) & 0x7ff);
    } else if (eb >= 1536) {               /* huge b: keep 2^(1023-eb) a normal */
        a *= 0x1p-600; b *= 0x1p-600;      /* number (a >= b 2^-121 stays normal */
        eb -= 600;                         /* on the path that uses it)          */
    }
    int ea = (int)((nvx_bits(a) >> 52) & 0x7ff);
    nvx_dd res;
    if (eb - ea > 120) {
        /* a/b < 2^-119: atan(a/b) = a/b to far below double-double precision;
         * one exact quotient is all that is needed                           */
        if (!swap && !xneg) {
            /* the only case where the tiny quotient IS the result            */
            double q = a / b;              /* correctly rounded (IEEE division) */
            return nvx_from_bits(nvx_bits(q) | sy);
        }
        res.hi = 0.0; res.lo = 0.0;        /* vanishes next to pi/2 or pi       */
    } else {
        /* b -> [1,2), a scaled identically (a >= 2^-121: still normal)        */
        uint64_t sc = (uint64_t)(2046 - eb) << 52;           /* 2^(1023-eb)     */
        double s = nvx_from_bits(sc);
        a *= s; b *= s;
        double t = a / b;
        int k = (int)(t * 64.0 + 0.5);
        double c = (double)k * 0.015625;
        /* z = (a - c b) / (b + c a) to ~2^-100: exact products, compensated sums,
         * one quotient and one correction quotient                            */
        nvx_dd pcb = nvx_two_prod(c, b), pca = nvx_two_prod(c, a);
        nvx_dd n1 = nvx_two_sum(a, -pcb.hi);
        nvx_dd num = nvx_fast_two_sum(n1.hi, n1.lo - pcb.lo);
        nvx_dd d1 = nvx_two_sum(b, pca.hi);
        nvx_dd den = nvx_fast_two_sum(d1.hi, d1.lo + pca.lo);
        double q1 = num.hi / den.hi;
        double r1 = __builtin_fma(-q1, den.hi, num.hi);       /* exact remainder of the leading parts */
        r1 += num.lo - q1 * den.lo;
        double q2 = r1 / den.hi;
        nvx_dd z = nvx_fast_two_sum(q1, q2);
        /* atan z = z - z^3/3 + z^5/5 - ...,  |z| <= 1/127.
         * The cubic term is <= 2^-15.6 of the result, so it needs ~2^-60 relative
         * accuracy: w = z^2 and z*w/3 are carried as (hi, lo) pairs.  Everything from
         * z^5 on is below 2^-30 of the result and plain double suffices.       */
        nvx_dd w = nvx_two_prod(z.hi, z.hi);
        w.lo += 2.0 * z.hi * z.lo;
        nvx_dd zw = nvx_two_prod(z.hi, w.hi);
        zw.lo += z.hi * w.lo;
        nvx_dd cub = nvx_two_prod(zw.hi, NVX_THIRD_HI);
        cub.lo += zw.hi * NVX_THIRD_LO + zw.lo * NVX_THIRD_HI;
        const double w1 = w.hi;
        const double tail = z.hi * (w1 * w1 * (1.0 / 5.0 - w1 * (1.0 / 7.0 - w1 * (1.0 / 9.0 - w1 * (1.0 / 11.0 - w1 * (1.0 / 13.0))))));
        nvx_dd s3 = nvx_two_sum(z.hi, -cub.hi);
        nvx_dd atz = nvx_fast_two_sum(s3.hi, s3.lo + ((z.lo - cub.lo) + tail));
        nvx_dd atc = { NVX_ATAN_HI[k], NVX_ATAN_LO[k] };
        res = nvx_dd_add(atc, atz);
    }
    if (swap) { nvx_dd h = { NVX_PIO2_HI, NVX_PIO2_LO }; res = nvx_dd_add(h, nvx_dd_neg(res)); }
    if (xneg) { nvx_dd pi = { NVX_PI_HI, NVX_PI_LO };    res = nvx_dd_add(pi, nvx_dd_neg(res)); }
    r = res.hi + res.lo;
    return nvx_from_bits(nvx_bits(r) | sy);
}

#endif
