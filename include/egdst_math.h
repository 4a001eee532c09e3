/*
 * egdst_math.h -- bit-reproducible exp / log / pow in IEEE-754 binary64.
 *
 * WHY: the DC-EGM algorithm takes discrete decisions from floating-point comparisons (which grid points
 * survive an envelope, on which side of a 1e-10 bisection bracket a crossing lies).  Two libm's that
 * differ in the last bit (glibc on the host vs ocml on the GPU) therefore produce solutions that are
 * equal to ~1e-13 where the algorithm is continuous but can differ in row counts and near kinks.  With
 * the transcendental functions below -- written with +,-,*,/ only, no fused multiply-add, no table
 * lookups, no platform libm -- the GPU path and the CPU oracle evaluate the very same expression trees
 * and agree BIT FOR BIT, so parity tests compare exactly.  Build every translation unit that includes
 * this header with -ffp-contract=off.
 *
 * exp and log follow the classic fdlibm kernels (argument reduction by ln2 in two pieces, minimax
 * polynomial in the reduced argument; error < 1 ulp):
 *   Copyright (C) 1993-2004 by Sun Microsystems, Inc. All rights reserved.
 *   Permission to use, copy, modify, and distribute this software is freely granted, provided that
 *   this notice is preserved.
 * pow(x,y) is exp(y*log(x)) with the special cases the model strings can reach; its error is about
 * (1 + |y ln x|) ulp, i.e. < 5e-15 relative for the CRRA forms of the shipped models.
 *
 * Selection: models are generated with MS_EXP/MS_LOG/MS_POW; defining EGDST_NATIVE_MATH maps them to
 * the platform libm instead (the oracle does this to reproduce the reference's glibc results).
 */
#ifndef EGDST_MATH_H
#define EGDST_MATH_H

#ifndef EGM_FN
#ifdef __HIPCC__
#define EGM_FN static __host__ __device__ __forceinline__
#else
#define EGM_FN static inline
#endif
#endif

EGM_FN unsigned long long egm_bits(double x)
{
    union { double d; unsigned long long u; } c;
    c.d = x;
    return c.u;
}
EGM_FN double egm_from_bits(unsigned long long u)
{
    union { double d; unsigned long long u; } c;
    c.u = u;
    return c.d;
}
EGM_FN unsigned egm_hi(double x) { return (unsigned)(egm_bits(x) >> 32); }
EGM_FN unsigned egm_lo(double x) { return (unsigned)(egm_bits(x) & 0xffffffffu); }
EGM_FN double egm_with_hi(double x, unsigned hi)
{
    return egm_from_bits(((unsigned long long)hi << 32) | (egm_bits(x) & 0xffffffffull));
}

EGM_FN double eg_exp(double x)
{
    const double o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02;
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    const double twom1000 = 9.33263618503218878990e-302; /* 2**-1000 */
    double hi = 0, lo = 0, c, t, y;
    int k = 0;
    unsigned hx = egm_hi(x);
    const int xsb = (int)((hx >> 31) & 1u);
    hx &= 0x7fffffffu;
    if (hx >= 0x40862E42u) { /* |x| >= 709.78... */
        if (hx >= 0x7ff00000u) {
            if (((hx & 0xfffffu) | egm_lo(x)) != 0) return x + x; /* NaN */
            return xsb == 0 ? x : 0.0;                              /* exp(+-inf) */
        }
        if (x > o_threshold) return egm_from_bits(0x7ff0000000000000ull);
        if (x < u_threshold) return 0.0;
    }
    if (hx > 0x3fd62e42u) { /* |x| > 0.5 ln2 */
        if (hx < 0x3FF0A2B2u) { /* and |x| < 1.5 ln2 */
            hi = xsb ? x + ln2HI : x - ln2HI;
            lo = xsb ? -ln2LO : ln2LO;
            k = 1 - xsb - xsb;
        } else {
            k = (int)(invln2 * x + (xsb ? -0.5 : 0.5));
            t = k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x3e300000u) { /* |x| < 2**-28 */
        return 1.0 + x;
    } else
        k = 0;
    t = x * x;
    c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    if (k >= -1021) return egm_with_hi(y, egm_hi(y) + ((unsigned)k << 20));
    y = egm_with_hi(y, egm_hi(y) + ((unsigned)(k + 1000) << 20));
    return y * twom1000;
}

EGM_FN double eg_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double two54 = 1.80143985094819840000e+16;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double hfsq, f, s, z, R, w, t1, t2, dk;
    int k = 0, hx, i, j;
    unsigned lx;
    hx = (int)egm_hi(x);
    lx = egm_lo(x);
    if (hx < 0x00100000) { /* x < 2**-1022 */
        if (((hx & 0x7fffffff) | (int)lx) == 0) return -egm_from_bits(0x7ff0000000000000ull); /* log(+-0) = -inf */
        if (hx < 0) return (x - x) / (x - x);                                                  /* log(-#) = NaN */
        k -= 54;
        x *= two54;
        hx = (int)egm_hi(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    x = egm_with_hi(x, (unsigned)(hx | (i ^ 0x3ff00000))); /* normalize x or x/2 */
    k += (i >> 20);
    f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2**-20 */
        if (f == 0.0) {
            if (k == 0) return 0.0;
            dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    s = f / (2.0 + f);
    dk = (double)k;
    z = s * s;
    i = hx - 0x6147a;
    w = z * z;
    j = 0x6b851 - hx;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    R = t2 + t1;
    if (i > 0) {
        hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* pow for the cases model strings use: positive base, or zero/negative base with the conventions of C
 * where they are unambiguous. */
EGM_FN double eg_pow(double x, double y)
{
    if (y == 0.0) return 1.0;
    if (x == 1.0) return 1.0;
    if (x != x || y != y) return x + y;
    if (x > 0.0) {
        if (y == 1.0) return x;
        if (y == 2.0) return x * x;
        if (y == -1.0) return 1.0 / x;
        return eg_exp(y * eg_log(x));
    }
    if (x == 0.0) return y > 0.0 ? 0.0 : 1.0 / 0.0;
    { /* negative base: defined for integer exponents only */
        const double yi = (double)(long long)y;
        if (yi != y || y > 9.0e15 || y < -9.0e15) return (x - x) / (x - x);
        const double r = eg_exp(y * eg_log(-x));
        return (((long long)y) & 1) ? -r : r;
    }
}

#ifdef EGDST_NATIVE_MATH
#define MS_EXP(x) exp(x)
#define MS_LOG(x) log(x)
#define MS_POW(x, y) pow(x, y)
#else
#define MS_EXP(x) eg_exp(x)
#define MS_LOG(x) eg_log(x)
#define MS_POW(x, y) eg_pow(x, y)
#endif

#endif
