/*
 * egdst_math.h -- exp / log / pow in IEEE-754 binary64 that return THE SAME DOUBLE on the GPU, in the CPU
 * oracle and in glibc's libm.
 *
 * WHY: the DC-EGM algorithm takes discrete decisions from floating-point comparisons (which grid points
 * survive an envelope, on which side of a 1e-10 bisection bracket a crossing lies, whether a segment folds
 * back).  Two libm's that differ in the last bit give solutions that agree to ~1e-13 where the algorithm is
 * continuous but differ in row counts and thresholds near kinks (round 1 measured this between an fdlibm-style
 * device libm and glibc at T=60, n=1000: 10 of 60 periods with different row counts).  The reference MEX runs
 * on the host's libm; on Linux x86-64 with FMA that is glibc's table-driven exp/log/pow (Szabolcs Nagy's
 * routines from ARM optimized-routines, in glibc since 2.28; sysdeps/ieee754/dbl-64/e_exp.c, e_log.c,
 * e_pow.c, built with -mfma for the __*_fma ifunc variants).  This header restates that algorithm operation by
 * operation -- the same tables (include/egdst_math_tables.h), the same polynomial evaluation order and the same
 * fused multiply-adds the x86-64 FMA build of glibc 2.35 performs (read off the machine code of
 * __ieee754_exp_fma / __ieee754_log_fma / __ieee754_pow_fma) -- with explicit fma() calls and nothing left to
 * the compiler: every translation unit that includes it is built with -ffp-contract=off.  Result:
 * eg_exp/eg_log/eg_pow == glibc exp/log/pow bit for bit (tests/test_math_vs_libm.py: 0 ulp over 10^7 arguments
 * per function, special values included), hence GPU == portable oracle == glibc oracle.
 *
 * PROVENANCE AND LICENCE: the algorithm and the constants are those of the exp / log / pow of ARM optimized-routines
 * (Szabolcs Nagy; "MIT OR Apache-2.0 WITH LLVM-exception"), which glibc imported in 2.28 and distributes under the LGPL 2.1+.
 * Nothing of either code base is copied here: the functions below are written from the published algorithm and checked
 * against the behaviour of the installed libm, and the tables in egdst_math_tables.h are numeric data extracted by
 * tools/make_math_tables.py from the libm-2.35.a of this image (the same numbers optimized-routines publishes in
 * math/exp_data.c, log_data.c, pow_log_data.c).  A distributor who wants no question about the LGPL can regenerate the tables
 * from the MIT-licensed optimized-routines sources instead: the values are identical.
 *
 * Selection: models are generated with MS_EXP/MS_LOG/MS_POW; defining EGDST_NATIVE_MATH maps them to the
 * platform libm instead (the oracle's second build, which checks the claim above at full problem sizes).
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
#ifndef EGM_TABLE
#define EGM_TABLE static const
#endif
#include "egdst_math_tables.h"

typedef unsigned long long egm_u64;

EGM_FN egm_u64 egm_bits(double x)
{
    union { double d; egm_u64 u; } c;
    c.d = x;
    return c.u;
}
EGM_FN double egm_from_bits(egm_u64 u)
{
    union { double d; egm_u64 u; } c;
    c.u = u;
    return c.d;
}
#define EGM_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define EGM_INF egm_from_bits(0x7ff0000000000000ull)

/* exp's result when the scale 2^(k/N) is not a normal double (|x| > 512): e_exp.c specialcase(). */
EGM_FN double egm_exp_special(double tmp, egm_u64 sbits, egm_u64 ki, int signed_scale)
{
    double scale, y;
    if ((ki & 0x80000000ull) == 0) { /* k > 0: the exponent of scale may have overflowed by <= 460 */
        sbits -= 1009ull << 52;
        scale = egm_from_bits(sbits);
        return 0x1p1009 * EGM_FMA(scale, tmp, scale);
    }
    sbits += 1022ull << 52; /* k < 0: care in the subnormal range */
    scale = egm_from_bits(sbits);
    const double st = scale * tmp; /* glibc's binary forms this product once and does not fuse it */
    y = scale + st;
    if ((y < 0 ? -y : y) < 1.0) {
        double one = 1.0, hi, lo;
        if (signed_scale && y < 0.0) one = -1.0;
        lo = scale - y + st;
        hi = one + y;
        lo = one - hi + y + lo;
        y = (hi + lo) - one;
        if (y == 0.0) y = signed_scale ? egm_from_bits(sbits & 0x8000000000000000ull) : 0.0;
    }
    return 0x1p-1022 * y;
}

/* The shared tail of exp(x) and pow's exp_inline(x, xtail, sign_bias): e_exp.c:__exp, e_pow.c:exp_inline.
 * abstop is the biased exponent of x, or 0 when the result needs egm_exp_special. */
EGM_FN double egm_exp_core(double x, double xtail, int have_tail, unsigned abstop, egm_u64 sign_bias)
{
    const double InvLn2N = egm_exp_k[0], Shift = egm_exp_k[1], NegLn2hiN = egm_exp_k[2], NegLn2loN = egm_exp_k[3];
    const double C2 = egm_exp_k[4], C3 = egm_exp_k[5], C4 = egm_exp_k[6], C5 = egm_exp_k[7];
    double kd = EGM_FMA(x, InvLn2N, Shift); /* z + Shift, fused in the FMA build */
    const egm_u64 ki = egm_bits(kd);
    kd -= Shift;
    double r = EGM_FMA(kd, NegLn2hiN, x);
    r = EGM_FMA(kd, NegLn2loN, r);
    if (have_tail) r += xtail;
    const unsigned idx = 2u * (unsigned)(ki & 127u);
    const egm_u64 top = (ki + sign_bias) << 45;
    const double tail = egm_from_bits(egm_exp_tab[idx]);
    const egm_u64 sbits = egm_exp_tab[idx + 1] + top;
    const double r2 = r * r;
    const double p23 = EGM_FMA(C3, r, C2);                 /* C2 + r*C3 */
    const double p45 = EGM_FMA(r, C5, C4);                 /* C4 + r*C5 */
    double tmp = EGM_FMA(p23, r2, r + tail);               /* tail + r + r2*(C2 + r*C3) */
    tmp = EGM_FMA(p45, r2 * r2, tmp);                      /* ... + r2*r2*(C4 + r*C5) */
    if (abstop == 0) return egm_exp_special(tmp, sbits, ki, have_tail);
    const double scale = egm_from_bits(sbits);
    return EGM_FMA(tmp, scale, scale);
}

EGM_FN double eg_exp(double x)
{
    const egm_u64 ix = egm_bits(x);
    unsigned abstop = (unsigned)(ix >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x3fu) {                         /* |x| < 2^-54 or |x| >= 512 or nan */
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x; /* tiny */
        if (abstop >= 0x409u) {                             /* |x| >= 1024 */
            if (ix == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (ix >> 63) ? 0.0 : EGM_INF;
        }
        abstop = 0; /* large x is special cased in the core */
    }
    return egm_exp_core(x, 0.0, 0, abstop, 0);
}

EGM_FN double eg_log(double x)
{
    egm_u64 ix = egm_bits(x);
    const unsigned top = (unsigned)(ix >> 48);
    const double Ln2hi = egm_log_k[0], Ln2lo = egm_log_k[1];
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) { /* 1-2^-4 <= x < 1+0x1.09p-4 */
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double B0 = egm_log_k[7], B1 = egm_log_k[8], B2 = egm_log_k[9], B3 = egm_log_k[10], B4 = egm_log_k[11],
                     B5 = egm_log_k[12], B6 = egm_log_k[13], B7 = egm_log_k[14], B8 = egm_log_k[15], B9 = egm_log_k[16],
                     B10 = egm_log_k[17];
        const double r = x - 1.0, r2 = r * r, r3 = r * r2;
        double q1 = EGM_FMA(r2, B3, EGM_FMA(B2, r, B1));   /* B1 + r*B2 + r2*B3 */
        double q4 = EGM_FMA(r2, B6, EGM_FMA(B5, r, B4));   /* B4 + r*B5 + r2*B6 */
        double q7 = EGM_FMA(r2, B9, EGM_FMA(B8, r, B7));   /* B7 + r*B8 + r2*B9 */
        q7 = EGM_FMA(r3, B10, q7);                         /* ... + r3*B10 */
        q4 = EGM_FMA(q7, r3, q4);
        q1 = EGM_FMA(q4, r3, q1);                          /* y = r3*q1 is added below */
        const double rw = EGM_FMA(r, 0x1p27, r);           /* r + w, w = r*2^27 */
        const double rhi = EGM_FMA(-0x1p27, r, rw);        /* (r + w) - w */
        const double rlo = r - rhi;
        const double rhi2 = rhi * rhi;
        const double hi = EGM_FMA(rhi2, B0, r);            /* r + rhi*rhi*B0 */
        double lo = EGM_FMA(rhi2, B0, r - hi);             /* r - hi + w */
        lo = EGM_FMA(B0 * rlo, rhi + r, lo);
        return hi + EGM_FMA(q1, r3, lo);
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) { /* x < 2^-1022, inf or nan */
        if (ix * 2 == 0) return -EGM_INF;
        if (ix == 0x7ff0000000000000ull) return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return (x - x) / (x - x);
        ix = egm_bits(x * 0x1p52); /* subnormal: normalise */
        ix -= 52ull << 52;
    }
    const double A0 = egm_log_k[2], A1 = egm_log_k[3], A2 = egm_log_k[4], A3 = egm_log_k[5], A4 = egm_log_k[6];
    const egm_u64 tmp = ix - 0x3fe6000000000000ull;
    const unsigned i = (unsigned)(tmp >> 45) & 127u;
    const int k = (int)((long long)tmp >> 52);
    const double z = egm_from_bits(ix - (tmp & 0xfff0000000000000ull));
    const double invc = egm_log_tab[2 * i], logc = egm_log_tab[2 * i + 1];
    const double r = EGM_FMA(z, invc, -1.0);
    const double kd = (double)k;
    const double w = EGM_FMA(kd, Ln2hi, logc);
    const double hi = w + r;
    const double lo = EGM_FMA(kd, Ln2lo, w - hi + r);
    const double r2 = r * r;
    const double p12 = EGM_FMA(A2, r, A1);                 /* A1 + r*A2 */
    const double p34 = EGM_FMA(r, A4, A3);                 /* A3 + r*A4 */
    const double p = EGM_FMA(p34, r2, p12);
    const double y = EGM_FMA(r * r2, p, EGM_FMA(r2, A0, lo));
    return y + hi;
}

/* 0: y is not an integer, 1: odd integer, 2: even integer (e_pow.c checkint) */
EGM_FN int egm_checkint(egm_u64 iy)
{
    const int e = (int)(iy >> 52) & 0x7ff;
    if (e < 0x3ff) return 0;
    if (e > 0x3ff + 52) return 2;
    if (iy & ((1ull << (0x3ff + 52 - e)) - 1)) return 0;
    if (iy & (1ull << (0x3ff + 52 - e))) return 1;
    return 2;
}
EGM_FN int egm_zeroinfnan(egm_u64 i) { return 2 * i - 1 >= 2 * 0x7ff0000000000000ull - 1; }

EGM_FN double eg_pow(double x, double y)
{
    egm_u64 sign_bias = 0;
    egm_u64 ix = egm_bits(x);
    const egm_u64 iy = egm_bits(y);
    unsigned topx = (unsigned)(ix >> 52);
    const unsigned topy = (unsigned)(iy >> 52);
    if (topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu) {
        if (egm_zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0;
            if (ix == 0x3ff0000000000000ull) return 1.0;
            if (2 * ix > 2 * 0x7ff0000000000000ull || 2 * iy > 2 * 0x7ff0000000000000ull) return x + y;
            if (2 * ix == 2 * 0x3ff0000000000000ull) return 1.0;
            if ((2 * ix < 2 * 0x3ff0000000000000ull) == !(iy >> 63)) return 0.0;
            return y * y;
        }
        if (egm_zeroinfnan(ix)) {
            double x2 = x * x;
            if ((ix >> 63) && egm_checkint(iy) == 1) x2 = -x2;
            return (iy >> 63) ? 1 / x2 : x2;
        }
        if (ix >> 63) { /* finite x < 0 */
            const int yint = egm_checkint(iy);
            if (yint == 0) return (x - x) / (x - x);
            if (yint == 1) sign_bias = 0x800ull << 7;
            ix &= 0x7fffffffffffffffull;
            topx &= 0x7ffu;
        }
        if ((topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu) {
            if (ix == 0x3ff0000000000000ull) return 1.0;
            if ((topy & 0x7ffu) < 0x3beu) return ix > 0x3ff0000000000000ull ? 1.0 + y : 1.0 - y;
            return (ix > 0x3ff0000000000000ull) == (topy < 0x800u) ? EGM_INF : 0.0;
        }
        if (topx == 0) { /* subnormal x */
            ix = egm_bits(x * 0x1p52);
            ix &= 0x7fffffffffffffffull;
            ix -= 52ull << 52;
        }
    }
    /* log_inline: hi + lo = log(x) to ~68 bits */
    const double Ln2hi = egm_powlog_k[0], Ln2lo = egm_powlog_k[1], A0 = egm_powlog_k[2], A1 = egm_powlog_k[3],
                 A2 = egm_powlog_k[4], A3 = egm_powlog_k[5], A4 = egm_powlog_k[6], A5 = egm_powlog_k[7],
                 A6 = egm_powlog_k[8];
    const egm_u64 tmp = ix - 0x3fe6955500000000ull;
    const unsigned i = (unsigned)(tmp >> 45) & 127u;
    const int k = (int)((long long)tmp >> 52);
    const double z = egm_from_bits(ix - (tmp & 0xfff0000000000000ull));
    const double kd = (double)k;
    const double invc = egm_powlog_tab[3 * i], logc = egm_powlog_tab[3 * i + 1], logctail = egm_powlog_tab[3 * i + 2];
    const double r = EGM_FMA(z, invc, -1.0);
    const double t1 = EGM_FMA(kd, Ln2hi, logc);
    const double t2 = t1 + r;
    const double lo1 = EGM_FMA(kd, Ln2lo, logctail);
    const double lo2 = t1 - t2 + r;
    const double ar = A0 * r, ar2 = r * ar, ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = EGM_FMA(ar, r, -ar2);
    const double lo4 = t2 - hi + ar2;
    const double p12 = EGM_FMA(A2, r, A1), p34 = EGM_FMA(A4, r, A3), p56 = EGM_FMA(r, A6, A5);
    const double p = EGM_FMA(ar2, EGM_FMA(p56, ar2, p34), p12);
    const double lo = EGM_FMA(ar3, p, lo1 + lo2 + lo3 + lo4);
    const double lhi = hi + lo;
    const double llo = hi - lhi + lo;
    const double ehi = y * lhi;
    const double elo = EGM_FMA(y, llo, EGM_FMA(lhi, y, -ehi));
    /* exp_inline(ehi, elo, sign_bias) */
    unsigned abstop = (unsigned)(egm_bits(ehi) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) {
            const double one = 1.0 + ehi;
            return sign_bias ? -one : one;
        }
        if (abstop >= 0x409u) {
            if (egm_bits(ehi) >> 63) return sign_bias ? -0.0 : 0.0;
            return sign_bias ? -EGM_INF : EGM_INF;
        }
        abstop = 0;
    }
    return egm_exp_core(ehi, elo, 1, abstop, sign_bias);
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
