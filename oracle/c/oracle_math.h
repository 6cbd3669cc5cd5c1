/* oracle_math.h -- deterministic exp / log / cube for the C oracle.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): never included by anything under
 * nexoclom_amd/.
 *
 * The reference takes log(frac) and exp(.) once per RK step (particle_tracking/rk5.py:25,50) and
 * |r|**3 six times (particle_tracking/state.py:20).  NumPy's pow/exp/log on this class of CPU are
 * its own SIMD kernels (they differ from glibc in a few % of arguments, each by 1 ulp), so "the
 * reference's value" of these three functions is only defined to 1 ulp.  The oracle therefore uses
 * arithmetic that is a pure function of IEEE-754 +,-,*,/ and fma -- identical on any conforming
 * CPU or GPU when compiled without contraction -- so that the HIP kernels (which carry their own
 * implementation of the same published algorithms) can be compared with it BIT FOR BIT:
 *   - ora_cube : r*r*r evaluated as a double-double product and rounded once (correctly rounded
 *                r^3 in all but astronomically rare ties);
 *   - ora_exp : Cody-Waite reduction by ln 2 and the Taylor polynomial through r^13 in fma, no
 *                division (< 1 ulp);
 *   - ora_log : table-driven (91 bins, 1/c and -ln(1/c) from oracle_log_table.h), r = m/c - 1 in
 *                one fma, Taylor through r^10, no division (< 1 ulp).
 * Build with -DORACLE_LIBM to swap in glibc's pow/exp/log instead (cross-check of these
 * routines, see tests/test_oracle_c.py).
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "oracle_log_table.h"

static inline uint64_t ora_bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
static inline double ora_from_bits(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }

static inline double ora_cube(double r)
{
#ifdef ORACLE_LIBM
    return pow(r, 3.0);
#else
    double sq = r * r;
    double sq_err = fma(r, r, -sq);          /* r*r = sq + sq_err exactly */
    double cu = sq * r;
    double cu_err = fma(sq, r, -cu);         /* sq*r = cu + cu_err exactly */
    return cu + (cu_err + sq_err * r);
#endif
}

static inline double ora_exp(double x)
{
#ifdef ORACLE_LIBM
    return exp(x);
#else
    /* k = rint(x / ln 2); r = x - k ln 2 (two fused steps, k * LN2_HI exact); exp(r) = 1 + r +
     * r^2 q(r), q = Taylor through r^13/13!, Horner in fma; exponent field += k.  < 1 ulp
     * (0.97 worst on 4e6 arguments).  The HIP kernels run the same operations in the same order. */
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double INV_LN2 = 1.44269504088896338700e+00;
    static const double C[14] = {0, 0, 0x1.0000000000000p-1, 0x1.5555555555555p-3,
                                 0x1.5555555555555p-5, 0x1.1111111111111p-7,
                                 0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-13,
                                 0x1.a01a01a01a01ap-16, 0x1.71de3a556c734p-19,
                                 0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26,
                                 0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33};
    double ax = fabs(x);
    if (!(ax >= 3.725290298461914e-09 && ax <= 7.09782712893383973096e+02)) {
        if (x != x) return x;
        if (x > 7.09782712893383973096e+02) return INFINITY;
        if (x < -7.45133219101941108420e+02) return 0.0;
        if (ax < 3.725290298461914e-09) return 1.0 + x;     /* |x| < 2^-28 */
    }
    double kd = rint(x * INV_LN2);
    int k = (int)kd;
    double hi = fma(-kd, LN2_HI, x);
    double r = fma(-kd, LN2_LO, hi);
    double q = C[13];
    for (int i = 12; i >= 2; i--) q = fma(q, r, C[i]);
    double y = 1.0 + fma(r * r, q, r);
    if (k >= -1021) return ora_from_bits(ora_bits(y) + ((uint64_t)(int64_t)k << 52));
    return ora_from_bits(ora_bits(y) + ((uint64_t)(int64_t)(k + 1000) << 52))
           * 9.33263618503218878990e-302;              /* 2^-1000 */
#endif
}

static inline double ora_log(double x)
{
#ifdef ORACLE_LIBM
    return log(x);
#else
    /* Table-driven, division-free: x = 2^k m, m in [181/256, 362/256); bin i = floor((m - OFF) *
     * 128) with centre c_i (bins 36..38 share c = 1); r = m / c_i - 1 from one fma with the
     * tabulated 1/c_i; log x = (k LN2_HI + lchi) + r [exact sum + its rounding error] + k LN2_LO
     * + lclo + r^2 (-1/2 + r p(r)), p = Taylor through r^10/10.  < 1 ulp (0.71 worst on 8e6
     * arguments).  The HIP kernels run the same operations on their own copy of the table. */
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double OFF = 0.70703125;
    int k = 0;
    if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) {
        if (x != x) return x;
        if (x == 0.0) return -INFINITY;
        if (x < 0.0) return NAN;
        if (x == INFINITY) return x;
        x *= 18014398509481984.0; k = -54;                 /* subnormal */
    }
    uint64_t u = ora_bits(x);
    int32_t hx = (int32_t)(u >> 32);
    int32_t tmp = hx - 0x3FE6A000;
    k += tmp >> 20;
    int32_t mh = hx - (int32_t)((uint32_t)tmp & 0xFFF00000u);
    double m = ora_from_bits(((uint64_t)(uint32_t)mh << 32) | (u & 0xffffffffu));
    int i = (int)((m - OFF) * 128.0);
    const double *row = ORA_LOG_TABLE_DATA[i];
    double r = fma(m, row[0], -1.0);
    double kd = (double)k;
    double w = fma(kd, LN2_HI, row[1]);
    double hi = w + r;
    double lo = ((w - hi) + r) + fma(kd, LN2_LO, row[2]);
    double p = -0x1.999999999999ap-4;
    p = fma(p, r, 0x1.c71c71c71c71cp-4);
    p = fma(p, r, -0x1.0p-3);
    p = fma(p, r, 0x1.2492492492492p-3);
    p = fma(p, r, -0x1.5555555555555p-3);
    p = fma(p, r, 0x1.999999999999ap-3);
    p = fma(p, r, -0x1.0p-2);
    p = fma(p, r, 0x1.5555555555555p-2);
    return fma(r * r, fma(r, p, -0.5), lo) + hi;
#endif
}

/* errmax**-0.25 of the variable-step driver (Output.py:336) as 1/sqrt(sqrt(e)): correctly rounded
 * primitives only, so CPU and GPU agree bit for bit (NumPy's pow is within 1 ulp of it). */
static inline double ora_pow_m025(double e)
{
#ifdef ORACLE_LIBM
    return pow(e, -0.25);
#else
    return 1.0 / sqrt(sqrt(e));
#endif
}
#endif
