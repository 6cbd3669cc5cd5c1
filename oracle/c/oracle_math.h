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
 *   - ora_log : the classic table-free argument-reduction + minimax-polynomial scheme of Sun's
 *                fdlibm (e_log.c, 1993/2004, < 1 ulp), coefficients from that publication.
 * Build with -DORACLE_LIBM to swap in glibc's pow/exp/log instead (cross-check of these
 * routines, see tests/test_oracle_c.py).
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

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
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
                 L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                 L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x == 0.0) return -INFINITY;
    if (x < 0.0) return NAN;
    if (x == INFINITY) return x;
    int k = 0;
    if (x < 2.2250738585072014e-308) { x *= 18014398509481984.0; k = -54; }   /* subnormal */
    uint64_t u = ora_bits(x);
    int32_t hx = (int32_t)(u >> 32);
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;             /* mantissa >= sqrt(2): halve it */
    u = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (u & 0xffffffffu);
    x = ora_from_bits(u);
    k += i >> 20;
    double f = x - 1.0, dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {                 /* |f| < 2^-20 */
        if (f == 0.0) return k == 0 ? 0.0 : dk * LN2_HI + dk * LN2_LO;
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        return k == 0 ? f - R : dk * LN2_HI - ((R - dk * LN2_LO) - f);
    }
    double s = f / (2.0 + f), z = s * s, w = z * z;
    double t1 = w * (L2 + w * (L4 + w * L6));
    double t2 = z * (L1 + w * (L3 + w * (L5 + w * L7)));
    double R = t2 + t1;
    i = hx - 0x6147a;
    int32_t j = 0x6b851 - hx;
    if ((i | j) > 0) {
        double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R))
                      : dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * LN2_HI - ((s * (f - R) - dk * LN2_LO) - f);
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
