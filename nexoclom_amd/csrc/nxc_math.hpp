// nxc_math.hpp -- device math of the nexoclom hot path on gfx950.
//
// The reference takes log(frac)/exp(.) once per RK step (particle_tracking/rk5.py:25,50), |r|**3
// six times per step (particle_tracking/state.py:20) and errmax**-0.25 on a rejected adaptive step
// (particle_tracking/Output.py:336).  NumPy's own SIMD pow/exp/log are only defined to 1 ulp (they
// differ from glibc's on a few % of arguments), so instead of OCML's routines -- a third set of
// 1-ulp answers -- these are built from IEEE-754 add/mul/div/fma/sqrt alone.  Compiled with
// -ffp-contract=off every operation rounds once, so results are a pure function of the inputs
// and can be checked bit for bit against a CPU evaluation of the same published algorithms
// (tests/ do that; nothing here depends on the test code).
//
//   nxc_cube : r^3 as an error-free (double-double) product rounded once -> correctly rounded
//   nxc_exp : Cody-Waite reduction + Taylor polynomial in fma, no division (< 1 ulp)
//   nxc_log : table-driven (91 bins of 1/128 over [181/256, 362/256), 1/c and -ln(1/c) from the
//             LDS header), r = m/c - 1 in one fma, Taylor through r^10, no division (< 1 ulp;
//             defined in nxc_device.hpp next to the header it reads)
//   nxc_pow_m025 : e^-0.25 = 1 / sqrt(sqrt(e))
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NXC_DEV __device__ __forceinline__

// --- correctly rounded sqrt and division without the range-scaling wrapper ------------------------
// hipcc expands fp64 sqrt and '/' into v_rsq_f64 / v_rcp_f64 + a fixed Newton/fma chain, wrapped in
// v_ldexp / v_div_scale / v_div_fixup steps that only act when an operand is near the ends of the
// exponent range.  Positions, speeds and GM of this problem sit within 2^+-200, so the wrappers are
// dead weight in a VALU-bound kernel.  These helpers run the SAME fma chains (so they return the
// same, correctly rounded, bits; checked against NumPy on the GPU in tests/test_gpu_parity.py)
// and fall back to the compiler's full sequence outside that range.  A reciprocal refined once is
// shared by the three quotients of the gravity term.
NXC_DEV bool nxc_mid_range(double a)
{
    const double m = __builtin_fabs(a);
    return (m > 0x1p-200) && (m < 0x1p+200);
}

// sqrt for an argument already known to lie in the middle exponent range
NXC_DEV double nxc_sqrt_mid(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d0 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, x);
    return __builtin_fma(d1, h, g);
}

NXC_DEV double nxc_sqrt(double x)
{
    if (!nxc_mid_range(x)) return __builtin_sqrt(x);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d0 = __builtin_fma(-g, g, x);
    g = __builtin_fma(d0, h, g);
    const double d1 = __builtin_fma(-g, g, x);
    return __builtin_fma(d1, h, g);
}

// 1/d refined by two Newton steps (the y of the division chain below)
NXC_DEV double nxc_recip_seed(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(y, e, y);
}

// n/d given y = nxc_recip_seed(d); both operands in the middle exponent range
NXC_DEV double nxc_div_seeded(double n, double d, double y)
{
    const double q = n * y;
    const double r = __builtin_fma(-d, q, n);
    return __builtin_fma(r, y, q);
}

// n/d for operands known to be mid-range (n may also be exactly 0)
NXC_DEV double nxc_div_mid(double n, double d) { return nxc_div_seeded(n, d, nxc_recip_seed(d)); }

NXC_DEV double nxc_div(double n, double d)
{
    if (!(nxc_mid_range(d) && (nxc_mid_range(n) || n == 0.0))) return n / d;
    return nxc_div_seeded(n, d, nxc_recip_seed(d));
}

// Three quotients by one divisor, each equal to nxc_div's (one refined reciprocal serves all).
NXC_DEV void nxc_div3(double n0, double n1, double n2, double d, double &q0, double &q1, double &q2)
{
    const bool ok = nxc_mid_range(d) && (nxc_mid_range(n0) || n0 == 0.0) &&
                    (nxc_mid_range(n1) || n1 == 0.0) && (nxc_mid_range(n2) || n2 == 0.0);
    if (!ok) {
        q0 = n0 / d; q1 = n1 / d; q2 = n2 / d;
        return;
    }
    const double y = nxc_recip_seed(d);
    q0 = nxc_div_seeded(n0, d, y);
    q1 = nxc_div_seeded(n1, d, y);
    q2 = nxc_div_seeded(n2, d, y);
}

NXC_DEV double nxc_cube(double r)
{
    double sq = r * r;
    double sq_lo = __builtin_fma(r, r, -sq);
    double cu = sq * r;
    double cu_lo = __builtin_fma(sq, r, -cu);
    return cu + (cu_lo + sq_lo * r);
}

// exp(x) without a division: k = rint(x / ln 2), r = x - k ln 2 in two fused steps (Cody-Waite:
// k * LN2_HI is exact, LN2_HI has 21 trailing zero bits), exp(r) = 1 + r + r^2 q(r) with q the
// Taylor polynomial through r^13 / 13! (|r| <= ln2/2: truncation 4e-18), Horner in fma, then the
// exponent field takes k.  Largest error seen on 4e6 arguments over [-700, 700]: 0.97 ulp (mean
// 0.27).  The C oracle carries the same operations in the same order (the checker under oracle/c);
// round 1 used fdlibm's rational form, whose division cost 11 of the routine's 62 issue slots --
// this one takes 21.
// A polynomial coefficient handed to v_fma_f64 from scalar registers.  Left alone, the compiler
// loads each coefficient into a vector register pair (two v_mov_b32, VALU issue slots) to use it as
// v_fmac_f64's accumulator; this costs two s_mov_b32 on the scalar unit instead, which has room.
NXC_DEV double nxc_sconst(double c)
{
    asm("" : "+s"(c));
    return c;
}

NXC_DEV double nxc_exp(double x)
{
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    constexpr double INV_LN2 = 1.44269504088896338700e+00;
    constexpr double C2 = 0x1.0000000000000p-1, C3 = 0x1.5555555555555p-3,
                     C4 = 0x1.5555555555555p-5, C5 = 0x1.1111111111111p-7,
                     C6 = 0x1.6c16c16c16c17p-10, C7 = 0x1.a01a01a01a01ap-13,
                     C8 = 0x1.a01a01a01a01ap-16, C9 = 0x1.71de3a556c734p-19,
                     C10 = 0x1.27e4fb7789f5cp-22, C11 = 0x1.ae64567f544e4p-26,
                     C12 = 0x1.1eed8eff8d898p-29, C13 = 0x1.6124613a86d09p-33;
    const double ax = __builtin_fabs(x);
    // one test sends NaN, overflow, underflow and the tiny arguments to the rare path
    if (__builtin_expect(!(ax >= 3.725290298461914e-09 && ax <= 7.09782712893383973096e+02), 0)) {
        if (x != x) return x;
        if (x > 7.09782712893383973096e+02) return __builtin_huge_val();
        if (x < -7.45133219101941108420e+02) return 0.0;
        if (ax < 3.725290298461914e-09) return 1.0 + x;
        // -745.13 <= x < -709.78: falls through to the general path (denormal results)
    }
    const double kd = __builtin_rint(x * INV_LN2);
    const int k = (int)kd;                                 // |kd| <= 1075
    const double hi = __builtin_fma(-kd, LN2_HI, x);
    const double r = __builtin_fma(-kd, LN2_LO, hi);
    double q = C13;
#define NXC_HORNER(C) q = __builtin_fma(q, r, nxc_sconst(C))
    NXC_HORNER(C12); NXC_HORNER(C11); NXC_HORNER(C10); NXC_HORNER(C9); NXC_HORNER(C8);
    NXC_HORNER(C7);  NXC_HORNER(C6);  NXC_HORNER(C5);  NXC_HORNER(C4); NXC_HORNER(C3);
    NXC_HORNER(C2);
#undef NXC_HORNER
    const double y = 1.0 + __builtin_fma(r * r, q, r);
    if (k >= -1021)
        return __longlong_as_double(__double_as_longlong(y) + ((long long)k << 52));
    return __longlong_as_double(__double_as_longlong(y) + ((long long)(k + 1000) << 52))
           * 9.33263618503218878990e-302;
}

// nxc_log lives in nxc_device.hpp: it reads its 91-entry table from the LDS header.
NXC_DEV double nxc_log(double x);

NXC_DEV double nxc_pow_m025(double e) { return nxc_div(1.0, nxc_sqrt(nxc_sqrt(e))); }
