// nxc_device.hpp -- device-side building blocks shared by every kernel of the hot path:
// LDS-resident lookup tables with np.interp semantics, the force/loss model, one Dormand-Prince
// step, the post-step fate tests and the per-sample image accumulation.
//
// Arithmetic contract: every expression below is written in the reference's operation order and
// the translation unit is compiled with -ffp-contract=off, so each fp64 operation rounds exactly
// once, as NumPy's element-wise kernels do.  Citations are paths under the reference tree's
// nexoclom/ directory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "nxc_math.hpp"

// ---------------------------------------------------------------------------------------------
// Lookup table with np.interp semantics (numpy compiled_base.c arr_interp): the end values outside
// the table, else slope_j*(x - xp[j]) + fp[j] with slope_j = (fp[j+1]-fp[j])/(xp[j+1]-xp[j])
// (computed once on the host: same IEEE quotient; at a node the product is +-0 and the sum is
// fp[j], np.interp's special case).
//
// Global image of one table of n nodes, staged verbatim into LDS (32-byte aligned).  Rows:
//   row 0      "below the table"  {xp = -DBL_MAX, xn = xp[0]}   {fp = fp[0],   slope = 0}
//   row j + 1  node j             {xp[j], xp[j+1] (+inf last)}  {fp[j], slope_j (0 for the last)}
//   row n + 1  sentinel           {+inf, +inf}                  {fp[n-1], 0}
// stored as two 16-byte-stride arrays ({xp, xn} pairs, then {fp, slope} pairs: a ds_read_b128 of
// random rows then spreads over 16 bank quads instead of 8, which halves the LDS bank conflicts),
// followed by (ncell + 2) uint16 cell entries, padded to a multiple of 32 bytes.
//
// A lookup needs no clamp and no lower-bound test: the cell of x is
//   c = clamp(int((x - xbase) * inv_w), 0, ncell + 1),   xbase = xp[0] - one cell width,
// so cell 0 collects everything below the table and cell ncell + 1 everything above it (the int
// conversion saturates; the end rows have slope 0, so slope*(x - xp) + fp is the end value
// there), and cell[c] = the last row whose xp is <= the SMALLEST double that maps to cell c -- the
// host finds that double by bisection over this very arithmetic (nxc_api.hip: pack_lut), so
// xp[row] <= x holds for every x of the cell whatever the rounding.  The lookup reads rows r and
// r + 1 (four ds_read_b128; the +16 is an immediate offset) and takes the first whose xn is above
// x; cells are fine enough (ncell >= 4n) that this almost always hits, otherwise a walk over the
// rows finds the interval.  The result is exactly np.interp's for every finite or NaN x; an
// INFINITE x gives NaN where np.interp gives the end value (0 * inf) -- callers treat a
// non-finite state as an error anyway.
// ---------------------------------------------------------------------------------------------
// A table as the kernels see it: byte offsets of its three arrays inside the LDS block (the host
// resolves them when it places the table in the blob, so a lookup adds nothing up), the index of
// the last cell entry and of the sentinel row, and the cell transform.
struct LutDesc {          // host-filled; in kernel arguments and in the LDS header
    int rec, fs, cell;    // {xp, xn} pairs, {fp, slope} pairs, cell index
    int top;              // ncell + 1: the last cell entry
    int last;             // n + 1: the sentinel row
    int pad_;
    double xbase;         // xp[0] - (xp[n-1] - xp[0]) / ncell
    double inv_w;         // ncell / (xp[n-1] - xp[0])
};
typedef LutDesc LutView;

// All tables live in the workgroup's dynamic LDS block; they are addressed by byte offset from
// its base so that every access is visibly an LDS (ds_read) access to the compiler.
extern __shared__ __align__(32) unsigned char nxc_lds[];

// Reads by absolute LDS address.  The dynamic block is the only LDS these kernels use, so it
// starts at LDS address 0 (stage_tables checks) and a byte offset IS the address: going through
// the nxc_lds symbol instead makes the compiler add its (zero) address to every computed offset,
// one wasted VALU instruction per table read.
#define NXC_LDS_AS __attribute__((address_space(3)))
#define NXC_GLOBAL_AS __attribute__((address_space(1)))
typedef double nxc_v2d __attribute__((ext_vector_type(2)));

NXC_DEV double lds_f64(int byte_off)
{
    return *(const NXC_LDS_AS double *)(unsigned)byte_off;
}
NXC_DEV int lds_u16(int byte_off)
{
    return *(const NXC_LDS_AS unsigned short *)(unsigned)byte_off;
}

NXC_DEV double2 lds_f64x2(int byte_off)
{
    const nxc_v2d v = *(const NXC_LDS_AS nxc_v2d *)(unsigned)byte_off;
    return make_double2(v.x, v.y);
}

NXC_DEV const LutView &lut_view(const LutDesc &d) { return d; }

// The cell of x: v_cvt_u32_f64 saturates (negative -> 0, too large -> UINT_MAX) and turns NaN into
// 0, which leaves one unsigned minimum to clamp; as the bare instruction because a C++
// double -> unsigned cast of an out-of-range value is undefined.
NXC_DEV int lut_cell(const LutView &t, double x)
{
    const double s = (x - t.xbase) * t.inv_w;
    unsigned c;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(c) : "v"(s));
    const unsigned top = (unsigned)t.top;
    return (int)(c > top ? top : c);
}

// A lookup in two halves, so that a caller can put independent arithmetic between the LDS reads
// and their first use (the two dependent round trips then run under that arithmetic):
//   lut_probe   cell -> rows r and r + 1 of both arrays (four ds_read_b128; the +16 is an
//               immediate offset)
//   lut_finish  select the row, walk in the rare miss, interpolate
struct LutProbe {
    double x;
    int r;
    double2 a0, a1, b0, b1;
};

NXC_DEV void lut_probe_cell(const LutView &t, double x, LutProbe &p)
{
    p.x = x;
    p.r = lds_u16(t.cell + 2 * lut_cell(t, x));
}

NXC_DEV void lut_probe_rows(const LutView &t, LutProbe &p)
{
    // each address as (launch constant) + 16 r: one v_lshl_add_u32 apiece (deriving the second
    // from the first costs the compiler an extra add for the LDS block's own base)
    const int ra = t.rec + 16 * p.r, rb = t.fs + 16 * p.r;
    p.a0 = lds_f64x2(ra); p.a1 = lds_f64x2(ra + 16);
    p.b0 = lds_f64x2(rb); p.b1 = lds_f64x2(rb + 16);
}

NXC_DEV LutProbe lut_probe(const LutView &t, double x)
{
    LutProbe p;
    lut_probe_cell(t, x, p);
    lut_probe_rows(t, p);
    return p;
}

NXC_DEV double lut_finish(const LutView &t, const LutProbe &p)
{
    const double x = p.x;
    const bool in0 = x < p.a0.y, in1 = x < p.a1.y;      // xp[r] <= x is built into the cell table
    double xp = in0 ? p.a0.x : p.a1.x;
    double fp = in0 ? p.b0.x : p.b1.x;
    double sl = in0 ? p.b0.y : p.b1.y;
    if (__builtin_expect(!(in0 || in1), 0)) {            // rare (and NaN): walk up to the interval
        int r = p.r + 1;
        while (r < t.last && x >= lds_f64(t.rec + 16 * r + 8)) ++r;   // ends at the sentinel
        const double2 aw = lds_f64x2(t.rec + 16 * r), bw = lds_f64x2(t.fs + 16 * r);
        xp = aw.x; fp = bw.x; sl = bw.y;
    }
    return sl * (x - xp) + fp;
}

NXC_DEV double lut_interp(const LutView &t, double x) { return lut_finish(t, lut_probe(t, x)); }

// ---------------------------------------------------------------------------------------------
// Force / loss model: particle_tracking/state.py:17-74
// ---------------------------------------------------------------------------------------------
enum : int { LOSS_NONE = 0, LOSS_LIFETIME = 1, LOSS_PHOTO = 2 };

struct ForceK {           // kernel-argument copy of nxc_forces' scalars
    double GM, vrplanet, photo, inv_lifetime;
    int grav, rad, loss, pad_;   // wave-uniform switches (inputs.forces.*, lifetime/photo mode)
    LutDesc tab;                 // radiation-acceleration table inside the blob
};

// (sqrt(x*x + z*z) > 1) | (y < 0)  (state.py:28-29,50-51).  For a correctly rounded sqrt,
// sqrt(s) > 1  <=>  s > 1 + 2^-52 (sqrt(1+2^-52) rounds to 1), so the root is not taken.
NXC_DEV bool sunlit(double x, double y, double z)
{
    double s = x * x + z * z;
    return (s > 0x1.0000000000001p+0) || (y < 0.0);
}

// FULL = gravity + radiation pressure + photo-loss known at compile time (the common run): the
// wave-uniform switches disappear and the force evaluation becomes one basic block, which lets
// the scheduler overlap the table's LDS round trips with the gravity arithmetic.
template <bool FULL>
NXC_DEV void state_eval(const ForceK &F, const LutView &T, double x, double y, double z, double vy,
                        double &ax, double &ay, double &az, double &ion)
{
    const bool lit = sunlit(x, y, z);
    // The lookup's two dependent LDS round trips are laid under the gravity arithmetic: cell read
    // | s2, GM r, sqrt | row reads | cube, reciprocal, quotients | select + interpolate.  The
    // gravity code is branch-free (its full-range fall-back is a fix-up afterwards) and the
    // scheduling barriers keep the compiler from pulling the first uses of the LDS data (and with
    // them the s_waitcnt) up in front of the arithmetic.
    constexpr bool OVERLAP = FULL;
    LutProbe probe{};
    if (FULL || F.rad) lut_probe_cell(T, vy + F.vrplanet, probe);     // state.py:27-36
    double gx = 0.0, gy = 0.0, gz = 0.0;
    if (FULL || F.grav) {                                 // state.py:19-21
        const double s2 = (x * x + y * y) + z * z;
        const double nx = F.GM * x, ny = F.GM * y, nz = F.GM * z;
        // r in 2^+-65, r^3 in 2^+-195: the un-wrapped sqrt / division chains are exact there; one
        // refined reciprocal serves the three quotients.  Evaluated unconditionally ...
        const double r = nxc_sqrt_mid(s2);
        if (OVERLAP) {
            asm volatile("" : : "v"(r), "v"(nx), "v"(ny), "v"(nz));
            __builtin_amdgcn_sched_barrier(0);
            lut_probe_rows(T, probe);
        }
        const double r3 = nxc_cube(r);
        const double rinv = nxc_recip_seed(r3);
        gx = nxc_div_seeded(nx, r3, rinv);
        gy = nxc_div_seeded(ny, r3, rinv);
        gz = nxc_div_seeded(nz, r3, rinv);
        if (OVERLAP) {
            // a use of the quotients HERE: without it the compiler sinks the gravity block below
            // the lookup's rare-miss branch, i.e. behind the LDS waits it is meant to cover
            asm volatile("" : : "v"(gx), "v"(gy), "v"(gz));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!OVERLAP && (FULL || F.rad)) lut_probe_rows(T, probe);
    double ry = 0.0;
    if (FULL || F.rad) {
        // interp * out_of_shadow: the product with False is a zero whose sign cannot matter in
        // gy + ry
        const double a = lut_finish(T, probe);
        ry = lit ? a : 0.0;
    }
    if (FULL || F.grav) {
        const double s2 = (x * x + y * y) + z * z;
        const double nx = F.GM * x, ny = F.GM * y, nz = F.GM * z;
        // ... and replaced by the compiler's full-range sequences when s2 is not in
        // [2^-130, 2^+130) (exponent field: one integer subtract + compare; NaN fails it too)
        if (__builtin_expect(!((unsigned)(__double2hiint(s2) - 0x37d00000) <
                               (unsigned)(0x48100000 - 0x37d00000)), 0)) {
            const double r3s = nxc_cube(__builtin_sqrt(s2));
            gx = nx / r3s; gy = ny / r3s; gz = nz / r3s;
        }
    }
    ax = gx;                        // state.py:41 adds 0.0 here: only the sign of a zero differs
    ay = gy + ry;
    az = gz;
    if (FULL) ion = lit ? F.photo : 0.0;
    else if (F.loss == LOSS_LIFETIME) ion = F.inv_lifetime;    // state.py:44-46
    else if (F.loss == LOSS_PHOTO) ion = lit ? F.photo : 0.0;   // state.py:48-52 (photo * bool)
    else ion = 0.0;
}

// Moons and plasma-torus loss (extension: the reference documents the equations,
// particle_tracking/state.py:5-10,56-70, but refuses such runs, Output.py:153-155).
//   gravity   a += GM_m (r - r_m) / |r - r_m|^3      for every included moon
//   loss      rate += k0 exp(-((rho - rho0)/w)^2 - (z/H)^2) [* |v - Omega z^ x r| / (Omega rho0)]
// Moons move on prescribed circles in the planet's equatorial (x, y) plane.  The orbital phase at
// stage n of step k is theta_k + delta_n with theta_k = phi - omega (t0 - k h) and
// delta_n = omega c_n h; the host tabulates (cos, sin) theta_k per step (base[step][moon][2], two
// doubles per moon per step from L2) and (cos, sin) delta_n per stage (here), and the position is
// formed with the angle-addition formulas -- the oracles do exactly the same:
//   r_m = a (-(S cd + C sd), C cd - S sd, 0)
constexpr int NXC_DEV_MAX_MOONS = 4;   // == NXC_MAX_MOONS of the C ABI
struct BodyK {
    int n_moons, chx_on, chx_vel, pad_;
    double gm[NXC_DEV_MAX_MOONS], rad2[NXC_DEV_MAX_MOONS], a[NXC_DEV_MAX_MOONS];
    double cd[6][NXC_DEV_MAX_MOONS], sd[6][NXC_DEV_MAX_MOONS];
    double chx_k0, chx_rho0, chx_inv_w, chx_inv_h, chx_omega, chx_inv_v0;
};

NXC_DEV void moon_position(const BodyK &Bd, const double *__restrict__ base, int m, int stage,
                           double &mx, double &my)
{
    const double C = base[2 * m], S = base[2 * m + 1];
    const double cd = Bd.cd[stage][m], sd = Bd.sd[stage][m];
    const double sn = S * cd + C * sd;
    const double cs = C * cd - S * sd;
    mx = -(Bd.a[m] * sn);
    my = Bd.a[m] * cs;
}

// Moon gravity and torus loss added to state_eval's result.  base: (cos, sin) theta_k of this
// step per moon.  Generic-range sqrt / division: moons are approached closely.
NXC_DEV void bodies_eval(const BodyK &Bd, const double *__restrict__ base, int stage, double x,
                         double y, double z, double vx, double vy, double vz, double &ax,
                         double &ay, double &az, double &ion)
{
    for (int m = 0; m < Bd.n_moons; m++) {
        double mx, my;
        moon_position(Bd, base, m, stage, mx, my);
        const double dx = x - mx, dy = y - my;
        const double r3 = nxc_cube(nxc_sqrt((dx * dx + dy * dy) + z * z));
        const double g = Bd.gm[m];
        double qx, qy, qz;
        nxc_div3(g * dx, g * dy, g * z, r3, qx, qy, qz);
        ax += qx;
        ay += qy;
        az += qz;
    }
    if (Bd.chx_on) {
        const double rho = nxc_sqrt(x * x + y * y);
        const double u = (rho - Bd.chx_rho0) * Bd.chx_inv_w, w = z * Bd.chx_inv_h;
        double rate = Bd.chx_k0 * nxc_exp(-(u * u + w * w));
        if (Bd.chx_vel) {
            const double ux = vx + Bd.chx_omega * y, uy = vy - Bd.chx_omega * x;
            rate = rate * (nxc_sqrt((ux * ux + uy * uy) + vz * vz) * Bd.chx_inv_v0);
        }
        ion += rate;
    }
}

// ---------------------------------------------------------------------------------------------
// Dormand-Prince tableau (rk5.py:5-18) and one step (rk5.py:21-54)
// ---------------------------------------------------------------------------------------------
struct Tableau {
    static constexpr double A[7][7] = {
        {0, 0, 0, 0, 0, 0, 0},
        {0.2, 0, 0, 0, 0, 0, 0},
        {3. / 40., 9. / 40., 0, 0, 0, 0, 0},
        {44. / 45., -56. / 15., 32. / 9., 0, 0, 0, 0},
        {19372. / 6561., -25360. / 2187., 64448. / 6561., -212. / 729., 0, 0, 0},
        {9017. / 3168., -355. / 33., 46732. / 5247., 49. / 176., -5103. / 18656., 0, 0},
        {35. / 384., 0., 500. / 1113., 125. / 192., -2187. / 6784., 11. / 84., 0.}};
    static constexpr double B5[7] = {35. / 384., 0., 500. / 1113., 125. / 192., -2187. / 6784.,
                                     11. / 84., 0.};
    static constexpr double B4[7] = {5179. / 57600., 0., 7571. / 16695., 393. / 640.,
                                     -92097. / 339200., 187. / 2100., 1. / 40.};
};

// Precomputed h*a[n+1][i] for a launch-uniform step (the same IEEE products NumPy forms per packet,
// rk5.py:33); index n(n+1)/2 + i.  Kept in scalar registers.
struct StepW {
    double h;
    double w[21];
};

// One term of a tableau sum, acc + w k.  NumPy rounds the product and the sum (rk5.py:33-35: two
// roundings per term); here -- and, call for call, in the C oracle (oracle/c/oracle.c: rk5_body) --
// the term is ONE fused multiply-add, i.e. the exact product enters the sum and the result is
// rounded once: at most half an ulp of the sum closer to the true value per term, 98 fewer fp64
// instructions per step (14 terms x 7 components).  -DNXC_TABLEAU_TWO_ROUNDINGS (with the oracle's
// -DORACLE_TABLEAU_TWO_ROUNDINGS) restores NumPy's roundings.
NXC_DEV double nxc_tab_fma(double w, double k, double acc)
{
#ifdef NXC_TABLEAU_TWO_ROUNDINGS
    return acc + w * k;
#else
    return __builtin_fma(w, k, acc);
#endif
}

// s[8] = t_remaining, x, y, z, vx, vy, vz, frac (in/out).  d[8] (DELTA only) = the reference's
// error estimate |h * sum_{i<6} (B5-B4)_i k_i| (rk5.py:38-46; the 7th stage is left out there).
// Each stage is accumulated from zero in the order i = 0..n with terms (h*a)*k and the initial
// state added last (rk5.py:32-36); frac is carried as log(frac) (rk5.py:25,35,50).
template <bool DELTA, bool UNIFORM_H, bool FULL = false, bool NBODY = false>
NXC_DEV void rk5_step(const ForceK &F, const LutView &T, double (&s)[8], double h,
                      const StepW &W, double (&d)[8], const BodyK *Bd = nullptr,
                      const double *__restrict__ mp = nullptr)
{
    if (UNIFORM_H) h = W.h;
    double kv[6][3], ka[6][3], kl[6];
    const double x0 = s[1], y0 = s[2], z0 = s[3], vx0 = s[4], vy0 = s[5], vz0 = s[6];
    const double lf0 = nxc_log(s[7]);
    double px = x0, py = y0, pz = z0, vx = vx0, vy = vy0, vz = vz0, lf = lf0;
#pragma unroll
    for (int n = 0; n < 6; n++) {
        kv[n][0] = vx; kv[n][1] = vy; kv[n][2] = vz;
        state_eval<FULL>(F, T, px, py, pz, vy, ka[n][0], ka[n][1], ka[n][2], kl[n]);
        if (NBODY)
            bodies_eval(*Bd, mp, n, px, py, pz, vx, vy, vz, ka[n][0], ka[n][1], ka[n][2], kl[n]);
        // The reference starts each sum from 0.0 (0 + t0): dropped, it can only change the sign
        // of an exactly-zero sum.
        double nx, ny, nz, nvx, nvy, nvz, nlf;
        {
            const double w = UNIFORM_H ? W.w[n * (n + 1) / 2] : h * Tableau::A[n + 1][0];
            nx = w * kv[0][0]; ny = w * kv[0][1]; nz = w * kv[0][2];
            nvx = w * ka[0][0]; nvy = w * ka[0][1]; nvz = w * ka[0][2];
            nlf = -(w * kl[0]);
        }
#pragma unroll
        for (int i = 1; i <= n; i++) {
            if (Tableau::A[n + 1][i] == 0.0) continue;   // b5[1] = 0: the term is a zero
            const double w = UNIFORM_H ? W.w[n * (n + 1) / 2 + i] : h * Tableau::A[n + 1][i];
            nx = nxc_tab_fma(w, kv[i][0], nx);
            ny = nxc_tab_fma(w, kv[i][1], ny);
            nz = nxc_tab_fma(w, kv[i][2], nz);
            nvx = nxc_tab_fma(w, ka[i][0], nvx);
            nvy = nxc_tab_fma(w, ka[i][1], nvy);
            nvz = nxc_tab_fma(w, ka[i][2], nvz);
            nlf = nxc_tab_fma(-w, kl[i], nlf);
        }
        px = nx + x0; py = ny + y0; pz = nz + z0;
        vx = nvx + vx0; vy = nvy + vy0; vz = nvz + vz0;
        lf = nlf + lf0;
    }
    if (DELTA) {
        double e[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double bd = Tableau::B5[i] - Tableau::B4[i];
            e[0] = nxc_tab_fma(bd, kv[i][0], e[0]);
            e[1] = nxc_tab_fma(bd, kv[i][1], e[1]);
            e[2] = nxc_tab_fma(bd, kv[i][2], e[2]);
            e[3] = nxc_tab_fma(bd, ka[i][0], e[3]);
            e[4] = nxc_tab_fma(bd, ka[i][1], e[4]);
            e[5] = nxc_tab_fma(bd, ka[i][2], e[5]);
            e[6] = nxc_tab_fma(bd, kl[i], e[6]);
        }
        d[0] = 0.0;
#pragma unroll
        for (int c = 0; c < 7; c++) d[c + 1] = __builtin_fabs(h * e[c]);
    }
    s[0] = (-h) + s[0];            // -h*c[6] + t, c[6] = 1 (rk5.py:31,36)
    s[1] = px; s[2] = py; s[3] = pz; s[4] = vx; s[5] = vy; s[6] = vz;
    s[7] = nxc_exp(lf);
}

// ---- launch constants kept in LDS ----------------------------------------------------------------
struct ImageK {            // kernel-argument scalars of nxc_image_desc
    double M[9];
    double vrplanet, apix_cm2;
    int quantity, n_lines, downcast_f32, dbg;   // dbg: timing experiments only (0 = normal)
    int nx, nz;
    int x_is_x, pad_;     // M = [[1,0,0],[0,*,*],[0,*,*]]: the observer's x axis is the Sun frame's
                          // (every sub-observer point on the x = 0 meridian, the default included)
    double x_lo, x_inv_step, z_lo, z_inv_step;   // only to seed the edge search
    int64_t xedges_off, zedges_off;              // byte offsets of the edge arrays in the blob
    LutDesc line[4];
};

// Launch-uniform constants that are read once per sample / once per step live at offset 0 of the
// LDS block instead of in scalar registers: together with the force constants they would exceed
// the 102 SGPRs of a wave, and every spilled SGPR costs v_readlane/v_writelane VALU slots in a
// VALU-bound kernel.  (StepW is defined above; ImageK here.)
// Surface re-emission constants (particle_tracking/bouncepackets.py; read only when a packet
// hits the surface).  tx/ty/coef: knots and coefficients of the bicubic spline v(T, p) in global
// memory (FITPACK layout: coef[(nx-4) x (ny-4)] row-major).
struct BounceK {
    double GM, unit_km, accom, stickcoef, A0, A1, A2, t0, t1;
    int temp_dependent, nx, ny, pad_;
    unsigned long long seed;
    const double *tx, *ty, *coef;
};

// Kernel arguments the persistent loop touches only rarely (a chunk claim per 64 packets, a final
// state per packet).  The kernel copies them here before its loop and reads them back at the
// point of use, so that they do not occupy scalar registers across the whole loop.
struct LoopK {
    const double *soa0;
    const unsigned *order;
    double *final_out;
    long long *steps_out;
    unsigned long long *head;
    long long n;
    const long long *offsets;     // ROWS pass: row offsets per packet index
    const unsigned long long *avail;   // streamed upload: queue positions published so far (null: all)
    // layout of soa0: doubles between consecutive queue positions / between a packet's columns.
    // The ordered queue holds one 64-byte record per packet (8, 1); unordered packets are read
    // where they were uploaded, as columns (1, n).
    long long q_rec, q_col;
};
// Refined reciprocals of the two launch-constant divisors of the weight (1e6, Apix), computed once
// per workgroup; read from LDS by the samples that fall inside the image.
struct WeightK {
    double rs_1e6, rs_apix;
};

constexpr int NXC_LOG_BINS = 91;
struct LdsHeader {
    ImageK G;
    StepW W;
    BounceK B;
    BodyK Bd;
    LoopK L;
    WeightK Wt;
    // table of nxc_log: {1/c, -ln(1/c) high, low, 0} per bin, 32-byte rows (one ds_read_b128 +
    // one ds_read_b64 per lookup); filled by the host from nxc_log_table.hpp
    alignas(32) double logtab[NXC_LOG_BINS][4];
};
constexpr int NXC_HEADER_BYTES = (int)((sizeof(LdsHeader) + 31) & ~size_t(31));

NXC_DEV const LdsHeader &lds_header()
{
    return *reinterpret_cast<const LdsHeader *>(nxc_lds);
}

// log(x), table-driven and division-free.  x = 2^k m with m in [OFF, 2 OFF), OFF = 181/256; bin
// i = floor((m - OFF) * 128) has the centre c_i (bins 36..38 share c = 1); r = m / c_i - 1 comes
// out of ONE fma with the tabulated 1/c_i (|r| <= 3/256), and
//   log x = (k LN2_HI - ln(1/c_i)_hi) + r  [exact sum + its rounding error]
//           + k LN2_LO - ln(1/c_i)_lo + r^2 (-1/2 + r p(r)),   p = Taylor through r^10 / 10.
// k LN2_HI + lchi is exact (LN2_HI has 21 trailing zero bits, lchi is a multiple of 2^-42).
// Largest error seen on 8e6 arguments (all magnitudes, dense around 1): 0.71 ulp, mean 0.25; the
// C oracle runs the same operations on its own copy of the table.  Round 1 used fdlibm's
// s = f / (2 + f) form: 88 issue slots with its division against 34 here.
NXC_DEV double nxc_log(double x)
{
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    constexpr double OFF = 0.70703125;                      // 181/256, high word 0x3FE6A000
    int k = 0;
    // one test sends NaN, zero, negatives, infinity and subnormals to the rare path
    if (__builtin_expect(!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308), 0)) {
        if (x != x) return x;
        if (x == 0.0) return -__builtin_huge_val();
        if (x < 0.0) return __builtin_nan("");
        if (x == __builtin_huge_val()) return x;
        x *= 18014398509481984.0; k = -54;          // subnormal
    }
    const int hx = __double2hiint(x);
    const int tmp = hx - 0x3FE6A000;                         // the low word of OFF is zero
    k += tmp >> 20;
    const double m = __hiloint2double(hx - (int)((unsigned)tmp & 0xFFF00000u), __double2loint(x));
    const int i = (int)((m - OFF) * 128.0);                  // 0..90, exact
    const int row = (int)offsetof(LdsHeader, logtab) + 32 * i;
    const double2 t01 = lds_f64x2(row);                      // {1/c, -ln(1/c) high}
    const double lclo = lds_f64(row + 16);
    const double r = __builtin_fma(m, t01.x, -1.0);
    const double kd = (double)k;
    const double w = __builtin_fma(kd, LN2_HI, t01.y);       // exact
    const double hi = w + r;
    const double lo = ((w - hi) + r) + __builtin_fma(kd, LN2_LO, lclo);
    double p = -0x1.999999999999ap-4;                        // -1/10
    p = __builtin_fma(p, r, nxc_sconst(0x1.c71c71c71c71cp-4));           //  1/9
    p = __builtin_fma(p, r, nxc_sconst(-0x1.0p-3));                      // -1/8
    p = __builtin_fma(p, r, nxc_sconst(0x1.2492492492492p-3));           //  1/7
    p = __builtin_fma(p, r, nxc_sconst(-0x1.5555555555555p-3));          // -1/6
    p = __builtin_fma(p, r, nxc_sconst(0x1.999999999999ap-3));           //  1/5
    p = __builtin_fma(p, r, nxc_sconst(-0x1.0p-2));                      // -1/4
    p = __builtin_fma(p, r, nxc_sconst(0x1.5555555555555p-2));           //  1/3
    return __builtin_fma(r * r, __builtin_fma(r, p, -0.5), lo) + hi;
}
NXC_DEV LdsHeader &lds_header_rw()
{
    return *reinterpret_cast<LdsHeader *>(nxc_lds);
}

// ---------------------------------------------------------------------------------------------
// Surface re-emission: particle_tracking/bouncepackets.py:5-100, SurfaceInteraction.py:10-61,
// initial_state/surface_temperature.py:4-19.  Uniforms come from Philox keyed by (seed; packet
// id, bounce number): the reference draws them from its sequential generator, so parity with
// it is statistical; parity with the oracle's restatement (same uniforms) is to libm rounding.
// ---------------------------------------------------------------------------------------------
NXC_DEV void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                           unsigned k1, unsigned (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
        const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// two uniform doubles in [0,1): ((a << 32 | b) >> 11) * 2^-53
NXC_DEV void philox_pair(unsigned long long index, unsigned block, unsigned stream,
                         unsigned long long seed, double &u0, double &u1)
{
    unsigned r[4];
    philox4x32_10((unsigned)index, (unsigned)(index >> 32), block, stream, (unsigned)seed,
                  (unsigned)(seed >> 32), r);
    u0 = (double)((((unsigned long long)r[0] << 32) | r[1]) >> 11) * 0x1p-53;
    u1 = (double)((((unsigned long long)r[2] << 32) | r[3]) >> 11) * 0x1p-53;
}

constexpr unsigned NXC_STREAM_SOURCE = 0x5a0u;
constexpr unsigned NXC_STREAM_BOUNCE = 0xb0cu;

// The four non-zero cubic B-spline basis values at x in knot interval l (FITPACK fpbspl).
NXC_DEV void bspline_basis3(const double *__restrict__ t, int l, double x, double (&h)[4])
{
    h[0] = 1.0; h[1] = h[2] = h[3] = 0.0;
#pragma unroll
    for (int j = 1; j <= 3; j++) {
        double hh[3];
#pragma unroll
        for (int i = 0; i < 3; i++) hh[i] = h[i];
        h[0] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (i < j) {
                const int li = l + i + 1, lj = li - j;
                const double f = hh[i] / (t[li] - t[lj]);
                h[i] += f * (t[li] - x);
                h[i + 1] = f * (x - t[lj]);
            }
        }
    }
}

NXC_DEV int knot_interval(const double *__restrict__ t, int n, double x)
{
    int lo = 3, hi = n - 4;                     // t[lo] <= x < t[hi] (x clamped to [t[3], t[n-4]])
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (x >= t[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

// scipy RectBivariateSpline(...).ev(x, y) for kx = ky = 3 (FITPACK bispev)
NXC_DEV double bispev3(const BounceK &B, double x, double y)
{
    x = __builtin_fmin(__builtin_fmax(x, B.tx[3]), B.tx[B.nx - 4]);
    y = __builtin_fmin(__builtin_fmax(y, B.ty[3]), B.ty[B.ny - 4]);
    const int l = knot_interval(B.tx, B.nx, x), m = knot_interval(B.ty, B.ny, y);
    double hx[4], hy[4];
    bspline_basis3(B.tx, l, x, hx);
    bspline_basis3(B.ty, m, y, hy);
    const int ncy = B.ny - 4;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) s += B.coef[(l - 3 + i) * ncy + (m - 3 + j)] * hx[i] * hy[j];
    return s;
}

// One impact: move the packet back to the surface along its velocity, re-emit it.
NXC_DEV void bounce_packet(const BounceK &B, double (&s)[8], double r2, unsigned long long id,
                           int &nbounce)
{
    const double TWO_PI = 6.283185307179586;
    double x = s[1], y = s[2], z = s[3];
    const double vx = s[4], vy = s[5], vz = s[6];
    const double a = (vx * vx + vy * vy) + vz * vz;                   // :45-52
    const double b = 2 * ((x * vx + y * vy) + z * vz);
    const double c = r2 - 1.;
    const double sq = nxc_sqrt(b * b - 4 * a * c);
    const double t = __builtin_fmin((-b - sq) / (2 * a), (-b + sq) / (2 * a));
    x = x + vx * t; y = y + vy * t; z = z + vz * t;                     // :55
    const double r = nxc_sqrt(r2);
    double v_old2 = a + 2 * B.GM * (1. / r - 1);                        // :59-61
    v_old2 = v_old2 < 0 ? 0. : v_old2;
    double u_alt, u_az, u_p, unused;
    philox_pair(id, 2u * (unsigned)nbounce, NXC_STREAM_BOUNCE, B.seed, u_alt, u_az);
    philox_pair(id, 2u * (unsigned)nbounce + 1u, NXC_STREAM_BOUNCE, B.seed, u_p, unused);
    nbounce++;
    const double alt = asin(u_alt), az = TWO_PI * u_az;                 // :9-18
    const double v_rad = sin(alt), v_t0 = cos(alt) * cos(az), v_t1 = cos(alt) * sin(az);
    const double rn = nxc_sqrt((x * x + y * y) + z * z);                // :23-33
    const double en = nxc_sqrt(y * y + x * x);
    const double n0 = -z * x, n1 = -z * y, n2 = x * x + y * y;
    const double nn = nxc_sqrt((n0 * n0 + n1 * n1) + n2 * n2);
    const double dx = (v_t0 * (n0 / nn) + v_t1 * (y / en)) + v_rad * (x / rn);
    const double dy = (v_t0 * (n1 / nn) + v_t1 * (-x / en)) + v_rad * (y / rn);
    const double dz = (v_t0 * (n2 / nn) + v_t1 * 0.0) + v_rad * (z / rn);
    const double lonhit = fmod(atan2(x, -y) + TWO_PI, TWO_PI);          // :68-69
    const double lathit = asin(z);
    double tsurf = B.t0;                                                // surface_temperature.py:12-17
    if (lonhit <= 1.5707963267948966 || lonhit >= 4.71238898038469)
        tsurf = B.t0 + B.t1 * nxc_sqrt(nxc_sqrt(__builtin_fabs(cos(lonhit) * cos(lathit))));
    double v_new;
    if (B.accom == 0.0) {
        v_new = nxc_sqrt(v_old2);                                       // :65-66
    } else {
        const double v_emit = bispev3(B, tsurf, u_p) / B.unit_km;       // :71-75
        v_new = nxc_sqrt(v_emit * v_emit * B.accom + v_old2 * (1 - B.accom));   // :77-78
    }
    s[1] = x; s[2] = y; s[3] = z;
    s[4] = dx * v_new; s[5] = dy * v_new; s[6] = dz * v_new;            // :80
    if (B.temp_dependent) {                                             // :83-89, SurfaceInteraction.py:13-20
        double st = B.A0 * exp(B.A1 * tsurf) + B.A2;
        st = st > 1. ? 1. : (st < 0. ? 0. : st);
        s[7] *= (1 - st);
    } else if (B.stickcoef > 0) {
        s[7] *= (1 - B.stickcoef);                                      // :92-93
    }
}

// Post-step tests with stickcoef == 1.  Constant driver: Output.py:395-416 (r = |x|); variable
// driver: Output.py:308-324, which compares r^2 with 1 AND with outeredge (reference quirk).
// (sqrt(r2) - 1) < 0  <=>  r2 < 1 for a correctly rounded sqrt.
// `edge2` is the host-computed threshold on r^2: for the constant driver the largest double whose
// correctly rounded square root is <= outeredge (so r2 > edge2  <=>  sqrt(r2) > outeredge), for
// the variable driver outeredge itself.
// With BOUNCE a packet that hits the surface is re-emitted instead of absorbed (Output.py:398-402);
// the escape test still uses the pre-impact radius, as the reference's tempR does.
template <bool BOUNCE, bool NBODY = false>
NXC_DEV void apply_fate(double (&s)[8], double edge2, unsigned long long id, int &nbounce,
                        const BodyK *Bd = nullptr, const double *__restrict__ base = nullptr)
{
    const double r2 = (s[1] * s[1] + s[2] * s[2]) + s[3] * s[3];
    if (r2 < 1.0) {
        if (BOUNCE) bounce_packet(lds_header().B, s, r2, id, nbounce);
        else s[7] = 0.0;
    }
    if (r2 > edge2) s[7] = 0.0;
    if (NBODY) {                    // absorbed by a moon (positions at the end of the step)
        for (int m = 0; m < Bd->n_moons; m++) {
            double mx, my;
            moon_position(*Bd, base, m, 5, mx, my);
            const double dx = s[1] - mx, dy = s[2] - my;
            if ((dx * dx + dy * dy) + s[3] * s[3] < Bd->rad2[m]) s[7] = 0.0;
        }
    }
    if (s[7] < 1e-10) s[7] = 0.0;
    if (s[7] == 0.0) s[0] = 0.0;
}

// ---------------------------------------------------------------------------------------------
// Image: data_simulation/ModelImage.py:242-269, ModelResult.py:140-170, math/histogram.py:32-36
// ---------------------------------------------------------------------------------------------

// Values read from LDS are wave-uniform but land in vector registers; readfirstlane moves them
// to scalar registers so that they do not add to the VGPR pressure of the step loop.
NXC_DEV double wave_uniform(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
    const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
NXC_DEV int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Launch constants of the image path that are worth a register: read from the LDS header once
// per thread before the step loop (the compiler cannot hoist LDS loads over the loop's LDS
// stores).  The two refined reciprocals serve the per-sample divisions by 1e6 and by Apix.
struct ImageRegs {
    double vrplanet;
    double x_lo, x_hi, x_inv_step, z_lo, z_hi, z_inv_step;
    int xedges, zedges, nx, nz, quantity, n_lines, downcast, dbg, x_is_x;
};

NXC_DEV ImageRegs image_regs(const ImageK &G)
{
    ImageRegs R;
    R.vrplanet = G.vrplanet;
    R.xedges = (int)G.xedges_off; R.zedges = (int)G.zedges_off;
    R.nx = G.nx; R.nz = G.nz;
    R.x_lo = lds_f64(R.xedges); R.x_hi = lds_f64(R.xedges + 8 * R.nx);
    R.z_lo = lds_f64(R.zedges); R.z_hi = lds_f64(R.zedges + 8 * R.nz);
    R.x_inv_step = G.x_inv_step; R.z_inv_step = G.z_inv_step;
    R.quantity = G.quantity; R.n_lines = G.n_lines; R.downcast = G.downcast_f32;
    R.x_is_x = wave_uniform(G.x_is_x);
#ifdef NXC_EXPERIMENT_KNOBS     // tools/ timing experiments only: the product build has no such switch
    R.dbg = G.dbg;
#else
    R.dbg = 0;
#endif
    R.vrplanet = wave_uniform(R.vrplanet);
    R.x_lo = wave_uniform(R.x_lo); R.x_hi = wave_uniform(R.x_hi);
    R.z_lo = wave_uniform(R.z_lo); R.z_hi = wave_uniform(R.z_hi);
    R.x_inv_step = wave_uniform(R.x_inv_step); R.z_inv_step = wave_uniform(R.z_inv_step);
    R.xedges = wave_uniform(R.xedges); R.zedges = wave_uniform(R.zedges);
    R.nx = wave_uniform(R.nx); R.nz = wave_uniform(R.nz);
    R.quantity = wave_uniform(R.quantity); R.n_lines = wave_uniform(R.n_lines);
    R.downcast = wave_uniform(R.downcast);
#ifdef NXC_EXPERIMENT_KNOBS
    R.dbg = wave_uniform(R.dbg);
#endif
    return R;
}

// n/d for a launch-constant d: same bits as nxc_div(n, d).  y = nxc_recip_seed(d), or NaN when d
// is outside the middle exponent range (decided once per launch in stage_tables).
NXC_DEV double nxc_div_const(double n, double d, double y)
{
    if (!((nxc_mid_range(n) || n == 0.0) && y == y)) return n / d;
    return nxc_div_seeded(n, d, y);
}

// np.histogram2d bin along one axis: searchsorted(edges, v, 'right') - 1, the right-most edge
// folded into the last bin, everything else (NaN included) outside = -1.  The arithmetic guess is
// verified against the very edge values np.linspace produced (staged in LDS, one ds_read2_b64);
// only a sample within rounding of an edge takes the walk.
NXC_DEV int bin_index(double v, int edges, int n, double lo, double hi, double inv_step)
{
    if (!(v >= lo) || !(v <= hi)) return -1;
    int k = (int)((v - lo) * inv_step);
    k = k > n - 1 ? n - 1 : k;
    const double e0 = lds_f64(edges + 8 * k), e1 = lds_f64(edges + 8 * k + 8);
    if (__builtin_expect(!((v >= e0) && (v < e1)), 0)) {
        while (k > 0 && v < lds_f64(edges + 8 * k)) --k;
        while (k < n - 1 && v >= lds_f64(edges + 8 * (k + 1))) ++k;
    }
    return k;
}

NXC_DEV double f32_round_trip(double v) { return (double)(float)v; }

// Image accumulator = one interleaved fp64 array acc2[2*pix] = weight sum, acc2[2*pix + 1] = packet
// count (integers are exact in fp64 up to 2^53), accumulated with global_atomic_add_f64.
//
// Atomics execute at the memory side, one request per 64-byte line touched by a wave instruction
// (tools/ubench_atomics.hip: 24 G scattered lane-atomics/s, but 47 G/s when lanes l and l+32 add
// to the two halves of one 16-byte record).  So a pixel's weight and count are added by two lanes
// of the SAME instruction: in the first instruction lanes 0..31 add their own weight while lanes
// 32..63 add the count of their partner lane (l - 32); the second instruction serves the upper
// half's samples the other way round.  One request per binned sample instead of two.
// value held by lane l ^ 32 (v_permlane32_swap: one VALU instruction, no LDS crossbar trip)
NXC_DEV int half_swap(int v, bool upper)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return upper ? (int)r[0] : (int)r[1];
}

// Wave-cooperative: all 64 lanes call it from uniform control flow; `has`: this lane holds a
// sample for pixel `pix` with weight `w` (a zero weight adds only the count).
NXC_DEV void image_add_pairs(bool has, int pix, double w, double *__restrict__ acc2)
{
    if (__ballot(has) == 0) return;
    const bool upper = (threadIdx.x & 32) != 0;
    const int ppix = half_swap(has ? pix : -1, upper);        // partner lane's pixel, -1 = none
    const bool own = has && w != 0.0;
    const bool partner = ppix >= 0;
    {   // samples of lanes 0..31
        const bool act = upper ? partner : own;
        if (act)
            unsafeAtomicAdd(&acc2[2ll * (upper ? ppix : pix) + (upper ? 1 : 0)], upper ? 1.0 : w);
    }
    {   // samples of lanes 32..63
        const bool act = upper ? own : partner;
        if (act)
            unsafeAtomicAdd(&acc2[2ll * (upper ? pix : ppix) + (upper ? 0 : 1)], upper ? w : 1.0);
    }
}

// The image work of one stored sample, in two stages so that the persistent kernel can put a
// compaction queue between them (about half the samples of the bench workload fall outside the
// image; the second stage -- g-value lookups, divisions, atomics -- then runs with full waves):
//   image_locate: float32 round trip, rotation, bins, occultation and shadow masks
//                 (ModelImage.py:242-258).  Returns the pixel (ix*nz + iz) or -1, the radial
//                 velocity for the g-value lookup and the masked fraction fw.
//   image_weight: weight of a located sample (ModelResult.py:148-161, ModelImage.py:262); false
//                 when it is not finite (the reference asserts, ModelResult.py:170; here the
//                 sample is counted in `nonfinite` and never reaches a pixel).
// (ix, iz as two numbers for the tiled image, which files a sample under its row's tile)
NXC_DEV bool image_locate_core_xz(const ImageK &G, const ImageRegs &R, double x, double y, double z,
                                  double vy, double frac, double &radvel_out, double &fw_out,
                                  unsigned long long &nonfinite, int &ix_out, int &iz_out)
{
    const double radvel = vy + R.vrplanet;                         // ModelImage.py:242-243
    double xo, yo, zo;                                             // ModelImage.py:249
    if (R.x_is_x) {
        // rows (1,0,0), (0,a,b), (0,c,d): the products with 1 and 0 are exact and adding a zero
        // changes at most the sign of a zero, which nothing below can see
        xo = x;
        yo = G.M[4] * y + G.M[5] * z;
        zo = G.M[7] * y + G.M[8] * z;
    } else {
        xo = (G.M[0] * x + G.M[1] * y) + G.M[2] * z;
        yo = (G.M[3] * x + G.M[4] * y) + G.M[5] * z;
        zo = (G.M[6] * x + G.M[7] * y) + G.M[8] * z;
    }
    const int ix = bin_index(xo, R.xedges, R.nx, R.x_lo, R.x_hi, R.x_inv_step);
    const int iz = bin_index(zo, R.zedges, R.nz, R.z_lo, R.z_hi, R.z_inv_step);
    if (ix < 0 || iz < 0) {
        // outside the image the weight is not formed; it is finite iff frac and radvel are
        if (!(__builtin_fabs(frac) <= 1.7976931348623157e308) || radvel != radvel) nonfinite++;
        return false;
    }
    const double s_obs = xo * xo + zo * zo;                        // ModelImage.py:252-254
    const bool inview = (s_obs > 0x1.0000000000001p+0) || (yo < 0.0);
    frac = inview ? frac : frac * 0.0;
    if (R.quantity != 0)                                           // ModelImage.py:257-258
        frac = sunlit(x, y, z) ? frac : frac * 0.0;                //   * out_of_shadow
    radvel_out = radvel;
    fw_out = frac;
    ix_out = ix; iz_out = iz;
    return true;
}

NXC_DEV int image_locate_core(const ImageK &G, const ImageRegs &R, double x, double y, double z,
                              double vy, double frac, double &radvel_out, double &fw_out,
                              unsigned long long &nonfinite)
{
    int ix = 0, iz = 0;
    if (!image_locate_core_xz(G, R, x, y, z, vy, frac, radvel_out, fw_out, nonfinite, ix, iz))
        return -1;
    return ix * R.nz + iz;
}

NXC_DEV int image_locate(const ImageK &G, const ImageRegs &R, double x, double y, double z,
                         double vy, double frac, double &radvel_out, double &fw_out,
                         unsigned long long &nonfinite)
{
    if (R.downcast) {
        x = f32_round_trip(x); y = f32_round_trip(y); z = f32_round_trip(z);
        vy = f32_round_trip(vy); frac = f32_round_trip(frac);
    }
    return image_locate_core(G, R, x, y, z, vy, frac, radvel_out, fw_out, nonfinite);
}

// The down-cast image path in the persistent kernel, stage 0: is the sample inside the image
// frame at all?  Only the float32 round trip of the position, the two observer-frame coordinates
// the bins are taken from and the range test of bin_index -- exactly the samples image_locate
// returns -1 for are turned away here, with the same bookkeeping: outside the image the weight is
// not formed, and it is finite iff the stored (float32) frac and vy are, i.e. iff |frac| is below
// the value that rounds to float infinity (2^128 - 2^103) and vy is not NaN (vrplanet is finite:
// nxc_set_image).  The sample then waits in the queue as the five float32 values save() would
// store; the rest (image_locate_core, image_weight) runs on full waves after the queue.
NXC_DEV bool image_frame_test(const ImageK &G, const ImageRegs &R, double x, double y, double z,
                              double vy, double frac, float &xf, float &yf, float &zf,
                              unsigned long long &nonfinite)
{
    xf = (float)x; yf = (float)y; zf = (float)z;
    const double xd = xf, yd = yf, zd = zf;
    double xo, zo;
    if (R.x_is_x) {
        xo = xd;
        zo = G.M[7] * yd + G.M[8] * zd;
    } else {
        xo = (G.M[0] * xd + G.M[1] * yd) + G.M[2] * zd;
        zo = (G.M[6] * xd + G.M[7] * yd) + G.M[8] * zd;
    }
    const bool inside = (xo >= R.x_lo) && (xo <= R.x_hi) && (zo >= R.z_lo) && (zo <= R.z_hi);
    if (!inside && (!(__builtin_fabs(frac) < 0x1.ffffffp127) || vy != vy)) nonfinite++;
    return inside;
}

NXC_DEV bool image_weight(const ImageK &G, const ImageRegs &R, double radvel, double fw,
                          double &w_out)
{
    double w = fw;                                                 // ModelResult.py:148-149
    if (R.quantity != 0) {                                         // ModelResult.py:150-161
        // the lines' lookups are issued together (cells, then rows, then the selects) so that
        // their LDS round trips overlap instead of running one table after the other
        double gg = 0.0;
        if (R.n_lines == 2) {
            const LutView v0 = lut_view(G.line[0]), v1 = lut_view(G.line[1]);
            LutProbe p0, p1;
            lut_probe_cell(v0, radvel, p0); lut_probe_cell(v1, radvel, p1);
            lut_probe_rows(v0, p0); lut_probe_rows(v1, p1);
            const double g0 = lut_finish(v0, p0);
            gg = g0 + lut_finish(v1, p1);
        } else {
            gg = R.n_lines > 0 ? lut_interp(lut_view(G.line[0]), radvel) : 0.0;
#pragma unroll
            for (int l = 1; l < 4; l++)
                if (l < R.n_lines) gg += lut_interp(lut_view(G.line[l]), radvel);
        }
        w = nxc_div_const(fw * gg, 1e6, lds_header().Wt.rs_1e6);
    }
    w = nxc_div_const(w, G.apix_cm2, lds_header().Wt.rs_apix);     // ModelImage.py:262
    w_out = w;
    return (__builtin_fabs(w) <= 1.7976931348623157e308) && radvel == radvel;   // :170
}

// locate + weight + add of one sample per lane, for the kernels that need no compaction.
// Wave-cooperative (all lanes call it; `has`: this lane offers a sample).
NXC_DEV void image_sample(const ImageK &G, const ImageRegs &R, bool has, double x, double y,
                          double z, double vy, double frac, double *__restrict__ acc2,
                          unsigned long long &binned, unsigned long long &nonfinite)
{
    int pix = -1;
    double radvel = 0.0, fw = 0.0, w = 0.0;
    if (has) pix = image_locate(G, R, x, y, z, vy, frac, radvel, fw, nonfinite);
    bool ok = pix >= 0;
    if (ok && !image_weight(G, R, radvel, fw, w)) { nonfinite++; ok = false; }
    binned += ok;
    image_add_pairs(ok, pix, w, acc2);
}

// Per-wave compaction queue between image_locate and image_weight: a ring of 128 located samples
// {pixel, radial velocity, masked fraction} in LDS.  push() appends the lanes' samples in lane
// order (ballot + prefix rank); once 64 are waiting, pop() hands one to every lane.  All calls
// are wave-uniform; head and tail are wave-uniform counters.
constexpr int NXC_IMGQ_SLOTS = 128;
constexpr int NXC_IMGQ_BYTES = NXC_IMGQ_SLOTS * (8 + 8 + 4);

struct ImageQueue {
    int head = 0, tail = 0;

    NXC_DEV void push(bool has, int pix, double radvel, double fw, int qoff)
    {
        const unsigned long long m = __ballot(has);
        if (m == 0) return;
        if (has) {
            const int lane = threadIdx.x & 63;
            const int slot = (tail + __popcll(m & ((1ull << lane) - 1ull))) & (NXC_IMGQ_SLOTS - 1);
            *reinterpret_cast<double *>(nxc_lds + qoff + 8 * slot) = radvel;
            *reinterpret_cast<double *>(nxc_lds + qoff + 8 * NXC_IMGQ_SLOTS + 8 * slot) = fw;
            *reinterpret_cast<int *>(nxc_lds + qoff + 16 * NXC_IMGQ_SLOTS + 4 * slot) = pix;
        }
        tail += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // the same ring holding the five float32 values of a sample that passed image_frame_test
    NXC_DEV void push_sample(bool has, float x, float y, float z, float vy, float frac, int qoff)
    {
        const unsigned long long m = __ballot(has);
        if (m == 0) return;
        if (has) {
            const int lane = threadIdx.x & 63;
            const int slot = (tail + __popcll(m & ((1ull << lane) - 1ull))) & (NXC_IMGQ_SLOTS - 1);
            float *q = reinterpret_cast<float *>(nxc_lds + qoff) + slot;
            q[0] = x; q[NXC_IMGQ_SLOTS] = y; q[2 * NXC_IMGQ_SLOTS] = z;
            q[3 * NXC_IMGQ_SLOTS] = vy; q[4 * NXC_IMGQ_SLOTS] = frac;
        }
        tail += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    NXC_DEV bool pop_sample(int qoff, double &x, double &y, double &z, double &vy, double &frac)
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n = waiting() < 64 ? waiting() : 64;
        const int lane = threadIdx.x & 63;
        const bool mine = lane < n;
        if (mine) {
            const int slot = (head + lane) & (NXC_IMGQ_SLOTS - 1);
            const float *q = reinterpret_cast<const float *>(nxc_lds + qoff) + slot;
            x = q[0]; y = q[NXC_IMGQ_SLOTS]; z = q[2 * NXC_IMGQ_SLOTS];
            vy = q[3 * NXC_IMGQ_SLOTS]; frac = q[4 * NXC_IMGQ_SLOTS];
        }
        head += n;
        __builtin_amdgcn_wave_barrier();
        return mine;
    }
    NXC_DEV int waiting() const { return tail - head; }
    // every lane below min(64, waiting) receives a sample; returns whether this lane did
    NXC_DEV bool pop(int qoff, int &pix, double &radvel, double &fw)
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n = waiting() < 64 ? waiting() : 64;
        const int lane = threadIdx.x & 63;
        const bool mine = lane < n;
        if (mine) {
            const int slot = (head + lane) & (NXC_IMGQ_SLOTS - 1);
            radvel = *reinterpret_cast<const double *>(nxc_lds + qoff + 8 * slot);
            fw = *reinterpret_cast<const double *>(nxc_lds + qoff + 8 * NXC_IMGQ_SLOTS + 8 * slot);
            pix = *reinterpret_cast<const int *>(nxc_lds + qoff + 16 * NXC_IMGQ_SLOTS + 4 * slot);
        }
        head += n;
        __builtin_amdgcn_wave_barrier();
        return mine;
    }
};
