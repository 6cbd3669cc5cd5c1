/* oracle.c -- plain-C CPU re-statement of nexoclom's particle_tracking + image hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Built into oracle/_build/liboracle.so by oracle/Makefile and loaded
 * (ctypes) only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
 * under nexoclom_amd/ links, loads or calls it.
 *
 * It follows the reference (paths under /root/reference/nexoclom/) operation by operation, in
 * the reference's floating-point order, compiled with -ffp-contract=off:
 *   ora_state            particle_tracking/state.py:17-74
 *   ora_rk5_step         particle_tracking/rk5.py:5-54
 *   ora_integrate_const  particle_tracking/Output.py:368-431 (one packet at a time; packets never
 *                        interact, so packet-major order gives the lock-step loop's results)
 *   ora_integrate_var    particle_tracking/Output.py:221-359
 *   ora_image            data_simulation/ModelImage.py:242-269, ModelResult.py:140-170,
 *                        math/histogram.py:32-36 (np.histogram2d binning rule)
 * Differences from NumPy are confined to pow/exp/log (see oracle_math.h) and are <= 1 ulp per
 * call; tests/test_oracle_c.py pins this file against oracle/np_oracle.py and the golden vectors
 * generated from the reference's own rk5.py/state.py.
 *
 * Packet arrays are struct-of-arrays: soa[c*n + i], c = 0..7 = t_remaining,x,y,z,vx,vy,vz,frac.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE            /* sincos */
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle_math.h"

typedef struct {
    double GM, vrplanet, photo, lifetime;
    int32_t gravity, radpres, has_photo, pad_;
    int64_t n_tab;
    const double *v_tab, *a_tab;
} ora_forces;

typedef struct {
    double M[9];
    double vrplanet, apix_cm2;
    int32_t quantity;      /* 0 = column/density, 1 = radiance/difrad */
    int32_t n_lines;
    int32_t downcast_f32;  /* emulate Output.save()/restore() float32 round trip */
    int32_t pad_;
    int64_t nx, nz;
    const double *xedges, *zedges;     /* np.linspace(lo, hi, n+1) */
    int64_t line_n[4];
    const double *line_v[4], *line_g[4];
} ora_image_desc;

/* --- Dormand-Prince tableau, rk5.py:5-18 ------------------------------------------------------ */
static const double CN[7] = {0, 0.2, 0.3, 0.8, 8./9., 1., 1.};
static const double B5[7] = {35./384., 0., 500./1113., 125./192., -2187./6784., 11./84., 0.};
static const double B4[7] = {5179./57600., 0., 7571./16695., 393./640., -92097./339200.,
                             187./2100., 1./40.};
static const double AT[7][7] = {
    {0},
    {0.2},
    {3./40., 9./40.},
    {44./45., -56./15., 32./9.},
    {19372./6561., -25360./2187., 64448./6561., -212./729.},
    {9017./3168., -355./33., 46732./5247., 49./176., -5103./18656.},
    {35./384., 0., 500./1113., 125./192., -2187./6784., 11./84., 0.}};

/* np.interp (numpy/_core/src/multiarray/compiled_base.c arr_interp) for a finite table */
static double interp1(double x, const double *xp, const double *fp, int64_t n)
{
    if (x != x) return x;
    if (x > xp[n-1]) return fp[n-1];
    if (x < xp[0]) return fp[0];
    int64_t lo = 0, hi = n;              /* invariant: xp[lo] <= x < xp[hi] (xp[n] = +inf) */
    while (hi - lo > 1) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (x >= xp[mid]) lo = mid; else hi = mid;
    }
    if (lo == n-1 || xp[lo] == x) return fp[lo];
    double slope = (fp[lo+1] - fp[lo]) / (xp[lo+1] - xp[lo]);
    return slope * (x - xp[lo]) + fp[lo];
}

static inline int sunlit(double x, double y, double z)
{
    double rho = sqrt(x*x + z*z);           /* norm(x[:, [1,3]]) */
    return (rho > 1.0) || (y < 0.0);
}

static inline void state1(const ora_forces *f, double x, double y, double z, double vy,
                          double *ax, double *ay, double *az, double *ion)
{
    double gx = 0.0, gy = 0.0, gz = 0.0;
    if (f->gravity) {
        double r3 = ora_cube(sqrt((x*x + y*y) + z*z));
        gx = f->GM * x / r3; gy = f->GM * y / r3; gz = f->GM * z / r3;
    }
    double ry = 0.0;
    if (f->radpres) {
        double vv = vy + f->vrplanet;
        ry = interp1(vv, f->v_tab, f->a_tab, f->n_tab) * (double)sunlit(x, y, z);
    }
    *ax = gx + 0.0; *ay = gy + ry; *az = gz + 0.0;
    if (f->lifetime > 0) *ion = 1.0 / f->lifetime;
    else if (f->has_photo) *ion = f->photo * (double)sunlit(x, y, z);
    else *ion = 0.0;
}

/* ---- Extension beyond the reference: moons + plasma-torus loss --------------------------------
 * The reference documents the equations (state.py:5-10, commented stub :56-70) and refuses such
 * runs (Output.py:153-155), so this part is OUR definition (include/nexoclom_hip.h,
 * nxc_bodies_desc) and is "parity unpinned" with respect to the reference. */
typedef struct {
    int32_t n_moons, chx_on;
    double gm[4], radius[4], a[4], omega[4], phi[4];
    double t0;
    double chx_k0, chx_rho0, chx_width, chx_height, chx_omega;
} ora_bodies;

static const double CSTAGE[6] = {0, 0.2, 0.3, 0.8, 8./9., 1.};

/* Phase at stage n of step k = theta_k + delta_n, theta_k = phi - omega (t0 - k h),
 * delta_n = omega c_n h; both sincos() pairs are formed separately and combined with the
 * angle-addition formulas (the definition of include/nexoclom_hip.h: the device gets theta_k from a
 * per-step table and delta_n from six constants).  sincos() explicitly: compilers merge sin + cos
 * into it at some optimisation levels, and its results are not always those of the separate
 * calls. */
static inline void moon_xy(const ora_bodies *b, int m, int64_t k, int stage, double h,
                           double *mx, double *my)
{
    double t = b->t0 - (double)k * h;
    double S, C, sd, cd;
    sincos(b->phi[m] - b->omega[m] * t, &S, &C);
    sincos(b->omega[m] * (CSTAGE[stage] * h), &sd, &cd);
    double sn = S * cd + C * sd;
    double cs = C * cd - S * sd;
    *mx = -(b->a[m] * sn);
    *my = b->a[m] * cs;
}

/* adds the moons' gravity and the torus loss to state1()'s result; st = stage state */
static inline void bodies1(const ora_bodies *b, int64_t k, int stage, double h, const double *st,
                           double *ax, double *ay, double *az, double *ion)
{
    double x = st[1], y = st[2], z = st[3];
    for (int m = 0; m < b->n_moons; m++) {
        double mx, my;
        moon_xy(b, m, k, stage, h, &mx, &my);
        double dx = x - mx, dy = y - my;
        double r3 = ora_cube(sqrt((dx*dx + dy*dy) + z*z));
        *ax += b->gm[m] * dx / r3;
        *ay += b->gm[m] * dy / r3;
        *az += b->gm[m] * z / r3;
    }
    if (b->chx_on) {
        double inv_w = 1.0 / b->chx_width, inv_h = 1.0 / b->chx_height;
        double rho = sqrt(x*x + y*y);
        double u = (rho - b->chx_rho0) * inv_w, w = z * inv_h;
        double rate = b->chx_k0 * ora_exp(-(u*u + w*w));
        if (b->chx_omega != 0) {
            double inv_v0 = 1.0 / (b->chx_omega * b->chx_rho0);
            double ux = st[4] + b->chx_omega * y, uy = st[5] - b->chx_omega * x;
            rate = rate * (sqrt((ux*ux + uy*uy) + st[6]*st[6]) * inv_v0);
        }
        *ion += rate;
    }
}

void ora_state(const ora_forces *f, int64_t n, const double *x, const double *y, const double *z,
               const double *vy, double *ax, double *ay, double *az, double *ion)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        state1(f, x[i], y[i], z[i], vy[i], &ax[i], &ay[i], &az[i], &ion[i]);
}

/* One term of a tableau sum, acc + w k (rk5.py:33-35,41-43).  NumPy rounds the product and the
 * sum; the kernels under test fuse the two (one rounding, nxc_device.hpp: nxc_tab_fma), and so does
 * this checker, call for call, so that the two stay bit-identical.  Against the reference's own
 * vectors that is at most half an ulp of the sum per term (tests/test_oracle_golden.py pins a step at
 * rtol 1e-13, the NumPy oracle keeps NumPy's roundings).  -DORACLE_TABLEAU_TWO_ROUNDINGS restores
 * them here (with -DNXC_TABLEAU_TWO_ROUNDINGS in the kernels). */
static inline double tab_fma(double w, double k, double acc)
{
#ifdef ORACLE_TABLEAU_TWO_ROUNDINGS
    return acc + w * k;
#else
    return fma(w, k, acc);
#endif
}

/* One step for one packet.  s[8] in/out; d[8] (nullable) receives |h * sum_{i<6} (b5-b4)_i k_i|. */
static void rk5_body(const ora_forces *f, const ora_bodies *b, int64_t k, double *s, double h,
                     double *d)
{
    double y0[8], st[8], kv[6][3], ka[6][3], kl[6];
    memcpy(y0, s, sizeof y0);
    y0[7] = ora_log(y0[7]);
    memcpy(st, y0, sizeof st);
    for (int n = 0; n < 6; n++) {
        kv[n][0] = st[4]; kv[n][1] = st[5]; kv[n][2] = st[6];
        state1(f, st[1], st[2], st[3], st[5], &ka[n][0], &ka[n][1], &ka[n][2], &kl[n]);
        if (b) bodies1(b, k, n, h, st, &ka[n][0], &ka[n][1], &ka[n][2], &kl[n]);
        double nx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        nx[0] = -h * CN[n+1];
        for (int i = 0; i <= n; i++) {
            double w = h * AT[n+1][i];
            for (int c = 0; c < 3; c++) {
                nx[1+c] = tab_fma(w, kv[i][c], nx[1+c]);
                nx[4+c] = tab_fma(w, ka[i][c], nx[4+c]);
            }
            nx[7] = tab_fma(-w, kl[i], nx[7]);
        }
        for (int c = 0; c < 8; c++) st[c] = nx[c] + y0[c];
    }
    if (d) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 6; i++) {
            double bd = B5[i] - B4[i];
            for (int c = 0; c < 3; c++) {
                acc[1+c] = tab_fma(bd, kv[i][c], acc[1+c]);
                acc[4+c] = tab_fma(bd, ka[i][c], acc[4+c]);
            }
            acc[7] = tab_fma(bd, kl[i], acc[7]);
        }
        for (int c = 0; c < 8; c++) d[c] = fabs(h * acc[c]);
    }
    st[7] = ora_exp(st[7]);
    memcpy(s, st, sizeof st);
}

static void rk5_one(const ora_forces *f, double *s, double h, double *d)
{
    rk5_body(f, NULL, 0, s, h, d);
}

void ora_rk5_step(const ora_forces *f, int64_t n, const double *in, const double *h, double *out,
                  double *delta)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double s[8], d[8];
        for (int c = 0; c < 8; c++) s[c] = in[c*n + i];
        rk5_one(f, s, h[i], delta ? d : NULL);
        for (int c = 0; c < 8; c++) out[c*n + i] = s[c];
        if (delta) for (int c = 0; c < 8; c++) delta[c*n + i] = d[c];
    }
}

/* Output.py:395-416 (constant: r) / :308-324 (variable: r^2 in both tests) */
static inline void fate(double *s, double outeredge, int r_squared)
{
    double r2 = (s[1]*s[1] + s[2]*s[2]) + s[3]*s[3];
    double rr = r_squared ? r2 : sqrt(r2);
    if (r_squared ? (rr < 1.0) : ((rr - 1.0) < 0.0)) s[7] = 0.0;
    if (rr > outeredge) s[7] = 0.0;
    if (s[7] < 1e-10) s[7] = 0.0;
    if (s[7] == 0.0) s[0] = 0.0;
}

/* constant-driver fate plus absorption by a moon (positions at the end of step k) */
static inline void fate_bodies(double *s, double outeredge, const ora_bodies *b, int64_t k, double h)
{
    double r2 = (s[1]*s[1] + s[2]*s[2]) + s[3]*s[3];
    double rr = sqrt(r2);
    if ((rr - 1.0) < 0.0) s[7] = 0.0;
    if (rr > outeredge) s[7] = 0.0;
    for (int m = 0; m < b->n_moons; m++) {
        double mx, my;
        moon_xy(b, m, k, 5, h, &mx, &my);
        double dx = s[1] - mx, dy = s[2] - my;
        if ((dx*dx + dy*dy) + s[3]*s[3] < b->radius[m] * b->radius[m]) s[7] = 0.0;
    }
    if (s[7] < 1e-10) s[7] = 0.0;
    if (s[7] == 0.0) s[0] = 0.0;
}

/* ---- image ------------------------------------------------------------------------------------ */
static inline int64_t bin_of(double v, const double *edges, int64_t n)
{
    /* searchsorted(edges, v, 'right') - 1 with v == edges[n] folded into the last bin;
       -1 = outside (also NaN) */
    if (!(v >= edges[0]) || !(v <= edges[n])) return -1;
    if (v == edges[n]) return n - 1;
    int64_t lo = 0, hi = n;               /* edges[lo] <= v < edges[hi] */
    while (hi - lo > 1) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (v >= edges[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

static inline double f32rt(double v) { return (double)(float)v; }

static inline void image_sample(const ora_image_desc *g, double x, double y, double z, double vy,
                                double frac, double *image, uint64_t *counts)
{
    if (g->downcast_f32) {
        x = f32rt(x); y = f32rt(y); z = f32rt(z); vy = f32rt(vy); frac = f32rt(frac);
    }
    const double *M = g->M;
    double radvel = vy + g->vrplanet;
    double xo = (M[0]*x + M[1]*y) + M[2]*z;
    double yo = (M[3]*x + M[4]*y) + M[5]*z;
    double zo = (M[6]*x + M[7]*y) + M[8]*z;
    double rho_obs = sqrt(xo*xo + zo*zo);
    int inview = (rho_obs > 1.0) || (yo < 0.0);
    frac = frac * (double)inview;
    double w;
    if (g->quantity == 0) {
        w = frac;
    } else {
        double gg = 0.0;
        for (int l = 0; l < g->n_lines; l++)
            gg += interp1(radvel, g->line_v[l], g->line_g[l], g->line_n[l]);
        w = frac * (double)sunlit(x, y, z) * gg / 1e6;
    }
    w = w / g->apix_cm2;
    int64_t ix = bin_of(xo, g->xedges, g->nx), iz = bin_of(zo, g->zedges, g->nz);
    if (ix < 0 || iz < 0) return;
    image[ix * g->nz + iz] += w;
    counts[ix * g->nz + iz] += 1;
}

void ora_image(const ora_image_desc *g, int64_t p, const double *x, const double *y,
               const double *z, const double *vy, const double *frac, double *image,
               uint64_t *counts)
{
    for (int64_t i = 0; i < p; i++)       /* sample order == np.bincount order */
        image_sample(g, x[i], y[i], z[i], vy[i], frac[i], image, counts);
}

/* Constant-step driver.  traj (nullable): [8][nrec][n], record 0 = initial state, record ct =
 * state after iteration ct (zero once dead, as the reference's results array).  final (nullable,
 * [8][n]) = the state stored at the packet's last processed iteration (frac = 0 and t = 0 if it
 * died).  steps (nullable) = iterations the packet was active.
 * img (nullable): every stored record with frac > 0 (records 0..n_iter, compress=True rule of
 * Output.py:523-524) is binned.  Returns the number of particle-steps.  n_threads <= 1 keeps the
 * image accumulation in exact sample order. */
static int64_t integrate_const_impl(const ora_forces *f, const ora_bodies *b, int64_t n,
                                    const double *soa0, double step,
                            int64_t n_iter, double outeredge, double *traj, int64_t nrec,
                            double *final, int64_t *steps, const ora_image_desc *img,
                            double *image, uint64_t *counts, int n_threads)
{
    int64_t work = 0;
    int64_t npix = img ? img->nx * img->nz : 0;
    if (n_threads < 1) n_threads = 1;
#ifndef _OPENMP
    n_threads = 1;
#endif
    double *pim = NULL; uint64_t *pct = NULL;
    if (img && n_threads > 1) {
        pim = calloc((size_t)npix * n_threads, sizeof(double));
        pct = calloc((size_t)npix * n_threads, sizeof(uint64_t));
    }
#pragma omp parallel num_threads(n_threads) reduction(+:work)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *im = (pim ? pim + (size_t)tid * npix : image);
        uint64_t *ct = (pct ? pct + (size_t)tid * npix : counts);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; i++) {
            double s[8];
            for (int c = 0; c < 8; c++) s[c] = soa0[c*n + i];
            if (traj) for (int c = 0; c < 8; c++) traj[((size_t)c*nrec + 0)*n + i] = s[c];
            int64_t k = 0;
            int alive = s[7] > 0;
            if (img && alive) image_sample(img, s[1], s[2], s[3], s[5], s[7], im, ct);
            while (alive && k < n_iter) {
                if (b) {
                    rk5_body(f, b, k, s, step, NULL);
                    fate_bodies(s, outeredge, b, k, step);
                } else {
                    rk5_one(f, s, step, NULL);
                    fate(s, outeredge, 0);
                }
                k++; work++;
                if (traj && k < nrec)
                    for (int c = 0; c < 8; c++) traj[((size_t)c*nrec + k)*n + i] = s[c];
                alive = s[7] > 0;
                if (img && alive) image_sample(img, s[1], s[2], s[3], s[5], s[7], im, ct);
            }
            if (final) for (int c = 0; c < 8; c++) final[c*n + i] = s[c];
            if (steps) steps[i] = k;
        }
    }
    if (pim) {
        for (int t = 0; t < n_threads; t++)
            for (int64_t q = 0; q < npix; q++) {
                image[q] += pim[(size_t)t*npix + q];
                counts[q] += pct[(size_t)t*npix + q];
            }
        free(pim); free(pct);
    }
    return work;
}

int64_t ora_integrate_const(const ora_forces *f, int64_t n, const double *soa0, double step,
                            int64_t n_iter, double outeredge, double *traj, int64_t nrec,
                            double *final, int64_t *steps, const ora_image_desc *img,
                            double *image, uint64_t *counts, int n_threads)
{
    return integrate_const_impl(f, NULL, n, soa0, step, n_iter, outeredge, traj, nrec, final,
                                steps, img, image, counts, n_threads);
}

int64_t ora_integrate_const_bodies(const ora_forces *f, const ora_bodies *b, int64_t n,
                                   const double *soa0, double step, int64_t n_iter,
                                   double outeredge, double *traj, int64_t nrec, double *final,
                                   int64_t *steps, const ora_image_desc *img, double *image,
                                   uint64_t *counts, int n_threads)
{
    return integrate_const_impl(f, b, n, soa0, step, n_iter, outeredge, traj, nrec, final, steps,
                                img, image, counts, n_threads);
}

/* Variable-step driver, one packet at a time (Output.py:221-359).  out [8][n]; hstore (nullable)
 * = the stored step_size column at exit.  Returns rk5 particle-steps attempted; *bad (nullable)
 * counts assertion-class events (non-finite errmax, negative accepted frac, non-positive step). */
int64_t ora_integrate_var(const ora_forces *f, int64_t n, const double *soa0, double resolution,
                          double outeredge, int64_t max_steps, double *out, double *hstore,
                          int64_t *bad)
{
    const double safety = 0.95;     /* shrink exponent -0.25: ora_pow_m025 */
    const double resx = resolution, resv = 0.1 * resolution, resf = resolution;
    int64_t work = 0, nbad = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+:work, nbad)
    for (int64_t i = 0; i < n; i++) {
        double s[8], hs = 1000.0;
        for (int c = 0; c < 8; c++) s[c] = soa0[c*n + i];
        int64_t it = 0;
        while (s[0] > resolution && s[7] > 0.0 && it < max_steps) {
            double h = fmin(s[0], hs);
            if (!(h > 0)) { nbad++; break; }
            double t[8], d[8];
            memcpy(t, s, sizeof t);
            rk5_one(f, t, h, d);
            work++; it++;
            double fr_scale = resf + fabs(t[7]) * resf;
            double e = d[0];                                   /* time column: 0 */
            int finite = 1;     /* fmax would drop a NaN quotient; the assert (Output.py:284) must see it */
            for (int c = 1; c <= 7; c++) {
                double scale = c <= 3 ? resx + fabs(t[c]) * resx
                             : c <= 6 ? resv + fabs(t[c]) * resv : fr_scale;
                double q = d[c] / scale;
                if (!isfinite(q)) finite = 0;
                e = fmax(e, q);
            }
            if (!finite) e = NAN;
            if (!isfinite(e)) { nbad++; break; }
            if (t[7] < 0 && e < 1) nbad++;
            if ((t[7] - s[7] > fr_scale) && (e > 1)) e = 1.1;
            double hold = h;
            if (e < 1e-7) { e = 1.0; hold = h * 10; }
            if (e < 1.0) {
                fate(t, outeredge, 1);
                memcpy(s, t, sizeof s);
            } else {
                double hn = safety * hold * ora_pow_m025(e);
                if (!isfinite(hn)) { nbad++; break; }
                hs = fmax(hn, 0.1 * hold);
            }
        }
        for (int c = 0; c < 8; c++) out[c*n + i] = s[c];
        if (hstore) hstore[i] = hs;
    }
    if (bad) *bad = nbad;
    return work;
}

int ora_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* exposed for the math cross-checks in tests/test_oracle_c.py */
void ora_math_batch(int which, int64_t n, const double *in, double *out)
{
    for (int64_t i = 0; i < n; i++)
        out[i] = which == 0 ? ora_exp(in[i]) : which == 1 ? ora_log(in[i]) : ora_cube(in[i]);
}
