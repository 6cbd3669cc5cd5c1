/* nexoclom_hip.h -- C ABI of libnexoclom_hip.so: nexoclom's particle_tracking + image hot path
 * on AMD MI355X (gfx950).
 *
 * The reference (mburger-stsci/nexoclom, pure Python/NumPy) has no FFI on this path; its "plugin
 * boundary" is a set of plain Python call sites.  Each entry point below replaces one of them and
 * is what a ctypes binding inside the reference would call (INTEGRATION.md shows the stubs).
 * Citations are paths under the reference tree's nexoclom/ directory.
 *
 *   nxc_state              state(x, output)                       particle_tracking/state.py:17-74
 *   nxc_rk5_step           rk5(output, X0, h)                     particle_tracking/rk5.py:21-54
 *   nxc_integrate_const    Output.constant_step_size_driver()     particle_tracking/Output.py:368-455
 *                          (+ optionally fused ModelImage.create_image of every stored step)
 *   nxc_integrate_const_rows / nxc_rows_fetch / nxc_rows_build
 *                          the same driver followed by save()'s   particle_tracking/Output.py:523-543
 *                          frac > 0 row filter and 32-bit cast;
 *                          the rows can stay on the device for
 *                          nxc_image_accumulate_rows /
 *                          nxc_los_accumulate_rows
 *   nxc_integrate_var      Output.variable_step_size_driver()     particle_tracking/Output.py:221-366
 *   nxc_image_accumulate   ModelImage.create_image()              data_simulation/ModelImage.py:229-274
 *                          + ModelResult.packet_weighting()       data_simulation/ModelResult.py:140-170
 *                          + Histogram2d()                        math/histogram.py:28-39
 *   nxc_image_allreduce    the per-output-file image sum          data_simulation/ModelImage.py:96-98
 *   nxc_los_accumulate     compute_iteration() inner work          data_simulation/compute_iteration.py:138-217
 *   nxc_set_bounce         bouncepackets() inside the drivers     particle_tracking/bouncepackets.py:5-100
 *   nxc_packets_sample     surface/speed/angular_distribution()   initial_state/source_distribution.py:37-283
 *   nxc_set_bodies         (extension) moons + plasma-torus loss  equations: particle_tracking/state.py:5-10
 *
 * Conventions
 *   - Every function returns 0 on success or a negative nxc_status; nothing is thrown across the
 *     boundary.  nxc_last_error_string() describes the last failure on the calling thread.
 *   - All array arguments are caller-owned HOST buffers, C-contiguous, fp64 unless stated; no
 *     pointer is retained after the call returns, except the table pointers inside nxc_forces /
 *     nxc_image_desc, which are copied to the device during nxc_set_forces / nxc_set_image.
 *   - Packet arrays are struct-of-arrays: soa[c*n + i], column c = 0..7 =
 *     t_remaining, x, y, z, vx, vy, vz, frac (the reference's (N,8) row layout, transposed);
 *     lengths in planet radii, times in seconds.
 *   - A handle owns one device, one HIP stream, its tables, a resident packet set and a resident
 *     image pair.  One host thread per handle; calls are synchronous unless named *_async.
 *   - The reference's asserts (Output.py:254,284,287,388-389; rk5.py:52) become counters
 *     (nxc_counters) that the Python shim turns back into AssertionError.
 */
#ifndef NEXOCLOM_HIP_H
#define NEXOCLOM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: nxc_source_desc grew (tabulated speeds, surface maps, generator), resident row stores
 * 3: bounded waits on collectives (nxc_comm_set_timeout / _abort / _request_abort), nxc_allreduce_f64,
 *    nxc_packets_upload_pieces, nxc_counters.wave_trips, NXC_ERR_NOMEM / NXC_ERR_INCOMPLETE */
#define NXC_ABI_VERSION 3
#define NXC_MAX_LINES 4

typedef enum {
    NXC_OK = 0,
    NXC_ERR_HIP = -1,        /* a HIP runtime call failed (message has the HIP error)   */
    NXC_ERR_ARG = -2,        /* invalid argument / missing prerequisite call            */
    NXC_ERR_NO_DEVICE = -3,  /* no gfx950-class device visible                          */
    NXC_ERR_RCCL = -4,       /* librccl missing or a collective failed                  */
    NXC_ERR_STATE = -5,      /* handle not in the state the call needs                  */
    NXC_ERR_NOMEM = -6,      /* device memory: an allocation failed or the result would
                                not fit (callers may split the work and call again)     */
    NXC_ERR_INCOMPLETE = -7  /* nxc_synchronize after nxc_integrate_const_streamed: the
                                kernel gave up waiting for its queue; results are partial */
} nxc_status;

/* Scalars and table consumed by state() (what Output.__init__ hangs on `output`,
 * particle_tracking/Output.py:105-128). */
typedef struct nxc_forces {
    double GM;          /* R^3/s^2; NEGATIVE as in the reference (solarsystem/SSObject.py:53)    */
    double vrplanet;    /* R/s, radial velocity of the planet w.r.t. the Sun                     */
    double photo;       /* 1/s, loss_info.photo (state.py:48-52); used when has_photo            */
    double lifetime;    /* s; > 0 selects the constant loss rate 1/lifetime (state.py:44-46)     */
    int32_t gravity;    /* inputs.forces.gravity                                                 */
    int32_t radpres;    /* inputs.forces.radpres                                                 */
    int32_t has_photo;  /* loss_info.photo is not None                                           */
    int32_t reserved;
    int64_t n_tab;      /* radiation-acceleration table length (>= 2)                            */
    const double *v_tab;/* R/s, strictly ascending  (radpres.velocity)                           */
    const double *a_tab;/* R/s^2                    (radpres.accel)                              */
} nxc_forces;

/* Everything create_image needs besides the packets (ModelImage.py:53-78,229-269). */
typedef struct nxc_image_desc {
    double M[9];          /* row-major sun->observer rotation (ModelImage.image_rotation)        */
    double vrplanet;      /* R/s, added to vy for the g-value lookup (ModelImage.py:242-243)     */
    double apix_cm2;      /* pixel area in cm^2; weights are divided by it (ModelImage.py:262)   */
    int32_t quantity;     /* 0 = column/density (w = frac), 1 = radiance/difrad                  */
    int32_t n_lines;      /* number of g-value tables summed for radiance (<= NXC_MAX_LINES)     */
    int32_t downcast_f32; /* 1: round x,y,z,vy,frac through float32 first, as the reference's
                             save()/restore() pair does (Output.py:528-543,555-570)              */
    int32_t reserved;
    int64_t nx, nz;       /* image dims (bins along x_obs, z_obs)                                */
    const double *xedges; /* nx+1 bin edges = np.linspace(lo, hi, nx+1)                          */
    const double *zedges; /* nz+1 bin edges                                                      */
    int64_t line_n[NXC_MAX_LINES];
    const double *line_v[NXC_MAX_LINES]; /* R/s ascending (gValue.velocity converted)            */
    const double *line_g[NXC_MAX_LINES]; /* 1/s           (gValue.g)                             */
} nxc_image_desc;

/* Work and assertion counters of the last integrate / image call on the handle. */
typedef struct nxc_counters {
    uint64_t particle_steps; /* rk5 steps taken = sum over iterations of active packets          */
    uint64_t samples;        /* packet samples offered to the image (frac > 0 records)           */
    uint64_t samples_binned; /* of those, inside the image range                                 */
    uint64_t nonfinite;      /* non-finite state / errmax / weight events                        */
    uint64_t bad_step;       /* step size <= 0 or not finite (variable driver)                   */
    uint64_t neg_frac;       /* accepted step with frac < 0 (variable driver, Output.py:287)     */
    uint64_t unfinished;     /* packets stopped by max_steps before reaching their end time      */
    uint64_t wave_trips;     /* measurement, not a result: trips of a wave through the persistent
                                step loop (64 lanes each); particle_steps / (64 wave_trips) is the
                                share of lanes that held a live packet.  Depends on scheduling. */
} nxc_counters;

typedef struct nxc_handle nxc_handle;

/* ---- device / library ------------------------------------------------------------------------ */
int nxc_abi_version(void);
int nxc_device_count(int *count);
const char *nxc_last_error_string(void);
int nxc_create(int device, nxc_handle **out);
int nxc_destroy(nxc_handle *h);
int nxc_device_name(nxc_handle *h, char *buf, int buflen);
int nxc_device_bus_id(nxc_handle *h, char *buf, int buflen);  /* PCI bus id, e.g. "0000:05:00.0" */
int nxc_synchronize(nxc_handle *h);
int nxc_mem_info(nxc_handle *h, uint64_t *free_bytes, uint64_t *total_bytes);  /* device memory */

/* ---- set-up ---------------------------------------------------------------------------------- */
int nxc_set_forces(nxc_handle *h, const nxc_forces *f);
int nxc_set_image(nxc_handle *h, const nxc_image_desc *d);   /* also zeroes the resident image    */

/* ---- f-2: surface re-emission (particle_tracking/bouncepackets.py:39-100) --------------------------
 * With a bounce description set, a packet that ends a step inside the planet is moved back to its
 * impact point and re-emitted (isotropic rebound, energy accommodation to the local surface
 * temperature through the v(T, probability) spline, sticking loss) instead of being absorbed
 * (Output.py:398-402).  NULL restores perfect sticking.  Random numbers are Philox draws keyed
 * by (seed; first_index + packet row, bounce number): statistically the reference's process.
 * tx[nx], ty[ny], coef[(nx-4)*(ny-4)]: knots/coefficients of scipy's RectBivariateSpline
 * (SurfaceInteraction.py:56) in km/s; ignored when accomfactor == 0. */
typedef struct nxc_bounce_desc {
    double GM;            /* R^3/s^2 (negative), for the impact speed (bouncepackets.py:59)      */
    double unit_km;
    double accomfactor;   /* 0 = elastic rebound at the impact speed                            */
    double stickcoef;     /* constant sticking (used when !temp_dependent)                      */
    double A[3];          /* temperature-dependent sticking A0 exp(A1 T) + A2                   */
    double t0, t1;        /* surface temperature: t0 night, t0 + t1 |cos lon cos lat|^0.25 day  */
    int32_t temp_dependent;
    int32_t reserved;
    int64_t nx, ny;
    const double *tx, *ty, *coef;
    uint64_t seed;
} nxc_bounce_desc;

int nxc_set_bounce(nxc_handle *h, const nxc_bounce_desc *d);
int nxc_set_first_index(nxc_handle *h, int64_t first_index);  /* RNG counter of resident packet 0 */

/* ---- f-4 (tail): moons and plasma-torus loss ---------------------------------------------------
 * EXTENSION -- no reference implementation exists: particle_tracking/state.py:5-10 documents
 * the multi-body equations of motion, :56-70 holds the commented charge-exchange stub, and
 * Output.py:153-155 asserts 'Not set up' for planets with moons.  Parity is therefore against
 * oracle/ only ("parity unpinned"); tests/ add physics checks (Jacobi integral, limits).
 *   accel += sum_m gm[m] (r - r_m(t)) / |r - r_m(t)|^3
 *   loss  += chx_k0 exp(-((rho - chx_rho0)/chx_width)^2 - (z/chx_height)^2)
 *            [* |v - chx_omega z^ x r| / (chx_omega chx_rho0)   when chx_omega > 0]
 *   a packet within radius[m] of moon m after a step is absorbed.
 * Moon m moves on a circle of radius a[m] in the planet's equatorial (x, y) plane; its orbital
 * phase is phi[m] at t_remaining = 0 (geometry.phi: 0 = superior conjunction (+y), pi/2 = over
 * the dawn terminator (-x), docs/nexoclom/inputfiles.rst:72-77) and phi[m] - omega[m] t at
 * t_remaining = t:  r_m = a (-sin, cos, 0).  At stage n (Dormand-Prince node c_n) of step k the
 * phase is evaluated as theta_k + delta_n, theta_k = phi - omega (t0 - k h), delta_n = omega c_n h,
 * with sin/cos of the sum formed from the two sincos() pairs by the angle-addition formulas (that
 * is the definition; the oracles follow it to the bit).  Applies to nxc_integrate_const* only
 * (every packet starts at t_remaining = t0); nxc_state / nxc_rk5_step / nxc_integrate_var and surface
 * re-emission refuse to run while bodies are set.  NULL or n_moons == 0 && !chx_on clears. */
#define NXC_MAX_MOONS 4
typedef struct nxc_bodies_desc {
    int32_t n_moons;
    int32_t chx_on;
    double gm[NXC_MAX_MOONS];      /* R^3/s^2, negative like nxc_forces.GM */
    double radius[NXC_MAX_MOONS];  /* R */
    double a[NXC_MAX_MOONS];       /* R */
    double omega[NXC_MAX_MOONS];   /* rad/s */
    double phi[NXC_MAX_MOONS];     /* rad */
    double t0;                     /* s: t_remaining of every packet at the start of the run */
    double chx_k0;                 /* 1/s */
    double chx_rho0, chx_width, chx_height;   /* R */
    double chx_omega;              /* rad/s; 0 = no dependence on the relative speed */
} nxc_bodies_desc;

int nxc_set_bodies(nxc_handle *h, const nxc_bodies_desc *d);

/* ---- a-2: state() ---------------------------------------------------------------------------- */
int nxc_state(nxc_handle *h, int64_t n, const double *x, const double *y, const double *z,
              const double *vy, double *ax, double *ay, double *az, double *ioniz);

/* ---- a-1: rk5() -- one Dormand-Prince step, per-packet step size h[n] -------------------------
 * soa_out receives the 5th-order state; delta_out (nullable, [8][n]) the reference's
 * |h * sum_{i<6} (b5-b4)_i k_i| error estimate (rk5.py:38-46).
 * Arithmetic: the reference's operations in the reference's order, each rounded once, with two
 * documented departures of at most an ulp -- deterministic r^3 / exp / log (NumPy's own are 1-ulp
 * routines) and the terms of the tableau sums (rk5.py:33-35,41-43), which are fused multiply-adds
 * (one rounding where NumPy has two).  Against rk5.py itself: 1e-13 relative per step. */
int nxc_rk5_step(nxc_handle *h, int64_t n, const double *soa_in, const double *hstep,
                 double *soa_out, double *delta_out);

/* ---- resident packets / image (what a long run keeps in HBM) ----------------------------------- */
int nxc_packets_upload(nxc_handle *h, int64_t n, const double *soa0);
/* The same for a resident set that the host holds in pieces (the Outputs of one launch of
 * Input.run, each an [8][counts[p]] array): piece p's packets follow those of the pieces before it. */
int nxc_packets_upload_pieces(nxc_handle *h, int32_t n_pieces, const int64_t *counts,
                              const double *const *soa);
int nxc_image_clear(nxc_handle *h);
int nxc_image_download(nxc_handle *h, double *image /* nx*nz */, uint64_t *counts /* nx*nz */);
int nxc_counters_get(nxc_handle *h, nxc_counters *out);
int nxc_last_kernel_ms(nxc_handle *h, float *ms);  /* HIP-event time of the last integrate/image launch */

/* ---- f-4: initial states sampled on the device -----------------------------------------------------
 * Fills the resident packet set with n packets of the sources of
 * initial_state/source_distribution.py:37-283 using a counter-based generator (Philox-4x32-10
 * keyed by seed; counter = first_index + i), so shards on different GPUs draw disjoint,
 * reproducible packets.  soa_out (nullable) receives the [8][n] states.  Statistically equivalent
 * to the reference's NumPy sampler, not draw-for-draw.
 *   surface   spatial_type 0  uniform in sin(latitude) and longitude (:47-62)
 *             spatial_type 1  'surface spot' (:96-118): accept/reject on a density map tabulated
 *                             on linspace(0, 2 pi, map_nlon) x linspace(-pi/2, pi/2, map_nlat),
 *                             bilinear between nodes (math/randomdeviates.py:36-83); the device
 *                             runs the trials per packet instead of in rounds of n candidates
 *   speed     speed_type 0/1  flat, gaussian (:141-147,169-171)
 *             speed_type 2    inverse CDF of a tabulated flux density (maxwellian, sputtering;
 *                             :148-168, math/randomdeviates.py:8-33): v = interp(u, speed_cdf,
 *                             speed_v), speed_cdf non-decreasing from 0 to 1
 *   direction angular_type 0  radial, 1 isotropic (:198-252) */
typedef struct nxc_source_desc {
    double endtime;        /* s                                                                 */
    double exobase;        /* R                                                                 */
    double sinlat0, sinlat1; /* sin of the latitude range                                       */
    double lon0, lon1;     /* rad; lon1 already += 2 pi when the range wraps                    */
    double vprob, vwidth;  /* km/s: flat = vprob +- vwidth (delv); gaussian = mean, sigma       */
    double unit_km;        /* planet radius                                                     */
    double sinalt0, sinalt1, az0, az1;   /* isotropic launch cone                               */
    int32_t random_time;   /* 1: t = u*endtime (variable-step runs, Output.py:138-139)          */
    int32_t speed_type;    /* 0 flat, 1 gaussian, 2 tabulated                                   */
    int32_t angular_type;  /* 0 radial, 1 isotropic                                             */
    int32_t is_planet;     /* longitude convention (source_distribution.py:13-28)               */
    uint64_t seed;
    int64_t first_index;
    int32_t spatial_type;  /* 0 uniform, 1 surface spot                                         */
    int32_t reserved;
    int64_t n_speed;       /* speed_type 2: table length (>= 2)                                 */
    const double *speed_cdf;   /* [n_speed] non-decreasing, first 0, last 1                     */
    const double *speed_v;     /* [n_speed] km/s                                                */
    int64_t map_nlon, map_nlat;   /* spatial_type 1: density map dims (>= 2 each)               */
    const double *map;     /* [map_nlon][map_nlat], >= 0                                        */
    /* generator 1: the reference's own seeded stream, numpy.random.default_rng(seed) = PCG64
     * (Output.py:92), reproduced on the device: the packets are rows pcg_row0 .. pcg_row0 + n - 1
     * of the pcg_n-packet vectors the reference would draw one after the other ([launch time,]
     * sin latitude, longitude, speed, [sin altitude, azimuth]); the uniforms are bit-identical
     * to Generator.random(pcg_n), the states equal the host sampler's to libm rounding.  Only
     * sources whose every draw is such a vector: spatial_type 0, speed_type 0 (flat), any
     * angular_type.  pcg_state / pcg_inc: PCG64(seed).state['state'] as {high, low} words.      */
    int32_t generator;     /* 0 Philox-4x32-10 (counter-based, statistical parity), 1 PCG64     */
    int32_t reserved2;
    uint64_t pcg_state[2], pcg_inc[2];
    int64_t pcg_n, pcg_row0;
    /* dest_total > 0: the n packets are piece [dest_offset, dest_offset + n) of a resident set of
     * dest_total packets that several calls fill (pieces in ascending order, the first with
     * dest_offset 0; the set is usable once the last piece is in).  0: they are the whole set.    */
    int64_t dest_offset, dest_total;
} nxc_source_desc;

int nxc_packets_sample(nxc_handle *h, const nxc_source_desc *d, int64_t n, double *soa_out);

/* ---- a-3 (+ fused a-6..a-8): constant-step driver over the resident packets --------------------
 * Runs n_iter iterations of {rk5(step); impact r<1; escape r>outeredge; vanish frac<1e-10}
 * (Output.py:384-431) for every packet until it dies.
 *   flags & NXC_RUN_IMAGE : every stored record with frac > 0 -- the initial state and the state
 *                           after each iteration -- is binned into the resident image with the
 *                           nxc_set_image description (compress=True rule, Output.py:523-524).
 *   traj_out (nullable)   : host [8][nrec][n]; record 0 = initial state, record k = state after
 *                           iteration k, zeros once dead (the reference's `results`, transposed).
 *                           nrec must be >= n_iter+1 when given.
 *   final_out (nullable)  : host [8][n], state at the packet's last processed iteration.
 *   steps_out (nullable)  : host int64[n], iterations the packet was active.
 * The kernel is always the persistent lane-refill integrator.  With traj_out == NULL no trajectory
 * is ever materialised; with traj_out a second pass of the same kernel writes the live records
 * (see nxc_integrate_const_rows) and a layout kernel expands them into the dense array. */
#define NXC_RUN_IMAGE 1u
int nxc_integrate_const(nxc_handle *h, double step, int64_t n_iter, double outeredge,
                        uint32_t flags, double *traj_out, int64_t nrec, double *final_out,
                        int64_t *steps_out);
/* The same trajectories in the form Output.save() keeps them (compress=True, Output.py:523-524):
 * only the records with frac > 0, packet-major (all records of packet 0 in step order, then packet
 * 1, ...), i.e. the row order of the reference's X frame after the frac > 0 filter.  A packet's
 * live records are a prefix of its step axis, so the zero padding of the dense [8][nrec][n] array
 * (typically > 90 % of it) is never materialised or transferred.
 *   nxc_integrate_const_rows : pass 1 (persistent kernel) counts the live records per packet;
 *                              lengths_out int64[n] (nullable), *total_out = their sum.
 *   nxc_rows_fetch           : pass 2 (the same persistent kernel, now writing: a lane owns its
 *                              packet for life, so record k goes to row offset[packet] + k)
 *                              re-integrates and delivers rows_out, host [9][total]: the 8 state
 *                              columns and lossfrac accumulated as (lossfrac + frac_before) -
 *                              frac_after per step (Output.py:420-421), starting from 0 (the
 *                              reference's starts from uninitialised memory, Output.py:378).  Must
 *                              follow nxc_integrate_const_rows on the same resident packets. */
int nxc_integrate_const_rows(nxc_handle *h, double step, int64_t n_iter, double outeredge,
                             int64_t *lengths_out, int64_t *total_out);
int nxc_rows_fetch(nxc_handle *h, double *rows_out);
/* The same rows narrowed to float32 on the device, host [9][total] floats: what save()'s down-cast
 * (Output.py:528-543, which every reference Output ends in, Output.py:202) makes of them, at half
 * the device-to-host bytes. */
int nxc_rows_fetch_f32(nxc_handle *h, float *rows_out);
/* The same rows kept ON THE DEVICE (pass 2 of the protocol above, in place of nxc_rows_fetch): an
 * nxc_rows store holds the nine columns [9][total] plus the packet-index column (row -> number of
 * its packet in the resident set: the reference's X.Index, Output.py:438), as float32 / int32
 * when narrow != 0 (what save() stores, Output.py:528-543) or float64 / int64.  A store outlives
 * the packets it was built from; it is what the reference's per-Output file is to
 * ModelImage / LOSResult (ModelImage.py:85-98, LOSResult.py:264-266), minus the disk and the
 * host.  Stores are freed explicitly; a handle may own any number of them.
 *   nxc_rows_download          rows [first, first + count) -> cols_out host [9][count] (nullable)
 *                              and index_out host [count] (nullable), in the store's types; runs on
 *                              a stream of its own and may be called from a second host thread
 *                              (a file writer) while the handle's thread computes
 *   nxc_image_accumulate_rows  create_image over those rows (columns x, y, z, vy, frac), as
 *                              nxc_image_accumulate[_f32] without the host round trip
 *   nxc_los_accumulate_rows    compute_iteration over those rows, as nxc_los_accumulate[_f32];
 *                              index_shift is subtracted from the store's index column (the first
 *                              packet of the Output the rows belong to) before `included` is set */
typedef struct nxc_rows nxc_rows;
int nxc_rows_build(nxc_handle *h, int narrow, nxc_rows **out);
int nxc_rows_info(const nxc_rows *r, int64_t *total, int32_t *is_f32);
int nxc_rows_download(nxc_handle *h, const nxc_rows *r, int64_t first, int64_t count,
                      void *cols_out, void *index_out);
/* Frees the store.  Its two device blocks (when large) are kept by the handle for the next store of
 * about the same size -- hipFree / hipMalloc of tens of GB cost seconds -- and go back to the driver
 * when anything else needs the memory; nxc_mem_info counts them as free. */
int nxc_rows_free(nxc_handle *h, nxc_rows *r);
int nxc_image_accumulate_rows(nxc_handle *h, const nxc_rows *r, int64_t first, int64_t count);
/* Same launch without any host transfer or synchronisation (bench / pipelining). */
int nxc_integrate_const_async(nxc_handle *h, double step, int64_t n_iter, double outeredge,
                              uint32_t flags);

/* Upload + integrate as one pipelined pass ("incl. H2D of X0"): the n
 * packets of host array soa0 [8][n] are cut into `pieces` (1..32); piece p + 1 crosses PCIe and is
 * put into queue order while piece p is integrated.  Asynchronous like nxc_integrate_const_async
 * (soa0 must stay valid until nxc_synchronize); afterwards the packets are the resident set, the
 * image holds their samples and nxc_counters_get reports the pass.  Same results as
 * nxc_packets_upload + nxc_integrate_const_async (the order of the queue changes no packet).
 * Not with moons or surface re-emission set (NXC_ERR_STATE: upload first).
 * The persistent kernel waits for its pieces while small ordering kernels on a second stream
 * prepare them; where those cannot run beside it (a profiler in counter mode serialises kernels)
 * a wave gives up after three seconds of waiting, the launch ends with part of the packets
 * integrated (nxc_counters.unfinished counts the rest), and the nxc_synchronize that follows
 * returns NXC_ERR_INCOMPLETE: the image and counters are partial, the resident set is dropped, the
 * handle stays usable (nxc_packets_upload + nxc_integrate_const_async is the sequential form). */
int nxc_integrate_const_streamed(nxc_handle *h, int64_t n, const double *soa0, int32_t pieces,
                                 double step, int64_t n_iter, double outeredge, uint32_t flags);

/* ---- a-4: variable-step driver over the resident packets ---------------------------------------
 * final_out host [8][n]; hstore_out (nullable) host [n] = stored step_size column at exit.
 * One persistent launch; the queue is ordered by remaining time over launch speed.  Under 24
 * packets per lane (4.7e6 packets) the waves of a SIMD take turns at issue priority and, once the
 * queue is drained, sparse waves hand their live packets to one wave per SIMD; above it the plain
 * form: same arithmetic, same bits (the environment variable NXC_TEST_VAR_VARIANT = "fair" /
 * "plain" forces either, for tests). */
int nxc_integrate_var(nxc_handle *h, double resolution, double outeredge, int64_t max_steps,
                      double *final_out, double *hstore_out);

/* ---- a-6..a-8: image of p stored samples -------------------------------------------------------
 * Adds to the resident image pair (use nxc_image_clear / nxc_image_download around it). */
int nxc_image_accumulate(nxc_handle *h, int64_t p, const double *x, const double *y,
                         const double *z, const double *vy, const double *frac);
/* The same for samples in the 32-bit form Output.save() stores them in (Output.py:528-543): what
 * restore()'s up-cast (Output.py:555-570) followed by nxc_image_accumulate gives, bit for bit,
 * with half the host-to-device bytes and no 64-bit copy on the host. */
int nxc_image_accumulate_f32(nxc_handle *h, int64_t p, const float *x, const float *y,
                             const float *z, const float *vy, const float *frac);
/* How the image calls above (and nxc_image_accumulate_rows) add their samples to the image:
 * mode 1 = one global atomic pair per binned sample (k_image); mode 2 = LDS-privatised tiles
 * (k_image_bin + k_image_tiles: the samples are filed by image tile first, a workgroup sums a tile
 * in LDS and hands it over once -- the replacement of np.histogram2d's bincount,
 * math/histogram.py:34, at HBM speed instead of atomic-request speed); mode 0 (default) = tiles for
 * 2^17 samples and more when the image fits them (up to 32 tiles of 8192 pixels: 512 x 512),
 * atomics otherwise.  Packet counts are identical either way, weight sums equal to the order of
 * fp64 additions.  tile_pixels: 0 = 8192; slab_samples: 0 = 2^28, the samples that go through the
 * two passes at a time (their chunk scratch is 10 bytes per sample at worst); smaller values of
 * both exist for tests (more tiles on a small image, several slabs of a small sample set). */
int nxc_image_mode(nxc_handle *h, int mode, int tile_pixels, int64_t slab_samples);

/* ---- f-1: spacecraft line-of-sight cones ----------------------------------------------------------
 * For each of S spectra (spacecraft position + boresight) sum weight/Apix over the stored samples
 * inside the view cone of half-angle dphi, in front of the planet cut-off, that the reference's
 * KD-tree ball pre-selection would have offered (compute_iteration.py:164-185), with the shadow test
 * at the line-of-sight foot point (:202-206).  All angles/trig values are computed by the caller
 * (NumPy) so that thresholds are the reference's own:
 *   sc            host [8][S]: x, y, z, xbore, ybore, zbore, dist_from_plan (1e30 if the LOS misses
 *                 the planet, :105-115), ladder length K_i (as a double)
 *   ladder        t_k = t_{k-1} (1 + sin dphi), t_0 = sin dphi, the longest of the S ladders (:164-167)
 *   cos_threshold smallest double c with arccos(c) <= dphi
 * Outputs: radiance[S] (fp64 sum), npackets[S], included[n_index] (nullable: set to 1 for every
 * packet index seen in a cone, :191), used_pairs (nullable, [2][used_cap]: spectrum, sample row of
 * every pair with weight > 0, :210) and *n_used (pairs found; may exceed used_cap). */
typedef struct nxc_los_desc {
    double dphi, sin_dphi, sin_2dphi, cos_threshold;
    double vrplanet;      /* R/s                                                                   */
    double unit_cm;       /* planet radius in cm: Apix = pi (d sin dphi)^2 unit_cm^2               */
    int32_t n_lines;      /* g-value tables summed (radiance is the only quantity the reference has
                             here, compute_iteration.py:198-213)                                   */
    int32_t reserved;
    int64_t line_n[NXC_MAX_LINES];
    const double *line_v[NXC_MAX_LINES];
    const double *line_g[NXC_MAX_LINES];
    int64_t n_ladder;
    const double *ladder;
} nxc_los_desc;

int nxc_los_accumulate(nxc_handle *h, const nxc_los_desc *d, int64_t S, const double *sc,
                       int64_t P, const double *x, const double *y, const double *z,
                       const double *vy, const double *frac, const int64_t *index,
                       int64_t n_index, double *radiance, int64_t *npackets, uint8_t *included,
                       int64_t used_cap, int64_t *used_pairs, int64_t *n_used);
/* The same for sample columns in the 32-bit form Output.save() stores (Output.py:528-543); the
 * device widens them exactly as restore() would (Output.py:555-570). */
int nxc_los_accumulate_f32(nxc_handle *h, const nxc_los_desc *d, int64_t S, const double *sc,
                           int64_t P, const float *x, const float *y, const float *z,
                           const float *vy, const float *frac, const int64_t *index,
                           int64_t n_index, double *radiance, int64_t *npackets,
                           uint8_t *included, int64_t used_cap, int64_t *used_pairs,
                           int64_t *n_used);

int nxc_los_accumulate_rows(nxc_handle *h, const nxc_los_desc *d, int64_t S, const double *sc,
                            const nxc_rows *r, int64_t first, int64_t count, int64_t index_shift,
                            int64_t n_index, double *radiance, int64_t *npackets,
                            uint8_t *included, int64_t used_cap, int64_t *used_pairs,
                            int64_t *n_used);

/* ---- a-9 / multi-GPU: sum of the per-GPU image pairs over RCCL ---------------------------------
 * One process per GPU.  Rank 0 calls nxc_comm_unique_id and hands the 128 bytes to the other
 * ranks (any side channel); every rank then calls nxc_comm_init (NXC_ERR_ARG when RCCL refuses
 * the layout, which is what happens when two ranks share one device).  nxc_image_allreduce sums the
 * resident image and packet counts over all ranks in place (one fp64 all-reduce: the device keeps
 * {weight sum, count} interleaved, counts as integer-valued doubles, exact below 2^53). */
#define NXC_UNIQUE_ID_BYTES 128
int nxc_comm_unique_id(uint8_t id[NXC_UNIQUE_ID_BYTES]);
int nxc_comm_init(nxc_handle *h, const uint8_t id[NXC_UNIQUE_ID_BYTES], int rank, int nranks);
int nxc_comm_destroy(nxc_handle *h);
int nxc_image_allreduce(nxc_handle *h);
int nxc_allreduce_max_f64(nxc_handle *h, double *value);   /* control plane: max-over-ranks timer */
int nxc_allreduce_sum_f64(nxc_handle *h, double *value);   /* control plane: whole-job work counters */
int nxc_barrier(nxc_handle *h);
/* In-place sum over the ranks of n host doubles (the per-file radiance sum of
 * data_simulation/LOSResult.py:264-266 across GPUs: S radiances + S packet counts). */
int nxc_allreduce_f64(nxc_handle *h, double *values, int64_t n);
/* No wait on a collective is unbounded.  nxc_image_allreduce only enqueues; the next call that
 * waits for the handle's stream (nxc_synchronize, a download, the scalar reductions above) polls
 * the stream together with ncclCommGetAsyncError, and when `seconds` have passed (default 120, or
 * the environment's NXC_COLLECTIVE_TIMEOUT_S at nxc_comm_init) -- a peer rank died or never issued
 * its half of the collective -- it calls ncclCommAbort, leaves the handle without a communicator
 * and returns NXC_ERR_RCCL.  The data the collective was to produce are then undefined; the process
 * is expected to report and exit.  nxc_comm_abort does the same on request (a rank that must
 * leave while its peers may already be inside a collective). */
int nxc_comm_set_timeout(nxc_handle *h, double seconds);
int nxc_comm_abort(nxc_handle *h);
/* The one call that may come from ANOTHER thread while the owning thread waits inside the library
 * (the control plane's failure watcher): the wait in progress, or the next collective, ends with
 * NXC_ERR_RCCL at once instead of at the deadline. */
int nxc_comm_request_abort(nxc_handle *h);
/* Fault injection for the tests of the deadline: holds the handle's stream for `seconds` (<= 30;
 * the kernel ends by itself) and marks it as a collective in flight. */
int nxc_comm_test_stall(nxc_handle *h, double seconds);

/* ---- measurement helpers for bench.py's roofline object -------------------------------------------
 * nxc_stream_copy_gbs: best of `reps` device-to-device streaming copies of `bytes` (16 B per lane),
 * GB/s of bytes read + written: the box's own HBM ceiling.  nxc_shader_clock_mhz: the clock the
 * chip holds under an fp64 load, from in-kernel stamps of a diagnostic launch. */
int nxc_stream_copy_gbs(nxc_handle *h, int64_t bytes, int reps, double *gbs);
int nxc_shader_clock_mhz(nxc_handle *h, double *mhz);

/* Diagnostics of generator 1: out[nvec][count] = the uniforms of draws 0..nvec-1 for rows row0 ..
 * row0 + count - 1 of n-packet vectors, i.e. default_rng(seed).random(n)[row0:row0+count] nvec
 * times in a row. */
int nxc_pcg64_uniforms(nxc_handle *h, const uint64_t state[2], const uint64_t inc[2], int64_t n,
                       int64_t row0, int64_t count, int32_t nvec, double *out);

/* ---- diagnostics used by the parity tests -------------------------------------------------------
 * which: 0 = exp, 1 = log, 2 = cube (r^3), 3 = sqrt, 4 = x/y with y = in2 (in2 nullable otherwise) */
int nxc_math_batch(nxc_handle *h, int which, int64_t n, const double *in, const double *in2,
                   double *out);

#ifdef __cplusplus
}
#endif
#endif
