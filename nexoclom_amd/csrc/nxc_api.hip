// nxc_api.hip -- C ABI (include/nexoclom_hip.h) over the gfx950 kernels of nxc_kernels.hpp.
//
// Host side of the boundary: owns the device, one stream, the packed table blob, the resident
// packet set and the resident image pair; translates nxc_forces / nxc_image_desc into kernel
// arguments; never throws across the boundary.  RCCL is loaded lazily with dlopen so that the
// library loads (and every single-GPU call works) where librccl is absent.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/nexoclom_hip.h"
#include "nxc_kernels.hpp"
#include "nxc_log_table.hpp"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg)
{
    g_error = msg;
    return code;
}

// No C++ exception may cross the C ABI: entry points that allocate on the host run inside this.
template <class F>
int guarded(F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(NXC_ERR_ARG, "out of host memory");
    } catch (const std::exception &e) {
        return fail(NXC_ERR_ARG, std::string("unexpected C++ exception: ") + e.what());
    } catch (...) {
        return fail(NXC_ERR_ARG, "unexpected C++ exception");
    }
}

// A wait that ended because a collective missed its deadline (stream_sync below) leaves its
// verdict here; the HIP-error path then reports NXC_ERR_RCCL with that text instead of the
// stand-in hipError_t.
thread_local std::string g_coll_failure;

int fail_hip(const char *expr, hipError_t e)
{
    if (!g_coll_failure.empty()) {
        std::string msg;
        msg.swap(g_coll_failure);
        return fail(NXC_ERR_RCCL, msg);
    }
    // out of device memory has a status of its own: callers split the work on it (Input.run)
    return fail(e == hipErrorOutOfMemory ? NXC_ERR_NOMEM : NXC_ERR_HIP,
                std::string(expr) + ": " + hipGetErrorString(e));
}

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail_hip(#expr, e_);                                    \
    } while (0)

constexpr int BLOCK_PERSIST = NXC_BLOCK_PERSIST;
#ifndef NXC_VAR_FAIR_PACKETS_PER_LANE_N       // (tools/gpu_exp_var_forms.sh)
#define NXC_VAR_FAIR_PACKETS_PER_LANE_N 24
#endif
constexpr double NXC_VAR_FAIR_PACKETS_PER_LANE = NXC_VAR_FAIR_PACKETS_PER_LANE_N;
static_assert(NXC_DEV_MAX_MOONS == NXC_MAX_MOONS, "device / ABI moon capacity");

// ---- RCCL, resolved at first use ------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t,
                              ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;

int rccl_load()
{
    if (g_rccl.ok) return NXC_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return fail(NXC_ERR_RCCL, std::string("dlopen(librccl): ") + dlerror());
#define SYM(field, name)                                                                     \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name));        \
    if (!g_rccl.field) return fail(NXC_ERR_RCCL, std::string("dlsym ") + name + " failed");
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(AllReduce, "ncclAllReduce")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(CommAbort, "ncclCommAbort")
    SYM(CommGetAsyncError, "ncclCommGetAsyncError")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl.ok = true;
    return NXC_OK;
}

#define NCCLCHK(expr)                                                                        \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess)                                                               \
            return fail(NXC_ERR_RCCL, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

// ---- packed lookup table ----------------------------------------------------------------------
struct PackedLut {
    LutDesc desc{};
    std::vector<unsigned char> bytes;
};
struct LosLutCache {
    int64_t n = 0;
    std::vector<double> v, g;
    PackedLut lut;
};

// The device's cell computation (nxc_device.hpp: lut_cell), operation for operation: two roundings,
// a saturating conversion, a clamp.
int lut_cell_host(double x, double xbase, double inv_w, int ncell)
{
    const double s = (x - xbase) * inv_w;
    long long c;
    if (!(s >= 0.0)) c = 0;                       // negatives and NaN
    else if (s >= 2147483647.0) c = 2147483647ll;
    else c = (long long)s;
    return (int)std::min<long long>(c, ncell + 1);
}

// Doubles as ordered unsigned integers (monotone for finite values and the infinities).
unsigned long long ordered_key(double v)
{
    unsigned long long b;
    std::memcpy(&b, &v, sizeof b);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
double from_ordered_key(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double v;
    std::memcpy(&v, &b, sizeof v);
    return v;
}

int pack_lut(const double *xp, const double *fp, int64_t n, PackedLut &out, const char *what,
             int cells_per_node = 4)
{
    if (n < 2 || n > 65000 || !xp || !fp)
        return fail(NXC_ERR_ARG, std::string(what) + ": table needs 2..65000 points");
    for (int64_t j = 0; j + 1 < n; j++)
        if (!(xp[j + 1] > xp[j]))
            return fail(NXC_ERR_ARG, std::string(what) + ": abscissae must be strictly ascending");
    for (int64_t j = 0; j < n; j++)
        if (!std::isfinite(xp[j]) || !std::isfinite(fp[j]))
            return fail(NXC_ERR_ARG, std::string(what) + ": table values must be finite");
    int ncell = 64;
    while (ncell < cells_per_node * n && ncell < 16384) ncell <<= 1;
    const int64_t rows = n + 2;
    const size_t cell_bytes = ((size_t)(ncell + 2) * sizeof(unsigned short) + 31) & ~size_t(31);
    out.bytes.assign((size_t)rows * 32 + cell_bytes, 0);
    double *rec = reinterpret_cast<double *>(out.bytes.data());     // {xp, xn} pairs
    double *fs = rec + 2 * rows;                                    // {fp, slope} pairs
    rec[0] = -std::numeric_limits<double>::max(); rec[1] = xp[0];   // row 0: below the table
    fs[0] = fp[0]; fs[1] = 0.0;
    for (int64_t j = 0; j < n; j++) {
        const bool last = j + 1 == n;
        rec[2 * (j + 1)] = xp[j];
        rec[2 * (j + 1) + 1] = last ? HUGE_VAL : xp[j + 1];
        fs[2 * (j + 1)] = fp[j];
        // np.interp's slope, the same IEEE quotient
        fs[2 * (j + 1) + 1] = last ? 0.0 : (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
        if (!std::isfinite(fs[2 * (j + 1) + 1]))
            return fail(NXC_ERR_ARG, std::string(what) + ": table slopes must be finite");
    }
    rec[2 * (n + 1)] = HUGE_VAL; rec[2 * (n + 1) + 1] = HUGE_VAL;   // sentinel: never selected
    fs[2 * (n + 1)] = fp[n - 1]; fs[2 * (n + 1) + 1] = 0.0;
    const double x0 = xp[0], xl = xp[n - 1];
    const double inv_w = (double)ncell / (xl - x0);
    const double xbase = x0 - (xl - x0) / (double)ncell;
    if (!std::isfinite(inv_w) || !std::isfinite(xbase) || !(xbase < x0))
        return fail(NXC_ERR_ARG, std::string(what) + ": abscissa range cannot be indexed");
    // cell[c] = the last row whose xp <= the smallest double that the device maps to cell c
    // (bisection over the ordered doubles with the device's own arithmetic; cell 0 starts at
    // -inf).  Row xp's: row 0 = -DBL_MAX, row j + 1 = xp[j].
    unsigned short *cell = reinterpret_cast<unsigned short *>(rec + 4 * rows);
    cell[0] = 0;
    const unsigned long long k_lo = ordered_key(-std::numeric_limits<double>::max());
    const unsigned long long k_hi = ordered_key(std::numeric_limits<double>::max());
    for (int c = 1; c <= ncell + 1; c++) {
        if (lut_cell_host(from_ordered_key(k_hi), xbase, inv_w, ncell) < c)
            return fail(NXC_ERR_ARG, std::string(what) + ": cell index is not onto");
        unsigned long long lo = k_lo, hi = k_hi;   // cell(lo) < c <= cell(hi)
        while (hi - lo > 1) {
            const unsigned long long mid = lo + (hi - lo) / 2;
            if (lut_cell_host(from_ordered_key(mid), xbase, inv_w, ncell) >= c) hi = mid;
            else lo = mid;
        }
        const double first = from_ordered_key(hi);
        const int64_t j = std::upper_bound(xp, xp + n, first) - xp;  // nodes with xp <= first
        cell[c] = (unsigned short)j;                                  // row j = node j - 1 (0: below)
    }
    out.desc.rec = 0;                               // offsets relative to the table: placed_lut()
    out.desc.fs = 16 * (int)rows;
    out.desc.cell = 32 * (int)rows;
    out.desc.top = ncell + 1;
    out.desc.last = (int)n + 1;
    out.desc.xbase = xbase;
    out.desc.inv_w = inv_w;
    return NXC_OK;
}

// The descriptor of a packed table that starts at byte `base` of the LDS block.
LutDesc placed_lut(LutDesc d, size_t base)
{
    d.rec += (int)base; d.fs += (int)base; d.cell += (int)base;
    return d;
}

// Affine maps of NumPy's PCG64 for the device sampler (nxc_kernels.hpp: PcgK): entry b <
// NXC_PCG_BITS advances 2^b steps, entry NXC_PCG_BITS + v advances v*n steps (the start of draw
// vector v).  state' = mult^d state + inc (mult^d - 1)/(mult - 1), accumulated by squaring like
// pcg_advance_lcg_128 (numpy/random/src/pcg64/pcg64.c).
typedef unsigned __int128 u128;
const u128 PCG_MULT = ((u128)2549297995355413924ULL << 64) | 4865540595714422341ULL;

void pcg_advance_map(u128 delta, u128 inc, u128 *a_out, u128 *c_out)
{
    u128 acc_mult = 1, acc_plus = 0, cur_mult = PCG_MULT, cur_plus = inc;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    *a_out = acc_mult;
    *c_out = acc_plus;
}

std::vector<u128> pcg_tables(u128 inc, int64_t n)
{
    std::vector<u128> t((size_t)2 * (NXC_PCG_BITS + NXC_PCG_VECS));
    for (int b = 0; b < NXC_PCG_BITS; b++) pcg_advance_map((u128)1 << b, inc, &t[2 * b], &t[2 * b + 1]);
    for (int v = 0; v < NXC_PCG_VECS; v++)
        pcg_advance_map((u128)v * (u128)n, inc, &t[2 * (NXC_PCG_BITS + v)], &t[2 * (NXC_PCG_BITS + v) + 1]);
    return t;
}

// Largest double x with sqrt(x) <= e (host sqrt is correctly rounded): r2 > x <=> sqrt(r2) > e,
// the constant driver's escape test (Output.py:395,410) without a device square root.
double sqrt_threshold(double e)
{
    if (!(e > 0) || !std::isfinite(e)) return e;
    double x = e * e;
    if (!std::isfinite(x)) return HUGE_VAL;
    while (std::sqrt(x) > e) x = std::nextafter(x, 0.0);
    while (std::sqrt(std::nextafter(x, HUGE_VAL)) <= e) x = std::nextafter(x, HUGE_VAL);
    return x;
}

// h*a[n+1][i] for a launch-uniform step: the products NumPy forms per packet (rk5.py:33).
StepW make_stepw(double h)
{
    StepW w{};
    w.h = h;
    for (int n = 0; n < 6; n++)
        for (int i = 0; i <= n; i++) w.w[n * (n + 1) / 2 + i] = h * Tableau::A[n + 1][i];
    return w;
}

}  // namespace

// Device-resident compact trajectory rows (nxc_rows_build): nine value columns [9][total] and the
// packet-index column, as float32/int32 (what save() stores, Output.py:528-543) or float64/int64.
struct nxc_rows {
    int device = 0;
    bool f32 = false;
    long long total = 0;
    void *d_cols = nullptr;
    void *d_index = nullptr;
    size_t cols_cap = 0, index_cap = 0;      // bytes of the two blocks (they may come from the pool)
};

// Device blocks of freed row stores, kept for the next store.  hipMalloc / hipFree themselves return
// in under a millisecond on this driver (now and then 30-150 ms beside a running kernel); what a
// fresh block costs is its first touch -- a 51 GB catalogue written into new memory takes about
// 0.1 s longer than into pooled blocks (NXC_POOL_TRACE=1 prints every take and give;
// tools/gpu_exp_run_breakdown.py).  The pool counts as free memory (nxc_mem_info) and is emptied
// before anything is refused for lack of it.
struct BlockPool {
    struct Block { void *p; size_t bytes; };
    std::vector<Block> blocks;
    size_t bytes = 0;
    std::mutex lock;                         // stores are freed by whichever thread drops them
};

struct nxc_handle {
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // row-store downloads, streamed uploads: beside the compute stream
    hipStream_t stream2 = nullptr;       // streamed pass: the pieces' ordering kernels
    unsigned long long *d_piece_hist = nullptr;   // streamed pass: sort bins (+ max) per piece, then the
                                                  // word that publishes the queue positions ready
    hipEvent_t ev_piece[33] = {};        // ... piece p uploaded; [32]: start / join
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    char name[256] = {0};

    // tables
    bool have_forces = false, have_image = false;
    ForceK F{};
    ImageK G{};
    PackedLut force_lut;
    std::vector<double> force_v, force_a;        // raw table (re-packed with fewer cells if LDS is short)
    int force_cells_per_node = 16;
    int force_lut_cells = 0;                     // cells per node force_lut was packed with
    std::vector<unsigned char> image_part;       // lines, then xedges, zedges
    LdsHeader header{};                          // host copy of the blob's first bytes
    LutDesc line_local[NXC_MAX_LINES]{};         // relative to the table's own start ...
    size_t line_start[NXC_MAX_LINES]{};          // ... which sits at this offset of image_part
    int64_t xedges_local = 0, zedges_local = 0;
    unsigned char *d_blob = nullptr;
    size_t blob_cap = 0, force_bytes = 0, all_bytes = 0;
    unsigned char *d_blob_img = nullptr;         // [LdsHeader | image part] for the kernels that bin
    size_t blob_img_cap = 0, img_bytes = 0;      // stored samples: no force table in their LDS

    // resident data
    double *d_image = nullptr;       // interleaved {weight sum, packet count} per pixel, fp64
    size_t npix = 0;
    double *d_packets = nullptr;
    size_t packets_cap = 0;
    int64_t n_packets = 0;
    unsigned *d_order = nullptr;     // packet indices by decreasing launch speed (queue order)
    double *d_queue = nullptr;       // the packets in that order, one 64-byte record each: [n][8]
    size_t queue_cap = 0;
    size_t order_cap = 0;
    bool have_order = false;
    int order_key = 0;               // what the queue is sorted by: 0 |v|^2, 1 known lifetimes, 2 the adaptive driver's key
    int64_t first_id = 0;            // global index of resident packet 0 (RNG counter space)
    bool have_bounce = false;
    double *d_bounce = nullptr;      // spline knots + coefficients of the accommodation table
    size_t bounce_cap = 0;
    double *d_source = nullptr;      // sampler tables: speed CDF + speeds, surface density map
    size_t source_cap = 0;
    DevCounters *d_ctr = nullptr;
    double *d_scratch = nullptr;     // final states / generic device scratch
    size_t scratch_cap = 0;
    unsigned char *d_samples = nullptr;   // host sample columns on their way to k_image / k_los
    size_t samples_cap = 0;
    unsigned char *d_tiles = nullptr;     // chunk scratch of the tiled image (k_image_bin -> k_image_tiles)
    size_t tiles_cap = 0;
    unsigned char *d_losblk = nullptr;    // block descriptors + spheres (k_los_blocks -> k_los)
    size_t losblk_cap = 0;
    int image_mode = 0;                   // nxc_image_mode: 0 by size, 1 k_image, 2 tiles
    int tile_pixels = NXC_TILE_PIXELS;
    int64_t tile_slab = int64_t(1) << 28; // samples per pass of the tiled image (bounds its scratch)
    long long *d_steps = nullptr;
    size_t steps_cap = 0;
    unsigned long long *d_hist = nullptr;   // counting-sort bins of the queue order
    size_t hist_cap = 0;
    void *d_rec = nullptr;           // pass 2's records on their way to columns (grow-only)
    size_t rec_cap = 0;

    BlockPool pool;
    LosLutCache los_lut[NXC_MAX_LINES];   // the g-value tables of the last line-of-sight call, packed

    // compact-rows protocol (nxc_integrate_const_rows -> nxc_rows_fetch)
    long long *d_offsets = nullptr;
    size_t offsets_cap = 0;
    long long rows_total = -1;
    double rows_step = 0, rows_edge = 0;
    int64_t rows_n_iter = 0, rows_n = 0;

    bool have_bodies = false;
    nxc_bodies_desc bodies{};
    double *d_moonpos = nullptr;     // [n_iter][n_moons][cos, sin] of the per-step base phase
    size_t moonpos_cap = 0;

    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    bool streamed_pending = false;   // nxc_integrate_const_streamed since the last nxc_synchronize
    bool coll_pending = false;       // a collective sits on `stream` and nobody has waited for it yet
    hipEvent_t ev_pre_coll = nullptr; // recorded in front of the first pending collective
    std::atomic<bool> abort_requested{false};   // nxc_comm_request_abort (any thread)
    double coll_timeout_s = 120.0;   // nxc_comm_set_timeout / NXC_COLLECTIVE_TIMEOUT_S
    double *d_reduce = nullptr;      // one double for control-plane reductions
    double *d_reduce_n = nullptr;    // nxc_allreduce_f64's staging (grow-only)
    size_t reduce_n_cap = 0;
};

static int order_on_device(nxc_handle *h, double k2max, const long long *d_lifetimes,
                           int64_t max_steps, bool flight_key = false);

namespace {

// Wait for the handle's stream.  Without a collective in flight this is hipStreamSynchronize.
// With one (nxc_image_allreduce / nxc_allreduce_* since the last wait) the wait is bounded: the
// stream is polled together with ncclCommGetAsyncError, and when `coll_timeout_s` have passed --
// a peer rank died or never issued its half -- the communicator is aborted (ncclCommAbort makes
// the RCCL kernel leave), the handle is left without a communicator and the caller gets
// NXC_ERR_RCCL (through fail_hip) instead of waiting forever.
hipError_t stream_sync(nxc_handle *h)
{
    if (!h->coll_pending || !h->comm || !g_rccl.ok) {
        h->coll_pending = false;
        return hipStreamSynchronize(h->stream);
    }
    // what was queued before the collective is this process's own work: it ends by itself, and is
    // waited for the ordinary way (no polling beside a 40 ms kernel); the deadline runs from there
    if (h->ev_pre_coll) {
        const hipError_t e = hipEventSynchronize(h->ev_pre_coll);
        if (e != hipSuccess) { h->coll_pending = false; return e; }
    }
    timespec t0{};
    clock_gettime(CLOCK_MONOTONIC, &t0);
    auto elapsed = [&]() {
        timespec t{};
        clock_gettime(CLOCK_MONOTONIC, &t);
        return double(t.tv_sec - t0.tv_sec) + 1e-9 * double(t.tv_nsec - t0.tv_nsec);
    };
    std::string why;
    for (long spin = 0;; spin++) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipSuccess) {
            h->coll_pending = false;
            return hipSuccess;
        }
        if (q != hipErrorNotReady) {
            h->coll_pending = false;
            return q;
        }
        if (h->abort_requested.load(std::memory_order_relaxed)) {
            why = "a peer rank reported a failure while this rank waited for a collective";
            break;
        }
        if ((spin & 63) == 63) {
            ncclResult_t async = ncclSuccess;
            const ncclResult_t r = g_rccl.CommGetAsyncError(h->comm, &async);
            if (r != ncclSuccess || (async != ncclSuccess && async != ncclInProgress)) {
                why = std::string("RCCL reported an asynchronous error (") +
                      g_rccl.GetErrorString(r != ncclSuccess ? r : async) + ")";
                break;
            }
            const double dt = elapsed();
            if (dt > h->coll_timeout_s) {
                char buf[160];
                std::snprintf(buf, sizeof buf,
                              "a collective of rank %d of %d did not complete within %.1f s (a peer "
                              "rank is gone or never joined it)", h->rank, h->nranks, h->coll_timeout_s);
                why = buf;
                break;
            }
            if (dt > 2e-3) {                       // past the latency of a healthy collective
                timespec nap{0, 200000};
                nanosleep(&nap, nullptr);
            }
        }
    }
    (void)g_rccl.CommAbort(h->comm);               // frees the communicator; its kernel exits
    h->comm = nullptr;
    h->nranks = 1;
    h->rank = 0;
    h->coll_pending = false;
    g_coll_failure = why + "; the communicator was aborted";
    return hipErrorNotReady;
}

void flush_all_pools();

int ensure(void **ptr, size_t *cap, size_t bytes)
{
    if (*cap >= bytes && *ptr) return NXC_OK;
    if (*ptr) HIPCHK(hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    if (hipMalloc(ptr, bytes ? bytes : 8) != hipSuccess) {
        (void)hipGetLastError();
        flush_all_pools();                 // memory kept from freed row stores goes back first
        HIPCHK(hipMalloc(ptr, bytes ? bytes : 8));
    }
    *cap = bytes;
    return NXC_OK;
}

// NXC_POOL_TRACE=1: every take / give that reaches the driver, with its duration, on stderr
const bool g_pool_trace = std::getenv("NXC_POOL_TRACE") != nullptr;
size_t pool_bytes(nxc_handle *h);

void pool_flush(nxc_handle *h)
{
    std::lock_guard<std::mutex> g(h->pool.lock);
    for (const auto &b : h->pool.blocks) (void)hipFree(b.p);
    h->pool.blocks.clear();
    h->pool.bytes = 0;
}

// A block of at least `bytes`: the best fit in the pool that wastes less than a quarter, else new
// memory (the pool is given back to the driver before an allocation is allowed to fail).
hipError_t pool_take(nxc_handle *h, size_t bytes, void **out, size_t *cap)
{
    {
        std::lock_guard<std::mutex> g(h->pool.lock);
        int best = -1;
        for (int k = 0; k < (int)h->pool.blocks.size(); k++) {
            const size_t have = h->pool.blocks[(size_t)k].bytes;
            if (have >= bytes && have - bytes <= bytes / 4 &&
                (best < 0 || have < h->pool.blocks[(size_t)best].bytes))
                best = k;
        }
        if (best >= 0) {
            *out = h->pool.blocks[(size_t)best].p;
            *cap = h->pool.blocks[(size_t)best].bytes;
            h->pool.bytes -= *cap;
            h->pool.blocks.erase(h->pool.blocks.begin() + best);
            if (g_pool_trace) std::fprintf(stderr, "[nxc pool] hit  %8.1f MB in a block of %8.1f MB\n", bytes / 1e6, *cap / 1e6);
            return hipSuccess;
        }
    }
    timespec t0{}, t1{};
    if (g_pool_trace) clock_gettime(CLOCK_MONOTONIC, &t0);
    hipError_t e = hipMalloc(out, bytes ? bytes : 8);
    if (g_pool_trace) {
        clock_gettime(CLOCK_MONOTONIC, &t1);
        std::fprintf(stderr, "[nxc pool] miss %8.1f MB: hipMalloc %.1f ms (pool holds %.1f MB in %zu blocks)\n", bytes / 1e6,
                     (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6, pool_bytes(h) / 1e6, h->pool.blocks.size());
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pool_flush(h);
        e = hipMalloc(out, bytes ? bytes : 8);
    }
    *cap = e == hipSuccess ? bytes : 0;
    return e;
}

void pool_give(nxc_handle *h, void *p, size_t bytes)
{
    if (!p) return;
    size_t free_b = 0, total_b = 0;
    const bool known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    std::lock_guard<std::mutex> g(h->pool.lock);
    // small blocks are cheap to allocate; the pool never holds more than a third of the device
    if (!known || bytes < (size_t(64) << 20) || h->pool.bytes + bytes > total_b / 3 ||
        h->pool.blocks.size() >= 64) {
        timespec t0{}, t1{};
        if (g_pool_trace) clock_gettime(CLOCK_MONOTONIC, &t0);
        (void)hipFree(p);
        if (g_pool_trace && bytes >= (size_t(64) << 20)) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            std::fprintf(stderr, "[nxc pool] full: hipFree of %8.1f MB %.1f ms\n", bytes / 1e6,
                         (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
        }
        return;
    }
    h->pool.blocks.push_back({p, bytes});
    h->pool.bytes += bytes;
}

// every live handle, for allocations that have no handle at hand (ensure)
std::mutex g_handles_lock;
std::vector<nxc_handle *> g_handles;

void flush_all_pools()
{
    std::lock_guard<std::mutex> g(g_handles_lock);
    for (nxc_handle *h : g_handles) pool_flush(h);
}

size_t pool_bytes(nxc_handle *h)
{
    std::lock_guard<std::mutex> g(h->pool.lock);
    return h->pool.bytes;
}

// Device blob = [LdsHeader | force table | g-value tables | x edges | z edges]; kernels stage
// the first force_bytes (integrator only) or all_bytes (image) of it into LDS.
int upload_blob(nxc_handle *h)
{
    const size_t hb = NXC_HEADER_BYTES;
    const size_t ib = h->have_image ? h->image_part.size() : 0;
    const size_t limit = 160 * 1024 - 32 - (size_t)(BLOCK_PERSIST / 64) * NXC_WAVE_LDS_BYTES;
    // (the row-writing kernels stage two more words per queued packet and no image tables)
    const size_t limit_rows = 160 * 1024 - 32 - (size_t)(BLOCK_PERSIST / 64) * NXC_WAVE_LDS_BYTES_ROWS;
    size_t fb = h->have_forces ? h->force_lut.bytes.size() : 0;
    while (h->have_forces && (hb + fb + ib > limit || hb + fb > limit_rows) &&
           h->force_cells_per_node > 2) {
        h->force_cells_per_node /= 2;          // trade lookup hit rate for LDS space
        PackedLut lut;
        int rc2 = pack_lut(h->force_v.data(), h->force_a.data(), (int64_t)h->force_v.size(), lut,
                           "nxc_forces radiation table", h->force_cells_per_node);
        if (rc2) return rc2;
        h->force_lut = std::move(lut);
        h->force_lut_cells = h->force_cells_per_node;
        fb = h->force_lut.bytes.size();
    }
    h->force_bytes = hb + fb;
    h->all_bytes = hb + fb + ib;
    if (h->all_bytes > limit || h->force_bytes > limit_rows)
        return fail(NXC_ERR_ARG, "lookup tables exceed the 160 KiB LDS of a gfx950 CU");
    int rc = ensure(reinterpret_cast<void **>(&h->d_blob), &h->blob_cap, h->all_bytes);
    if (rc) return rc;
    h->F.tab = placed_lut(h->force_lut.desc, hb);
    if (h->have_image) {
        for (int l = 0; l < h->G.n_lines; l++) {
            h->G.line[l] = placed_lut(h->line_local[l], hb + fb + h->line_start[l]);
        }
        h->G.xedges_off = h->xedges_local + (int64_t)(hb + fb);
        h->G.zedges_off = h->zedges_local + (int64_t)(hb + fb);
    }
    h->header.G = h->G;
    HIPCHK(hipMemcpyAsync(h->d_blob, &h->header, sizeof(LdsHeader), hipMemcpyHostToDevice,
                          h->stream));
    if (fb) HIPCHK(hipMemcpyAsync(h->d_blob + hb, h->force_lut.bytes.data(), fb,
                                  hipMemcpyHostToDevice, h->stream));
    if (ib) HIPCHK(hipMemcpyAsync(h->d_blob + hb + fb, h->image_part.data(), ib,
                                  hipMemcpyHostToDevice, h->stream));
    HIPCHK(stream_sync(h));
    if (h->have_image) {
        // the same image tables right behind the header: k_image stages 45 KB instead of 100+, so
        // that several of its workgroups fit a CU (it was one 256-thread group per CU)
        LdsHeader hdr = h->header;
        for (int l = 0; l < h->G.n_lines; l++)
            hdr.G.line[l] = placed_lut(h->line_local[l], hb + h->line_start[l]);
        hdr.G.xedges_off = h->xedges_local + (int64_t)hb;
        hdr.G.zedges_off = h->zedges_local + (int64_t)hb;
        h->img_bytes = hb + ib;
        rc = ensure(reinterpret_cast<void **>(&h->d_blob_img), &h->blob_img_cap, h->img_bytes);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(h->d_blob_img, &hdr, sizeof(LdsHeader), hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_blob_img + hb, h->image_part.data(), ib, hipMemcpyHostToDevice,
                              h->stream));
        HIPCHK(stream_sync(h));      // hdr is a local
    }
    return NXC_OK;
}

// Put this launch's step weights into the header (stream-ordered before the kernel).
int upload_step(nxc_handle *h, double step)
{
    h->header.W = make_stepw(step);
    HIPCHK(hipMemcpyAsync(h->d_blob + offsetof(LdsHeader, W), &h->header.W, sizeof(StepW),
                          hipMemcpyHostToDevice, h->stream));
    return NXC_OK;
}

template <class K>
int prep_kernel(K kernel, size_t lds_bytes)
{
    if (lds_bytes > 64 * 1024)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    return NXC_OK;
}

// Grid of a persistent kernel: one resident block per CU slot.  When there are fewer packets than
// lanes (one reference-sized chunk of 80 467 packets against 196 608 lanes), the packets are
// spread over ALL CUs with fewer waves per block instead of filling a few CUs three waves deep: a
// packet's steps run one after the other, and a wave that has its SIMD to itself takes them
// faster.  *block comes back as the number of threads to launch per block (a multiple of 64).
template <class K>
int persistent_grid(nxc_handle *h, K kernel, int *block, size_t lds_bytes, int64_t n, int *grid)
{
    int per_cu = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, *block, lds_bytes));
    if (per_cu < 1) per_cu = 1;
    int64_t g = (int64_t)h->n_cu * per_cu;
    const int64_t waves = (n + 63) / 64;
    if (waves < g * (*block / 64)) {
        int64_t per_block = (waves + g - 1) / g;
        if (per_block < 1) per_block = 1;
        *block = (int)per_block * 64;
        g = (waves + per_block - 1) / per_block;
    }
    if (g < 1) g = 1;
    *grid = (int)g;
    return NXC_OK;
}

int flat_grid(nxc_handle *h, int64_t n, int block)
{
    int64_t g = (n + block - 1) / block;
    const int64_t cap = (int64_t)h->n_cu * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

int begin_timed(nxc_handle *h)
{
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    return NXC_OK;
}
int end_timed(nxc_handle *h)
{
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return NXC_OK;
}

int need_forces(nxc_handle *h)
{
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    if (!h->have_forces) return fail(NXC_ERR_STATE, "nxc_set_forces has not been called");
    HIPCHK(hipSetDevice(h->device));
    return NXC_OK;
}

size_t persist_lds(size_t table_bytes)
{
    return ((table_bytes + 31) & ~size_t(31)) + (size_t)(BLOCK_PERSIST / 64) * NXC_WAVE_LDS_BYTES;
}

// Per-step base phases of the moons, (cos, sin)(phi - omega (t0 - k h)), and the per-stage rotations
// (cos, sin)(omega c_n h) in the header.  Host libm sincos(); the oracle builds the same numbers
// the same way.
int upload_moon_table(nxc_handle *h, double step, int64_t n_iter)
{
    static const double cn[6] = {0, 0.2, 0.3, 0.8, 8. / 9., 1.};
    const nxc_bodies_desc &b = h->bodies;
    const int nm = b.n_moons;
    BodyK &K = h->header.Bd;
    for (int n = 0; n < 6; n++)
        for (int m = 0; m < nm; m++) {
            double sn, cs;
            ::sincos(b.omega[m] * (cn[n] * step), &sn, &cs);
            K.cd[n][m] = cs; K.sd[n][m] = sn;
        }
    HIPCHK(hipMemcpyAsync(h->d_blob + offsetof(LdsHeader, Bd), &h->header.Bd, sizeof(BodyK),
                          hipMemcpyHostToDevice, h->stream));
    const size_t count = (size_t)(n_iter > 0 ? n_iter : 1) * 2 * (size_t)(nm > 0 ? nm : 1);
    std::vector<double> base(count, 0.0);
    for (int64_t k = 0; k < n_iter; k++) {
        const double t = b.t0 - (double)k * step;
        for (int m = 0; m < nm; m++) {
            double sn, cs;
            ::sincos(b.phi[m] - b.omega[m] * t, &sn, &cs);
            base[((size_t)k * nm + m) * 2] = cs;
            base[((size_t)k * nm + m) * 2 + 1] = sn;
        }
    }
    int rc = ensure(reinterpret_cast<void **>(&h->d_moonpos), &h->moonpos_cap,
                    count * sizeof(double));
    if (rc) return rc;
    // pageable source: the copy is staged before the call returns
    HIPCHK(hipMemcpyAsync(h->d_moonpos, base.data(), count * sizeof(double), hipMemcpyHostToDevice,
                          h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
}

// What one launch of the persistent integrator works on: the resident set (default) or a piece of
// it with its own queue copy, counters (the queue head lives there) and stream.
struct FusedJob {
    const double *soa = nullptr;
    const unsigned *order = nullptr;
    int64_t n = 0, first_id = 0;
    DevCounters *ctr = nullptr;
    hipStream_t stream = nullptr;
    bool timed = true;
    const unsigned long long *avail = nullptr;   // streamed upload: positions published so far
};

FusedJob whole_set(nxc_handle *h)
{
    FusedJob j;
    j.soa = h->have_order ? h->d_queue : h->d_packets;
    j.order = h->have_order ? h->d_order : nullptr;
    j.n = h->n_packets; j.first_id = h->first_id; j.ctr = h->d_ctr; j.stream = h->stream;
    return j;
}

template <int IMAGE, bool BOUNCE, bool FULL = false, bool NBODY = false, bool STREAMED = false>
int launch_fused(nxc_handle *h, size_t tables, size_t lds, int64_t n_iter, double edge2,
                 double *d_final, long long *d_steps, const FusedJob &job)
{
    int grid = 1, block = BLOCK_PERSIST, rc;
    auto kernel = k_const_fused<IMAGE, BOUNCE, FULL, NBODY, 0, STREAMED>;
    if ((rc = prep_kernel(kernel, lds))) return rc;
    if ((rc = persistent_grid(h, kernel, &block, lds, job.n, &grid))) return rc;
    if (job.timed && (rc = begin_timed(h))) return rc;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, job.stream, h->F, h->d_blob,
                       (int64_t)tables, job.n, job.soa, job.order, job.first_id, n_iter,
                       edge2, d_final, d_steps, IMAGE ? h->d_image : (double *)nullptr, job.ctr,
                       NBODY ? h->d_moonpos : (const double *)nullptr, (const long long *)nullptr,
                       (void *)nullptr, job.avail);
    HIPCHK(hipGetLastError());
    return job.timed ? end_timed(h) : NXC_OK;
}

// The kernel variant for the handle's force model / re-emission / moons.  Image: 1 = samples
// binned as they are (64-bit), 2 = binned as the float32 values save() stores (nxc_image_desc.
// downcast_f32): the frame test comes before the compaction queue, everything else after it.
int pick_fused(nxc_handle *h, size_t tables, size_t lds, int64_t n_iter, double edge2, bool image,
               double *d_final, long long *d_steps, const FusedJob &job)
{
    const int img = image ? (h->G.downcast_f32 ? 2 : 1) : 0;
#define NXC_FUSED(...)                                                                              \
    (img == 2 ? launch_fused<2, __VA_ARGS__>(h, tables, lds, n_iter, edge2, d_final, d_steps, job)  \
     : img == 1 ? launch_fused<1, __VA_ARGS__>(h, tables, lds, n_iter, edge2, d_final, d_steps, job) \
                : launch_fused<0, __VA_ARGS__>(h, tables, lds, n_iter, edge2, d_final, d_steps, job))
    // gravity + radiation pressure + photo-loss: the compile-time specialisation of the force model
    const bool full = h->F.grav && h->F.rad && h->F.loss == LOSS_PHOTO;
    if (job.avail)          // streamed upload: the plain force models only (checked by the caller)
        return full ? NXC_FUSED(false, true, false, true) : NXC_FUSED(false, false, false, true);
    if (h->have_bodies) return full ? NXC_FUSED(false, true, true) : NXC_FUSED(false, false, true);
    if (h->have_bounce) return NXC_FUSED(true, false);
    return full ? NXC_FUSED(false, true) : NXC_FUSED(false, false);
#undef NXC_FUSED
}

// The constant-step kernels want the fastest (longest-lived) packets first: a resident set that the
// adaptive driver has re-sorted for itself goes back to that order.
int speed_order_for_const(nxc_handle *h)
{
    if (h->order_key != 2 || h->n_packets < 2) return NXC_OK;
    return order_on_device(h, -1.0, nullptr, 0);
}

int launch_const(nxc_handle *h, double step, int64_t n_iter, double outeredge, bool image,
                 double *d_final, long long *d_steps)
{
    const size_t tables = image ? h->all_bytes : h->force_bytes;
    const size_t lds = persist_lds(tables);
    int rc;
    if ((rc = upload_step(h, step))) return rc;
    HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(DevCounters), h->stream));
    const double edge2 = sqrt_threshold(outeredge);
    if (h->have_bodies) {
        if (h->have_bounce)
            return fail(NXC_ERR_STATE, "surface re-emission is not available with moons set");
        if ((rc = upload_moon_table(h, step, n_iter))) return rc;
    }
    return pick_fused(h, tables, lds, n_iter, edge2, image, d_final, d_steps, whole_set(h));
}

size_t persist_lds_rows(size_t table_bytes)
{
    return ((table_bytes + 31) & ~size_t(31)) + (size_t)(BLOCK_PERSIST / 64) * NXC_WAVE_LDS_BYTES_ROWS;
}

template <bool BOUNCE, bool FULL, bool NBODY, int ROWS>
int launch_rows(nxc_handle *h, int64_t n_iter, double edge2, void *d_rec)
{
    int grid = 1, block = BLOCK_PERSIST, rc;
    auto kernel = k_const_fused<0, BOUNCE, FULL, NBODY, ROWS>;
    const size_t tables = h->force_bytes, lds = persist_lds_rows(tables);
    if ((rc = prep_kernel(kernel, lds))) return rc;
    if ((rc = persistent_grid(h, kernel, &block, lds, h->n_packets, &grid))) return rc;
    if ((rc = begin_timed(h))) return rc;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, h->stream, h->F, h->d_blob,
                       (int64_t)tables, h->n_packets, h->have_order ? h->d_queue : h->d_packets,
                       h->have_order ? h->d_order : (const unsigned *)nullptr, h->first_id, n_iter,
                       edge2, (double *)nullptr, (long long *)nullptr, (double *)nullptr, h->d_ctr,
                       NBODY ? h->d_moonpos : (const double *)nullptr,
                       (const long long *)h->d_offsets, d_rec, (const unsigned long long *)nullptr);
    HIPCHK(hipGetLastError());
    return end_timed(h);
}

// Pass 1 of every trajectory-producing run: the persistent integrator (optionally binning the
// image) with final states and step counts kept on the device, then the row offsets: records
// 0..k of a packet exist, the last one is live unless the packet died in iteration k.
int count_rows(nxc_handle *h, double step, int64_t n_iter, double outeredge, bool image,
               int64_t *lengths_out, long long *total_out)
{
    const int64_t n = h->n_packets;
    const size_t col = (size_t)n * sizeof(double);
    int rc;
    h->rows_total = -1;
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_scratch), &h->scratch_cap, 8 * col))) return rc;
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_steps), &h->steps_cap,
                     (size_t)n * sizeof(long long))))
        return rc;
    if ((rc = launch_const(h, step, n_iter, outeredge, image, h->d_scratch, h->d_steps))) return rc;
    std::vector<long long> steps((size_t)n), off((size_t)n + 1);
    std::vector<double> frac((size_t)n);
    HIPCHK(hipMemcpyAsync(steps.data(), h->d_steps, (size_t)n * sizeof(long long),
                          hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(frac.data(), h->d_scratch + 7 * n, col, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    long long acc = 0;
    for (int64_t i = 0; i < n; i++) {
        off[(size_t)i] = acc;
        const long long len = steps[(size_t)i] + (frac[(size_t)i] > 0.0 ? 1 : 0);
        if (lengths_out) lengths_out[i] = len;
        acc += len;
    }
    off[(size_t)n] = acc;
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_offsets), &h->offsets_cap,
                     ((size_t)n + 1) * sizeof(long long))))
        return rc;
    HIPCHK(hipMemcpyAsync(h->d_offsets, off.data(), ((size_t)n + 1) * sizeof(long long),
                          hipMemcpyHostToDevice, h->stream));
    HIPCHK(stream_sync(h));
    // the lifetimes are now known exactly: re-sort the queue by them (longest first) for pass 2
    // and for every later pass over these packets
    if ((rc = order_on_device(h, 0.0, h->d_steps, n_iter))) return rc;
    h->rows_total = acc;
    h->rows_step = step; h->rows_edge = outeredge; h->rows_n_iter = n_iter; h->rows_n = n;
    *total_out = acc;
    return NXC_OK;
}

// Pass 2: the records themselves, rec[total][10] (doubles, or floats + int32 when narrow) in the
// handle's record scratch.  `reserve`: bytes the caller is about to allocate next to it (checked
// against free memory).
int write_records(nxc_handle *h, bool narrow, size_t reserve)
{
    int rc = need_forces(h);
    if (rc) return rc;
    if (h->rows_total < 0 || h->rows_n != h->n_packets)
        return fail(NXC_ERR_STATE, "the trajectory rows need a preceding nxc_integrate_const_rows");
    const long long total = h->rows_total;
    if (total == 0) {
        // nothing to write: the counters of "pass 2" are zeros, not a second copy of pass 1's
        HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(DevCounters), h->stream));
        return NXC_OK;
    }
    const size_t bytes = (size_t)total * 10 * (narrow ? sizeof(float) : sizeof(double));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t grow = bytes > h->rec_cap ? bytes : 0, back = grow ? h->rec_cap : 0;
    if (grow + reserve > free_b + back + pool_bytes(h))
        return fail(NXC_ERR_NOMEM, "trajectory rows do not fit in device memory; run fewer packets "
                                 "per call (the reference chunks too, Input.py:219-222)");
    if (grow > free_b + back) pool_flush(h);      // the scratch grows into what the pool holds
    // the launch groups of an Input.run differ by a per cent or so: an eighth of slack saves the
    // later ones a free + malloc of ten-odd GB each
    const size_t roomy = bytes + bytes / 8;
    const bool slack = grow && roomy + reserve <= free_b + back;
    if ((rc = ensure(&h->d_rec, &h->rec_cap, slack ? roomy : bytes))) return rc;
    if (h->have_bodies) {
        if (h->have_bounce)
            return fail(NXC_ERR_STATE, "surface re-emission is not available with moons set");
        if ((rc = upload_moon_table(h, h->rows_step, h->rows_n_iter))) return rc;
    }
    if ((rc = upload_step(h, h->rows_step))) return rc;
    HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(DevCounters), h->stream));
    const double edge2 = sqrt_threshold(h->rows_edge);
    const int64_t n_iter = h->rows_n_iter;
    const bool full = h->F.grav && h->F.rad && h->F.loss == LOSS_PHOTO;
#define NXC_ROWS_CASE(B, F, N)                                                                  \
    (narrow ? launch_rows<B, F, N, 2>(h, n_iter, edge2, h->d_rec)                               \
            : launch_rows<B, F, N, 1>(h, n_iter, edge2, h->d_rec))
    if (h->have_bodies) return full ? NXC_ROWS_CASE(false, true, true) : NXC_ROWS_CASE(false, false, true);
    if (h->have_bounce) return NXC_ROWS_CASE(true, false, false);
    return full ? NXC_ROWS_CASE(false, true, false) : NXC_ROWS_CASE(false, false, false);
#undef NXC_ROWS_CASE
}

template <typename T, typename I>
int transpose_rows(nxc_handle *h, const void *d_rec, long long total, void *d_cols, void *d_index)
{
    int64_t g = (total + NXC_TR_ROWS - 1) / NXC_TR_ROWS;
    const int64_t cap = (int64_t)h->n_cu * 16;
    if (g > cap) g = cap;
    hipLaunchKernelGGL((k_rows_transpose<T, I>), dim3((unsigned)g), dim3(NXC_TR_ROWS), 0, h->stream,
                       d_rec, total, static_cast<T *>(d_cols), static_cast<I *>(d_index));
    HIPCHK(hipGetLastError());
    return NXC_OK;
}

// Passes 1 -> 2 -> columns: the device-resident row store of the run counted last.
int rows_build(nxc_handle *h, bool narrow, nxc_rows **out)
{
    *out = nullptr;
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    const long long total = h->rows_total;
    const size_t vsz = narrow ? sizeof(float) : sizeof(double), isz = narrow ? sizeof(int) : sizeof(long long);
    const size_t cols_bytes = (size_t)(total > 0 ? total : 0) * 9 * vsz;
    const size_t idx_bytes = (size_t)(total > 0 ? total : 0) * isz;
    int rc = write_records(h, narrow, cols_bytes + idx_bytes);
    if (rc) return rc;
    h->rows_total = -1;
    nxc_rows *r = new (std::nothrow) nxc_rows();
    if (!r) return fail(NXC_ERR_ARG, "out of host memory");
    r->device = h->device; r->f32 = narrow; r->total = total;
    hipError_t e = hipSuccess;
    if (total > 0) {
        e = pool_take(h, cols_bytes, &r->d_cols, &r->cols_cap);
        if (e == hipSuccess) e = pool_take(h, idx_bytes, &r->d_index, &r->index_cap);
        if (e == hipSuccess)
            rc = narrow ? transpose_rows<float, int>(h, h->d_rec, total, r->d_cols, r->d_index)
                        : transpose_rows<double, long long>(h, h->d_rec, total, r->d_cols, r->d_index);
        if (e == hipSuccess && !rc) e = stream_sync(h);
    }
    if (e != hipSuccess || rc) {
        if (r->d_cols) (void)hipFree(r->d_cols);
        if (r->d_index) (void)hipFree(r->d_index);
        delete r;
        return rc ? rc : fail_hip("rows run", e);
    }
    *out = r;
    return NXC_OK;
}

int rows_check(nxc_handle *h, const nxc_rows *r, int64_t first, int64_t count)
{
    if (!h || !r) return fail(NXC_ERR_ARG, "null argument");
    if (r->device != h->device) return fail(NXC_ERR_ARG, "the rows live on another device");
    if (first < 0 || count < 0 || first + count > r->total)
        return fail(NXC_ERR_ARG, "row range outside the store");
    return NXC_OK;
}

// nxc_rows_fetch / nxc_rows_fetch_f32: build, copy out, free
int rows_fetch(nxc_handle *h, void *rows_out, bool narrow)
{
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    if (h->rows_total < 0 || h->rows_n != h->n_packets)
        return fail(NXC_ERR_STATE, "nxc_rows_fetch needs a preceding nxc_integrate_const_rows");
    if (h->rows_total > 0 && !rows_out) return fail(NXC_ERR_ARG, "rows_out is null");
    nxc_rows *r = nullptr;
    int rc = rows_build(h, narrow, &r);
    if (rc) return rc;
    hipError_t e = hipSuccess;
    if (r->total > 0) {
        e = hipMemcpyAsync(rows_out, r->d_cols, (size_t)r->total * 9 * (narrow ? 4 : 8),
                           hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = stream_sync(h);
    }
    nxc_rows_free(h, r);
    if (e != hipSuccess) return fail_hip("rows copy", e);
    return NXC_OK;
}

// f-1 over stored samples that are already on the device (64-bit, or 32-bit as save() keeps them)
template <typename T, typename I>
int los_run(nxc_handle *h, const nxc_los_desc *d, int64_t S, const double *sc, int64_t P,
            const T *dx, const T *dy, const T *dz, const T *dvy, const T *dfrac, const I *d_index,
            int64_t index_shift, int64_t n_index, double *radiance, int64_t *npackets, uint8_t *included,
            int64_t used_cap, int64_t *used_pairs, int64_t *n_used)
{
    // LDS block: [header space | g-value tables | spectra tile]
    std::vector<unsigned char> blob((size_t)NXC_HEADER_BYTES, 0);
    std::memcpy(blob.data(), &h->header, sizeof(LdsHeader));      // nxc_log's table lives there
    LosK K{};
    K.sin_dphi = d->sin_dphi;
    K.sin_2dphi = d->sin_2dphi;
    K.cos_thr = d->cos_threshold;
    K.cos_thr2_lo = d->cos_threshold * d->cos_threshold * (1.0 - 1e-9);
    K.vrplanet = d->vrplanet;
    K.unit_cm2 = d->unit_cm * d->unit_cm;
    K.t0 = d->ladder[0];
    K.log1p_s_inv = 1.0 / std::log1p(d->sin_dphi);
    K.n_lines = d->n_lines;
    K.n_ladder = (int)d->n_ladder;
    K.index_shift = index_shift;
    K.tan_dphi = std::tan(d->dphi);
    // block culling measures distances along the boresights: they must be unit vectors (the
    // reference's are); anything else switches the culling off, never the exact pair test
    K.cull = (d->dphi > 0 && d->dphi < 1.5 && std::isfinite(K.tan_dphi)) ? 1 : 0;
    for (int64_t i = 0; i < S && K.cull; i++) {
        const double b2 = sc[3 * S + i] * sc[3 * S + i] + sc[4 * S + i] * sc[4 * S + i] +
                          sc[5 * S + i] * sc[5 * S + i];
        if (!(std::fabs(b2 - 1.0) <= 1e-9)) K.cull = 0;
    }
    for (int l = 0; l < d->n_lines; l++) {
        // a run's Outputs share their g-value tables (one aplanet): the packed form of the last
        // call's tables is kept (packing places every cell by bisection: 0.5 ms a table, which for
        // the 125 Outputs of an Input.run(1e7) was a third of LOSResult's time)
        const size_t nb = (size_t)d->line_n[l] * sizeof(double);
        LosLutCache &c = h->los_lut[l];
        const bool same = c.n == d->line_n[l] && d->line_n[l] > 0 && c.v.size() * sizeof(double) == nb &&
                          std::memcmp(c.v.data(), d->line_v[l], nb) == 0 &&
                          std::memcmp(c.g.data(), d->line_g[l], nb) == 0;
        if (!same) {
            c.n = 0;
            int rc = pack_lut(d->line_v[l], d->line_g[l], d->line_n[l], c.lut, "g-value table");
            if (rc) return rc;
            c.v.assign(d->line_v[l], d->line_v[l] + d->line_n[l]);
            c.g.assign(d->line_g[l], d->line_g[l] + d->line_n[l]);
            c.n = d->line_n[l];
        }
        K.line[l] = placed_lut(c.lut.desc, blob.size());
        blob.insert(blob.end(), c.lut.bytes.begin(), c.lut.bytes.end());
    }
    const size_t stage_bytes = blob.size();
    K.tile_off = (int64_t)((stage_bytes + 31) & ~size_t(31));
    // ... | per-wave candidate queues
    // all the spectra of a launch in LDS, 512 at a time -- fewer when the g-value tables are large
    // (four lines of 389 points: 66 KB) and leave less room beside the per-wave lists
    const size_t lds_rest = (size_t)K.tile_off + (size_t)(NXC_LOS_THREADS / 64) * NXC_LOS_WAVE_BYTES + 32;
    K.tile_cap = NXC_LOS_TILE;
    while (K.tile_cap > 32 && lds_rest + (size_t)K.tile_cap * NXC_LOS_SP * sizeof(double) > 160 * 1024)
        K.tile_cap >>= 1;
    const size_t lds = lds_rest + (size_t)K.tile_cap * NXC_LOS_SP * sizeof(double);   // (+ trip counter, counter sums)
    if (lds > 160 * 1024) return fail(NXC_ERR_ARG, "g-value tables exceed the LDS");

    // device scratch: blob | sc | ladder | radiance | npackets | included | used
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return o; };
    const size_t o_blob = take(stage_bytes), o_sc = take((size_t)8 * S * 8),
                 o_lad = take((size_t)d->n_ladder * 8),
                 o_rad = take((size_t)S * 8), o_np = take((size_t)S * 8),
                 o_inc = take(included ? (size_t)n_index : 0),
                 o_used = take(used_pairs ? (size_t)used_cap * 16 : 0), o_nu = take(8);
    int rc = ensure(reinterpret_cast<void **>(&h->d_scratch), &h->scratch_cap, off);
    if (rc) return rc;
    unsigned char *base = reinterpret_cast<unsigned char *>(h->d_scratch);
    hipStream_t st = h->stream;
    HIPCHK(hipMemcpyAsync(base + o_blob, blob.data(), stage_bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(base + o_sc, sc, (size_t)8 * S * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(base + o_lad, d->ladder, (size_t)d->n_ladder * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(base + o_rad, 0, (size_t)S * 8, st));
    HIPCHK(hipMemsetAsync(base + o_np, 0, (size_t)S * 8, st));
    if (included) HIPCHK(hipMemsetAsync(base + o_inc, 0, (size_t)n_index, st));
    HIPCHK(hipMemsetAsync(base + o_nu, 0, 8, st));
    HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(DevCounters), st));
    if (P > 0) {
        if ((rc = prep_kernel(k_los<T, I>, lds))) return rc;
        const int tiles = (int)((S + K.tile_cap - 1) / K.tile_cap);
        constexpr int WPG = NXC_LOS_THREADS / 64;                  // waves per workgroup of k_los
        // the samples go through in slabs: the block scratch is sized for the worst case of one
        // slot per row (40 bytes; the bench cloud uses a fifth of a slot per row)
        int64_t slab = std::min<int64_t>(P, int64_t(1) << 24);
        if (const char *t = std::getenv("NXC_TEST_LOS_SLAB_ROWS"))       // tests: several slabs at small sizes
            slab = std::max<int64_t>(256, std::min<int64_t>(slab, std::atoll(t)));
        // (every region may add 7 empty slots to complete its last group)
        const int64_t max_regions = (slab + NXC_LOS_FORM - 1) / NXC_LOS_FORM;
        const size_t max_slots = (size_t)(slab + 8 * max_regions);
        const size_t o_desc = 0, o_sph = (max_slots * 8 + 255) & ~size_t(255),
                     o_n = o_sph + max_slots * 32;
        // the list of (row, spectrum) pairs near a cone: chunks of 64 (k_los -> k_los_pairs); when
        // it is full k_los decides the pairs itself
        constexpr int pair_chunks = 1 << 15;
        const size_t o_fill = o_n + 256, o_list = o_fill + (size_t)pair_chunks * 4;
        if ((rc = ensure(reinterpret_cast<void **>(&h->d_losblk), &h->losblk_cap,
                         o_list + (size_t)pair_chunks * 64 * 8)))
            return rc;
        size_t lds_pairs = ((stage_bytes + 31) & ~size_t(31)) + (size_t)((d->n_ladder + 3) & ~int64_t(3)) * 8;
        if (lds_pairs > 160 * 1024) return fail(NXC_ERR_ARG, "the ladder of ball centres exceeds the LDS");
        // the per-spectrum sums of a workgroup in LDS when they leave room for two workgroups per CU
        const int lds_sums = lds_pairs + (size_t)S * 16 <= 80 * 1024 ? 1 : 0;
        if (lds_sums) lds_pairs += (size_t)S * 16;
        if ((rc = prep_kernel(k_los_pairs<T, I>, lds_pairs))) return rc;
        unsigned *pair_used = reinterpret_cast<unsigned *>(h->d_losblk + o_n + 8);
        unsigned *pair_fill = reinterpret_cast<unsigned *>(h->d_losblk + o_fill);
        unsigned long long *pair_list = reinterpret_cast<unsigned long long *>(h->d_losblk + o_list);
        unsigned long long *bdesc = reinterpret_cast<unsigned long long *>(h->d_losblk + o_desc);
        double *bsph = reinterpret_cast<double *>(h->d_losblk + o_sph);
        unsigned long long *n_slots = reinterpret_cast<unsigned long long *>(h->d_losblk + o_n);
        if ((rc = begin_timed(h))) return rc;
        for (int64_t first = 0; first < P; first += slab) {
            const int64_t n = std::min<int64_t>(slab, P - first);
            const int64_t regions = (n + NXC_LOS_FORM - 1) / NXC_LOS_FORM;
            K.row_base = first;
            HIPCHK(hipMemsetAsync(n_slots, 0, 256 + (size_t)pair_chunks * 4, st));   // counters + chunk fills
            hipLaunchKernelGGL((k_los_blocks<T, I>),
                               dim3((unsigned)((regions + NXC_LOS_BLOCKS_THREADS / 64 - 1) /
                                               (NXC_LOS_BLOCKS_THREADS / 64))),
                               dim3(NXC_LOS_BLOCKS_THREADS), 0, st, n, K.cull, dx + first, dy + first, dz + first,
                               d_index ? d_index + first : d_index, bdesc, bsph, n_slots);
            HIPCHK(hipGetLastError());
            // persistent waves, one workgroup per CU (its LDS holds all the spectra of a tile)
            const int64_t groups = std::max<int64_t>(1, std::min<int64_t>(h->n_cu, (n / 8 / 64 + WPG) / WPG));
            hipLaunchKernelGGL((k_los<T, I>), dim3((unsigned)groups, tiles),
                               dim3(NXC_LOS_THREADS), lds, st, K, base + o_blob,
                               (int64_t)stage_bytes, S, reinterpret_cast<const double *>(base + o_sc),
                               n_slots, bdesc, bsph, pair_list, pair_fill, pair_used, pair_chunks,
                               dx + first, dy + first, dz + first, dvy + first, dfrac + first,
                               d_index ? d_index + first : d_index,
                               reinterpret_cast<const double *>(base + o_lad),
                               reinterpret_cast<double *>(base + o_rad),
                               reinterpret_cast<unsigned long long *>(base + o_np),
                               included ? base + o_inc : nullptr, (long long)used_cap,
                               used_pairs ? reinterpret_cast<long long *>(base + o_used) : nullptr,
                               reinterpret_cast<unsigned long long *>(base + o_nu), h->d_ctr);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL((k_los_pairs<T, I>), dim3((unsigned)(h->n_cu * 2)), dim3(NXC_BLOCK),
                               lds_pairs, st, K, base + o_blob, (int64_t)stage_bytes, S,
                               reinterpret_cast<const double *>(base + o_sc), pair_list, pair_fill,
                               pair_used, pair_chunks, lds_sums, dx + first, dy + first, dz + first,
                               dvy + first, dfrac + first, d_index ? d_index + first : d_index,
                               reinterpret_cast<const double *>(base + o_lad),
                               reinterpret_cast<double *>(base + o_rad),
                               reinterpret_cast<unsigned long long *>(base + o_np),
                               included ? base + o_inc : nullptr, (long long)used_cap,
                               used_pairs ? reinterpret_cast<long long *>(base + o_used) : nullptr,
                               reinterpret_cast<unsigned long long *>(base + o_nu), h->d_ctr);
            HIPCHK(hipGetLastError());
        }
        if ((rc = end_timed(h))) return rc;
    }
    HIPCHK(hipMemcpyAsync(radiance, base + o_rad, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(npackets, base + o_np, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    if (included) HIPCHK(hipMemcpyAsync(included, base + o_inc, (size_t)n_index, hipMemcpyDeviceToHost, st));
    unsigned long long nu = 0;
    HIPCHK(hipMemcpyAsync(&nu, base + o_nu, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (n_used) *n_used = (int64_t)nu;
    if (used_pairs) {
        const size_t got = (size_t)std::min<unsigned long long>(nu, (unsigned long long)used_cap);
        HIPCHK(hipMemcpy(used_pairs, base + o_used, got * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(used_pairs + used_cap, base + o_used + (size_t)used_cap * 8, got * 8,
                         hipMemcpyDeviceToHost));
    }
    return NXC_OK;
}

int los_check(nxc_handle *h, const nxc_los_desc *d, int64_t S, const double *sc, int64_t P,
              const double *radiance, const int64_t *npackets, const uint8_t *included,
              int64_t n_index, int64_t used_cap, const int64_t *used_pairs, const int64_t *n_used)
{
    if (!h || !d || S < 1 || P < 0 || !sc || !radiance || !npackets)
        return fail(NXC_ERR_ARG, "bad arguments");
    if (d->n_lines < 0 || d->n_lines > NXC_MAX_LINES || d->n_ladder < 1 || !d->ladder)
        return fail(NXC_ERR_ARG, "bad nxc_los_desc");
    if (included && n_index < 1) return fail(NXC_ERR_ARG, "included needs n_index");
    if (used_pairs && (used_cap < 1 || !n_used)) return fail(NXC_ERR_ARG, "used_pairs needs a capacity");
    HIPCHK(hipSetDevice(h->device));
    return NXC_OK;
}

// ... over samples in host memory: five columns (+ the index column) go to the device first
template <typename T>
int los_accumulate(nxc_handle *h, const nxc_los_desc *d, int64_t S, const double *sc, int64_t P,
                   const T *x, const T *y, const T *z, const T *vy, const T *frac,
                   const int64_t *index, int64_t n_index, double *radiance, int64_t *npackets,
                   uint8_t *included, int64_t used_cap, int64_t *used_pairs, int64_t *n_used)
{
    int rc = los_check(h, d, S, sc, P, radiance, npackets, included, n_index, used_cap, used_pairs,
                       n_used);
    if (rc) return rc;
    if (P && (!x || !y || !z || !vy || !frac)) return fail(NXC_ERR_ARG, "bad arguments");
    const size_t colP = ((size_t)P * sizeof(T) + 255) & ~size_t(255);
    const size_t idx_bytes = index ? (size_t)P * 8 : 0;
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_samples), &h->samples_cap, 5 * colP + idx_bytes)))
        return rc;
    const T *cols[5] = {x, y, z, vy, frac};
    for (int c = 0; c < 5 && P; c++)
        HIPCHK(hipMemcpyAsync(h->d_samples + c * colP, cols[c], (size_t)P * sizeof(T),
                              hipMemcpyHostToDevice, h->stream));
    if (index && P)
        HIPCHK(hipMemcpyAsync(h->d_samples + 5 * colP, index, idx_bytes, hipMemcpyHostToDevice, h->stream));
    auto col = [&](int c) { return reinterpret_cast<const T *>(h->d_samples + c * colP); };
    return los_run<T, long long>(h, d, S, sc, P, col(0), col(1), col(2), col(3), col(4),
                                 index ? reinterpret_cast<const long long *>(h->d_samples + 5 * colP)
                                       : (const long long *)nullptr,
                                 0, n_index, radiance, npackets, included, used_cap, used_pairs, n_used);
}

// Geometry of the tiled image: tile b = image rows ix = b (mod nb), nb a power of two; a tile's
// pixels (ix / nb, iz) must fit the LDS tile; the chunk shrinks as the tiles multiply
// (nb x cap = NXC_TILE_STAGE, at most 256).  False when the image has too many pixels for
// NXC_TILE_MAX tiles (above 1024^2): such images stay with k_image.
struct TilePlan {
    int nb_log2 = 0, tile_used = 0, cap = 256;
    size_t lds_bin = 0;
};
bool tile_plan(const nxc_handle *h, TilePlan *out)
{
    const int nx = h->header.G.nx, nz = h->header.G.nz;
    if (nx < 1 || nz < 1 || nz > h->tile_pixels) return false;
    const int rows = h->tile_pixels / nz;                    // image rows per tile
    int lg = 0;
    while (((nx + (1 << lg) - 1) >> lg) > rows) lg++;
    if ((1 << lg) > NXC_TILE_MAX) return false;
    out->nb_log2 = lg;
    out->tile_used = ((nx + (1 << lg) - 1) >> lg) * nz;
    out->cap = std::min(256, NXC_TILE_STAGE >> lg);
    out->lds_bin = ((h->img_bytes + 15) & ~size_t(15)) + (size_t)(1 << lg) * out->cap * 10 +
                   (2 * (size_t)(1 << lg) + 3) * 4;
    return out->lds_bin <= 160 * 1024;
}

// a-6..a-8 over samples on the device, through LDS-privatised tiles (nxc_kernels.hpp: k_image_bin,
// k_image_tiles).  The samples go through in slabs so that the chunk scratch stays bounded
// (10 bytes per sample of a slab at worst: every sample inside the image).
template <typename T, bool DEFER, int CAP>
int image_run_tiles(nxc_handle *h, const TilePlan &tp, int64_t p, const T *dx, const T *dy,
                    const T *dz, const T *dvy, const T *dfrac)
{
    int rc, per_cu = 0;
    // pass 2 forms the weights (DEFER): the image tables sit in front of its tile
    const size_t tables = DEFER ? (h->img_bytes + 15) & ~size_t(15) : 0;
    const size_t lds_tiles = tables + (size_t)NXC_TILE_PIXELS * 12;
    if ((rc = prep_kernel(k_image_bin<T, DEFER, CAP>, tp.lds_bin))) return rc;
    if ((rc = prep_kernel(k_image_tiles<DEFER, CAP>, lds_tiles))) return rc;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_image_bin<T, DEFER, CAP>,
                                                        NXC_TILE_BIN_BLOCK, tp.lds_bin));
    const int nb = 1 << tp.nb_log2;
    const int64_t per_trip = (int64_t)NXC_TILE_BIN_BLOCK * nxc_tile_unroll<T>();
    const int ng = std::max(1, h->n_cu / nb);             // consumer groups per tile
    int64_t slab_max = std::min<int64_t>(p, h->tile_slab), slab, prod, span, mc;
    size_t o_sl, o_list, o_n;
    for (;;) {
        // slabs of equal size: a short last one would run on a fraction of the chip
        const int64_t n_slabs = (p + slab_max - 1) / slab_max;
        slab = (p + n_slabs - 1) / n_slabs;
        prod = (int64_t)h->n_cu * (per_cu > 0 ? per_cu : 1);
        prod = std::max<int64_t>(1, std::min<int64_t>(prod, (slab + per_trip - 1) / per_trip));
        span = ((slab + prod - 1) / prod + per_trip - 1) / per_trip * per_trip;
        if (span > (int64_t(1) << 15) * CAP) {            // a producer's chunk numbers are 16 bits
            slab_max = (int64_t(1) << 15) * CAP * prod;
            continue;
        }
        mc = span / CAP + nb;
        // scratch: payloads | pixels-in-tile | chunk lists [producer][tile][mc] | their lengths
        const size_t entries = (size_t)prod * (size_t)mc * CAP;
        o_sl = entries * 8;
        o_list = o_sl + entries * 2;
        o_n = (o_list + (size_t)prod * nb * (size_t)mc * 2 + 255) & ~size_t(255);
        rc = ensure(reinterpret_cast<void **>(&h->d_tiles), &h->tiles_cap, o_n + (size_t)prod * nb * 4);
        if (rc == NXC_OK) break;
        if (slab <= (int64_t(1) << 22)) return rc;        // HBM is full: not even 50 MB
        (void)hipGetLastError();
        slab_max = slab >> 1;                             // a smaller slab needs less scratch
    }
    double *sw = reinterpret_cast<double *>(h->d_tiles);
    unsigned short *sl = reinterpret_cast<unsigned short *>(h->d_tiles + o_sl);
    unsigned short *list = reinterpret_cast<unsigned short *>(h->d_tiles + o_list);
    unsigned *nlist = reinterpret_cast<unsigned *>(h->d_tiles + o_n);
    if ((rc = begin_timed(h))) return rc;
    for (int64_t first = 0; first < p; first += slab) {
        const int64_t n = std::min<int64_t>(slab, p - first);
        const int64_t grid = (n + span - 1) / span;       // <= prod; the scratch regions keep their place
        hipLaunchKernelGGL((k_image_bin<T, DEFER, CAP>), dim3((unsigned)grid), dim3(NXC_TILE_BIN_BLOCK),
                           tp.lds_bin, h->stream, h->d_blob_img, (int64_t)h->img_bytes, n, span,
                           (int)mc, tp.nb_log2, dx + first, dy + first, dz + first, dvy + first,
                           dfrac + first, sw, sl, list, nlist, h->d_ctr);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_image_tiles<DEFER, CAP>), dim3((unsigned)(nb * ng)), dim3(NXC_IMAGE_BLOCK),
                           lds_tiles, h->stream, h->d_blob_img, (int64_t)h->img_bytes, (int)grid,
                           (int)mc, tp.nb_log2, ng, tp.tile_used, (int)h->header.G.nz, sw, sl, list,
                           nlist, h->d_image, h->d_ctr);
        HIPCHK(hipGetLastError());
    }
    if ((rc = end_timed(h))) return rc;
    HIPCHK(stream_sync(h));
    return NXC_OK;
}

// a-6..a-8 over samples on the device
// below this the two launches do not pay (measured: 2^16 samples 0.027 ms either way, 2^18 0.055
// against 0.092, 2^23 0.18 against 0.31, 2^26 0.64 against 1.57)
constexpr int64_t NXC_TILE_MIN_SAMPLES = int64_t(1) << 17;
template <typename T>
int image_run(nxc_handle *h, int64_t p, const T *dx, const T *dy, const T *dz, const T *dvy,
              const T *dfrac)
{
    int rc, per_cu = 0;
    TilePlan tp;
    const bool fits = tile_plan(h, &tp);
    if (h->image_mode == 2 && !fits)
        return fail(NXC_ERR_ARG, "this image does not fit the tiled path (too many pixels)");
    if (fits && (h->image_mode == 2 || (h->image_mode == 0 && p >= NXC_TILE_MIN_SAMPLES))) {
        // float32 samples (stored rows, the down-cast image) leave the weight to pass 2 when the
        // image tables fit a CU's LDS next to a tile
        const bool f32_values = sizeof(T) == 4 || h->header.G.downcast_f32 != 0;
        const bool room = ((h->img_bytes + 15) & ~size_t(15)) + (size_t)NXC_TILE_PIXELS * 12 <= 160 * 1024;
        const bool defer = f32_values && room;
#define NXC_TILES_CASE(C)                                                                       \
        case C:                                                                                 \
            return defer ? image_run_tiles<T, true, C>(h, tp, p, dx, dy, dz, dvy, dfrac)        \
                         : image_run_tiles<T, false, C>(h, tp, p, dx, dy, dz, dvy, dfrac);
        switch (tp.cap) {
            NXC_TILES_CASE(256)
            NXC_TILES_CASE(128)
            NXC_TILES_CASE(64)
        default:
            return fail(NXC_ERR_STATE, "tile plan with an unknown chunk size");
        }
#undef NXC_TILES_CASE
    }
    if ((rc = prep_kernel(k_image<T>, h->img_bytes))) return rc;
    // as many 1024-thread groups as fit a CU (two for Na's 45 KB of tables), each staging the
    // tables once and striding over the samples
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_image<T>, NXC_IMAGE_BLOCK,
                                                        h->img_bytes));
    int64_t grid = (int64_t)h->n_cu * (per_cu > 0 ? per_cu : 1);
    grid = std::max<int64_t>(1, std::min<int64_t>(grid, (p + NXC_IMAGE_BLOCK - 1) / NXC_IMAGE_BLOCK));
    if ((rc = begin_timed(h))) return rc;
    hipLaunchKernelGGL(k_image<T>, dim3((unsigned)grid), dim3(NXC_IMAGE_BLOCK), h->img_bytes,
                       h->stream, h->d_blob_img, (int64_t)h->img_bytes, p, dx, dy, dz, dvy, dfrac,
                       h->d_image, h->d_ctr);
    HIPCHK(hipGetLastError());
    if ((rc = end_timed(h))) return rc;
    HIPCHK(stream_sync(h));
    return NXC_OK;
}

// ... over samples in host memory, 64-bit or as save() keeps them (32-bit)
template <typename T>
int image_accumulate(nxc_handle *h, int64_t p, const T *x, const T *y, const T *z, const T *vy,
                     const T *frac)
{
    if (!h || !h->have_image) return fail(NXC_ERR_STATE, "nxc_set_image has not been called");
    HIPCHK(hipSetDevice(h->device));
    if (p < 0 || (p && (!x || !y || !z || !vy || !frac))) return fail(NXC_ERR_ARG, "bad arguments");
    HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(DevCounters), h->stream));
    if (p == 0) return NXC_OK;
    const size_t col = (size_t)p * sizeof(T);
    int rc = ensure(reinterpret_cast<void **>(&h->d_samples), &h->samples_cap, 5 * col);
    if (rc) return rc;
    T *d = reinterpret_cast<T *>(h->d_samples);
    const T *src[5] = {x, y, z, vy, frac};
    for (int c = 0; c < 5; c++)
        HIPCHK(hipMemcpyAsync(d + c * p, src[c], col, hipMemcpyHostToDevice, h->stream));
    return image_run<T>(h, p, d, d + p, d + 2 * p, d + 3 * p, d + 4 * p);
}

}  // namespace

// =============================================================================================
extern "C" {

int nxc_abi_version(void) { return NXC_ABI_VERSION; }

const char *nxc_last_error_string(void) { return g_error.c_str(); }

int nxc_device_count(int *count)
{
    if (!count) return fail(NXC_ERR_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(NXC_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return NXC_OK;
}

int nxc_create(int device, nxc_handle **out)
{
    return guarded([&]() -> int {
    if (!out) return fail(NXC_ERR_ARG, "out is null");
    *out = nullptr;
    int n = 0;
    int rc = nxc_device_count(&n);
    if (rc) return rc;
    if (n < 1) return fail(NXC_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(NXC_ERR_ARG, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    nxc_handle *h = new (std::nothrow) nxc_handle();
    if (!h) return fail(NXC_ERR_ARG, "out of host memory");
    h->device = device;
    h->n_cu = prop.multiProcessorCount;
    for (int i = 0; i < NXC_LOG_BINS; i++)          // the table nxc_log reads from the LDS header
        for (int c = 0; c < 3; c++) h->header.logtab[i][c] = NXC_LOG_TABLE_DATA[i][c];
    std::snprintf(h->name, sizeof h->name, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
                  prop.multiProcessorCount);
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&h->d_ctr), sizeof(DevCounters));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&h->d_reduce), 64);
    if (e == hipSuccess) e = hipMemset(h->d_ctr, 0, sizeof(DevCounters));
    if (e != hipSuccess) {
        nxc_destroy(h);
        return fail(NXC_ERR_HIP, std::string("nxc_create: ") + hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> g(g_handles_lock);
        g_handles.push_back(h);
    }
    *out = h;
    return NXC_OK;
    });
}

int nxc_destroy(nxc_handle *h)
{
    if (!h) return NXC_OK;
    {
        std::lock_guard<std::mutex> g(g_handles_lock);
        g_handles.erase(std::remove(g_handles.begin(), g_handles.end(), h), g_handles.end());
    }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)stream_sync(h);          // bounded when a collective is still in flight
    g_coll_failure.clear();
    if (h->comm && g_rccl.ok) g_rccl.CommDestroy(h->comm);
    void *ptrs[] = {h->d_blob, h->d_image, h->d_packets, h->d_ctr, h->d_scratch,
                    h->d_steps, h->d_reduce, h->d_order, h->d_bounce, h->d_moonpos, h->d_offsets,
                    h->d_source, h->d_queue, h->d_samples, h->d_tiles, h->d_hist, h->d_rec, h->d_piece_hist,
                    h->d_blob_img, h->d_reduce_n, h->d_losblk};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    pool_flush(h);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_pre_coll) (void)hipEventDestroy(h->ev_pre_coll);
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    for (hipEvent_t ev : h->ev_piece)
        if (ev) (void)hipEventDestroy(ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return NXC_OK;
}

int nxc_device_name(nxc_handle *h, char *buf, int buflen)
{
    if (!h || !buf || buflen < 1) return fail(NXC_ERR_ARG, "bad arguments");
    std::snprintf(buf, (size_t)buflen, "%s", h->name);
    return NXC_OK;
}

int nxc_device_bus_id(nxc_handle *h, char *buf, int buflen)
{
    if (!h || !buf || buflen < 16) return fail(NXC_ERR_ARG, "bad arguments");
    HIPCHK(hipDeviceGetPCIBusId(buf, buflen, h->device));
    return NXC_OK;
}

int nxc_mem_info(nxc_handle *h, uint64_t *free_bytes, uint64_t *total_bytes)
{
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    f += pool_bytes(h);          // blocks kept from freed row stores are reused or given back on demand
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return NXC_OK;
}

int nxc_synchronize(nxc_handle *h)
{
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(stream_sync(h));
    if (h->streamed_pending) {
        // the pipelined pass is the one launch whose kernel may give up by itself (a wave that has
        // waited three seconds for the next piece of its queue): say so instead of handing over a
        // partial image as if it were the result
        h->streamed_pending = false;
        DevCounters c;
        HIPCHK(hipMemcpy(&c, h->d_ctr, sizeof c, hipMemcpyDeviceToHost));
        if (c.unfinished != 0) {
            h->n_packets = 0;                  // the queue holds a partly ordered set: upload again
            char buf[400];
            std::snprintf(buf, sizeof buf,
                          "the pipelined pass gave up waiting for its queue (%llu packets not "
                          "integrated): the ordering kernels did not run beside the persistent "
                          "kernel (a profiler serialising kernels?); image and counters are partial "
                          "-- upload with nxc_packets_upload and run nxc_integrate_const_async",
                          (unsigned long long)c.unfinished);
            return fail(NXC_ERR_INCOMPLETE, buf);
        }
    }
    return NXC_OK;
}

int nxc_set_forces(nxc_handle *h, const nxc_forces *f)
{
    return guarded([&]() -> int {
    if (!h || !f) return fail(NXC_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    PackedLut lut;
    const double zero_one[2] = {0.0, 1.0}, zeros[2] = {0.0, 0.0};
    int rc;
    // (the launch groups of an Input.run set the same forces again and again: the packed table of
    // the last call is kept when its nodes are the same)
    const double *nv = f->radpres ? f->v_tab : zero_one, *na = f->radpres ? f->a_tab : zeros;
    const size_t nn = f->radpres ? (size_t)(f->n_tab > 0 ? f->n_tab : 0) : 2;
    if (f->radpres && (f->n_tab < 2 || !f->v_tab || !f->a_tab)) return fail(NXC_ERR_ARG, "radiation table missing");
    const bool same_table = h->have_forces && h->force_v.size() == nn && !h->force_lut.bytes.empty() &&
                            std::memcmp(h->force_v.data(), nv, nn * sizeof(double)) == 0 &&
                            std::memcmp(h->force_a.data(), na, nn * sizeof(double)) == 0;
    h->force_v.assign(nv, nv + nn);
    h->force_a.assign(na, na + nn);
    // 8 cells per node: a cell then rarely holds two nodes (0.5 % of the Na table's cells against
    // 2 % at 4), so the two-row probe of lut_interp hits and the divergent walk mostly stays out
    // of the step loop; 16 measured slower again in the image kernel (LDS footprint).
    h->force_cells_per_node = 8;
#ifdef NXC_EXPERIMENT_KNOBS
    if (const char *e = std::getenv("NXC_FORCE_CELLS")) h->force_cells_per_node = std::max(2, std::atoi(e));
#endif
    if (!same_table || h->force_lut_cells != h->force_cells_per_node) {
        rc = pack_lut(h->force_v.data(), h->force_a.data(), (int64_t)h->force_v.size(), lut,
                      "nxc_forces radiation table", h->force_cells_per_node);
        if (rc) { h->have_forces = false; return rc; }
        h->force_lut = std::move(lut);
        h->force_lut_cells = h->force_cells_per_node;
    }
    h->F.GM = f->GM;
    h->F.vrplanet = f->vrplanet;
    h->F.photo = f->photo;
    h->F.inv_lifetime = f->lifetime > 0 ? 1.0 / f->lifetime : 0.0;   // np.ones(n)/lifetime
    h->F.grav = f->gravity ? 1 : 0;
    h->F.rad = f->radpres ? 1 : 0;
    h->F.loss = f->lifetime > 0 ? LOSS_LIFETIME : (f->has_photo ? LOSS_PHOTO : LOSS_NONE);
    h->have_forces = true;
    return upload_blob(h);
    });
}

int nxc_set_image(nxc_handle *h, const nxc_image_desc *d)
{
    return guarded([&]() -> int {
    if (!h || !d) return fail(NXC_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    if (d->nx < 1 || d->nz < 1 || d->nx > 32768 || d->nz > 32768 || !d->xedges || !d->zedges)
        return fail(NXC_ERR_ARG, "image dims/edges invalid");
    if (d->quantity != 0 && d->quantity != 1) return fail(NXC_ERR_ARG, "quantity must be 0 or 1");
    const int nl = d->quantity == 1 ? d->n_lines : 0;
    if (nl < 0 || nl > NXC_MAX_LINES) return fail(NXC_ERR_ARG, "n_lines out of range");
    if (!(d->apix_cm2 > 0)) return fail(NXC_ERR_ARG, "apix_cm2 must be positive");
    if (!std::isfinite(d->vrplanet)) return fail(NXC_ERR_ARG, "vrplanet must be finite");
    for (int64_t k = 0; k < d->nx; k++)
        if (!(d->xedges[k + 1] > d->xedges[k])) return fail(NXC_ERR_ARG, "xedges not ascending");
    for (int64_t k = 0; k < d->nz; k++)
        if (!(d->zedges[k + 1] > d->zedges[k])) return fail(NXC_ERR_ARG, "zedges not ascending");

    std::vector<unsigned char> part;
    ImageK G{};
    std::memcpy(G.M, d->M, sizeof G.M);
    G.vrplanet = d->vrplanet;
    G.apix_cm2 = d->apix_cm2;
    G.quantity = d->quantity;
    G.n_lines = nl;
    G.downcast_f32 = d->downcast_f32 ? 1 : 0;
#ifdef NXC_EXPERIMENT_KNOBS
    if (const char *dbg = std::getenv("NXC_DEBUG_IMAGE")) G.dbg = std::atoi(dbg);
#endif
    G.nx = (int)d->nx;
    G.nz = (int)d->nz;
    G.x_is_x = (d->M[0] == 1.0 && d->M[1] == 0.0 && d->M[2] == 0.0 && d->M[3] == 0.0 &&
                d->M[6] == 0.0) ? 1 : 0;
    for (int l = 0; l < nl; l++) {
        PackedLut lut;
        int rc = pack_lut(d->line_v[l], d->line_g[l], d->line_n[l], lut, "g-value table");
        if (rc) return rc;
        h->line_local[l] = lut.desc;
        h->line_start[l] = part.size();
        part.insert(part.end(), lut.bytes.begin(), lut.bytes.end());
    }
    h->xedges_local = (int64_t)part.size();
    const unsigned char *xe = reinterpret_cast<const unsigned char *>(d->xedges);
    part.insert(part.end(), xe, xe + (d->nx + 1) * sizeof(double));
    h->zedges_local = (int64_t)part.size();
    const unsigned char *ze = reinterpret_cast<const unsigned char *>(d->zedges);
    part.insert(part.end(), ze, ze + (d->nz + 1) * sizeof(double));
    G.x_lo = d->xedges[0];
    G.x_inv_step = (double)d->nx / (d->xedges[d->nx] - d->xedges[0]);
    G.z_lo = d->zedges[0];
    G.z_inv_step = (double)d->nz / (d->zedges[d->nz] - d->zedges[0]);
    h->image_part = std::move(part);
    h->G = G;
    h->have_image = true;

    const size_t npix = (size_t)d->nx * (size_t)d->nz;
    if (npix != h->npix) {
        if (h->d_image) HIPCHK(hipFree(h->d_image));
        h->d_image = nullptr; h->npix = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_image), 2 * npix * sizeof(double)));
        h->npix = npix;
    }
    int rc = upload_blob(h);
    if (rc) return rc;
    return nxc_image_clear(h);
    });
}

int nxc_set_bounce(nxc_handle *h, const nxc_bounce_desc *d)
{
    return guarded([&]() -> int {
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!d) {                       // back to perfect sticking
        h->have_bounce = false;
        return NXC_OK;
    }
    if (d->nx < 8 || d->ny < 8 || !d->tx || !d->ty || !d->coef || !(d->unit_km > 0) ||
        d->accomfactor < 0 || d->accomfactor > 1 || d->stickcoef < 0 || d->stickcoef > 1)
        return fail(NXC_ERR_ARG, "bad nxc_bounce_desc");
    const size_t ncoef = (size_t)(d->nx - 4) * (size_t)(d->ny - 4);
    const size_t total = (size_t)d->nx + (size_t)d->ny + ncoef;
    int rc = ensure(reinterpret_cast<void **>(&h->d_bounce), &h->bounce_cap, total * sizeof(double));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(h->d_bounce, d->tx, (size_t)d->nx * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_bounce + d->nx, d->ty, (size_t)d->ny * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_bounce + d->nx + d->ny, d->coef, ncoef * 8, hipMemcpyHostToDevice, h->stream));
    BounceK &B = h->header.B;
    B.GM = d->GM; B.unit_km = d->unit_km; B.accom = d->accomfactor; B.stickcoef = d->stickcoef;
    B.A0 = d->A[0]; B.A1 = d->A[1]; B.A2 = d->A[2]; B.t0 = d->t0; B.t1 = d->t1;
    B.temp_dependent = d->temp_dependent ? 1 : 0; B.nx = (int)d->nx; B.ny = (int)d->ny;
    B.seed = d->seed;
    B.tx = h->d_bounce; B.ty = h->d_bounce + d->nx; B.coef = h->d_bounce + d->nx + d->ny;
    h->have_bounce = true;
    if (h->d_blob)
        HIPCHK(hipMemcpyAsync(h->d_blob + offsetof(LdsHeader, B), &h->header.B, sizeof(BounceK),
                              hipMemcpyHostToDevice, h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

int nxc_set_bodies(nxc_handle *h, const nxc_bodies_desc *d)
{
    return guarded([&]() -> int {
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    BodyK &K = h->header.Bd;
    K = BodyK{};
    h->have_bodies = false;
    if (d && (d->n_moons != 0 || d->chx_on)) {
        if (d->n_moons < 0 || d->n_moons > NXC_MAX_MOONS)
            return fail(NXC_ERR_ARG, "nxc_bodies_desc: n_moons must be 0..4");
        for (int m = 0; m < d->n_moons; m++) {
            if (!(d->a[m] > 0) || !(d->radius[m] >= 0) || !std::isfinite(d->gm[m]) ||
                !std::isfinite(d->omega[m]) || !std::isfinite(d->phi[m]))
                return fail(NXC_ERR_ARG, "nxc_bodies_desc: bad moon parameters");
        }
        if (d->chx_on && (!(d->chx_width > 0) || !(d->chx_height > 0) || !(d->chx_k0 >= 0) ||
                          (d->chx_omega != 0 && !(d->chx_omega * d->chx_rho0 > 0))))
            return fail(NXC_ERR_ARG, "nxc_bodies_desc: bad torus parameters");
        if (!std::isfinite(d->t0)) return fail(NXC_ERR_ARG, "nxc_bodies_desc: bad t0");
        h->bodies = *d;
        K.n_moons = d->n_moons;
        for (int m = 0; m < d->n_moons; m++) {
            K.gm[m] = d->gm[m];
            K.rad2[m] = d->radius[m] * d->radius[m];
            K.a[m] = d->a[m];
        }
        K.chx_on = d->chx_on ? 1 : 0;
        if (K.chx_on) {
            K.chx_k0 = d->chx_k0; K.chx_rho0 = d->chx_rho0;
            K.chx_inv_w = 1.0 / d->chx_width; K.chx_inv_h = 1.0 / d->chx_height;
            K.chx_vel = d->chx_omega != 0 ? 1 : 0;
            K.chx_omega = d->chx_omega;
            K.chx_inv_v0 = K.chx_vel ? 1.0 / (d->chx_omega * d->chx_rho0) : 0.0;
        }
        h->have_bodies = true;
    }
    if (h->d_blob)
        HIPCHK(hipMemcpyAsync(h->d_blob + offsetof(LdsHeader, Bd), &h->header.Bd, sizeof(BodyK),
                              hipMemcpyHostToDevice, h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

int nxc_set_first_index(nxc_handle *h, int64_t first_index)
{
    if (!h || first_index < 0) return fail(NXC_ERR_ARG, "bad arguments");
    h->first_id = first_index;
    return NXC_OK;
}

int nxc_image_clear(nxc_handle *h)
{
    if (!h || !h->have_image) return fail(NXC_ERR_STATE, "nxc_set_image has not been called");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemsetAsync(h->d_image, 0, 2 * h->npix * sizeof(double), h->stream));
    return NXC_OK;
}

int nxc_image_mode(nxc_handle *h, int mode, int tile_pixels, int64_t slab_samples)
{
    return guarded([&]() -> int {
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    if (mode < 0 || mode > 2) return fail(NXC_ERR_ARG, "mode is 0 (by size), 1 (atomics) or 2 (tiles)");
    if (tile_pixels < 0 || tile_pixels > NXC_TILE_PIXELS)
        return fail(NXC_ERR_ARG, "tile_pixels is 0 (default) or at most 8192");
    if (slab_samples < 0 || slab_samples > (int64_t(1) << 31))
        return fail(NXC_ERR_ARG, "slab_samples is 0 (default) or at most 2^31");
    h->image_mode = mode;
    h->tile_pixels = tile_pixels ? tile_pixels : NXC_TILE_PIXELS;
    h->tile_slab = slab_samples ? slab_samples : int64_t(1) << 28;
    return NXC_OK;
    });
}

int nxc_image_download(nxc_handle *h, double *image, uint64_t *counts)
{
    if (!h || !h->have_image) return fail(NXC_ERR_STATE, "nxc_set_image has not been called");
    HIPCHK(hipSetDevice(h->device));
    return guarded([&]() -> int {
    // the device keeps {weight sum, count} interleaved in fp64 (see image_add_pairs); split here
    std::vector<double> both(2 * h->npix);
    HIPCHK(hipMemcpyAsync(both.data(), h->d_image, 2 * h->npix * sizeof(double),
                          hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    for (size_t q = 0; q < h->npix; q++) {
        if (image) image[q] = both[2 * q];
        if (counts) counts[q] = (uint64_t)both[2 * q + 1];      // integer-valued, < 2^53
    }
    return NXC_OK;
    });
}

#ifdef NXC_STAMPS
extern "C" int nxc_debug_stamps(nxc_handle *h, unsigned long long out[8])
{
    DevCounters c;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(&c, h->d_ctr, sizeof c, hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; k++) out[k] = c.stamp[k];
    return NXC_OK;
}
#endif

int nxc_counters_get(nxc_handle *h, nxc_counters *out)
{
    if (!h || !out) return fail(NXC_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->device));
    DevCounters c;
    HIPCHK(hipMemcpyAsync(&c, h->d_ctr, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    out->particle_steps = c.particle_steps;
    out->samples = c.samples;
    out->samples_binned = c.samples_binned;
    out->nonfinite = c.nonfinite;
    out->bad_step = c.bad_step;
    out->neg_frac = c.neg_frac;
    out->unfinished = c.unfinished;
    out->wave_trips = c.wave_trips;
    return NXC_OK;
}

int nxc_last_kernel_ms(nxc_handle *h, float *ms)
{
    if (!h || !ms) return fail(NXC_ERR_ARG, "null argument");
    if (!h->timed) return fail(NXC_ERR_STATE, "no timed launch yet");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return NXC_OK;
}

int nxc_state(nxc_handle *h, int64_t n, const double *x, const double *y, const double *z,
              const double *vy, double *ax, double *ay, double *az, double *ioniz)
{
    return guarded([&]() -> int {
    int rc = need_forces(h);
    if (rc) return rc;
    if (n < 0 || (n && (!x || !y || !z || !vy || !ax || !ay || !az || !ioniz)))
        return fail(NXC_ERR_ARG, "bad arguments");
    if (h->have_bodies) return fail(NXC_ERR_STATE, "nxc_state: not available with moons set");
    if (n == 0) return NXC_OK;
    const size_t bytes = (size_t)n * sizeof(double);
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_scratch), &h->scratch_cap, 8 * bytes)))
        return rc;
    double *d = h->d_scratch;
    const double *src[4] = {x, y, z, vy};
    for (int c = 0; c < 4; c++)
        HIPCHK(hipMemcpyAsync(d + c * n, src[c], bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = prep_kernel(k_state, h->force_bytes))) return rc;
    hipLaunchKernelGGL(k_state, dim3(flat_grid(h, n, NXC_BLOCK)), dim3(NXC_BLOCK), h->force_bytes,
                       h->stream, h->F, h->d_blob, (int64_t)h->force_bytes, n, d, d + n, d + 2 * n,
                       d + 3 * n, d + 4 * n, d + 5 * n, d + 6 * n, d + 7 * n);
    HIPCHK(hipGetLastError());
    double *dst[4] = {ax, ay, az, ioniz};
    for (int c = 0; c < 4; c++)
        HIPCHK(hipMemcpyAsync(dst[c], d + (4 + c) * n, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

int nxc_rk5_step(nxc_handle *h, int64_t n, const double *soa_in, const double *hstep,
                 double *soa_out, double *delta_out)
{
    return guarded([&]() -> int {
    int rc = need_forces(h);
    if (rc) return rc;
    if (n < 0 || (n && (!soa_in || !hstep || !soa_out))) return fail(NXC_ERR_ARG, "bad arguments");
    if (h->have_bodies) return fail(NXC_ERR_STATE, "nxc_rk5_step: not available with moons set");
    if (n == 0) return NXC_OK;
    const size_t col = (size_t)n * sizeof(double);
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_scratch), &h->scratch_cap, 25 * col)))
        return rc;
    double *d_in = h->d_scratch, *d_h = d_in + 8 * n, *d_out = d_h + n, *d_delta = d_out + 8 * n;
    HIPCHK(hipMemcpyAsync(d_in, soa_in, 8 * col, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(d_h, hstep, col, hipMemcpyHostToDevice, h->stream));
    const int grid = flat_grid(h, n, NXC_BLOCK);
    if (delta_out) {
        if ((rc = prep_kernel(k_rk5_step<true>, h->force_bytes))) return rc;
        hipLaunchKernelGGL(k_rk5_step<true>, dim3(grid), dim3(NXC_BLOCK), h->force_bytes,
                           h->stream, h->F, h->d_blob, (int64_t)h->force_bytes, n, d_in, d_h,
                           d_out, d_delta);
    } else {
        if ((rc = prep_kernel(k_rk5_step<false>, h->force_bytes))) return rc;
        hipLaunchKernelGGL(k_rk5_step<false>, dim3(grid), dim3(NXC_BLOCK), h->force_bytes,
                           h->stream, h->F, h->d_blob, (int64_t)h->force_bytes, n, d_in, d_h,
                           d_out, d_delta);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(soa_out, d_out, 8 * col, hipMemcpyDeviceToHost, h->stream));
    if (delta_out)
        HIPCHK(hipMemcpyAsync(delta_out, d_delta, 8 * col, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

// Counting sort of `n` packets (columns `stride` apart, starting at `soa`) by decreasing |v|^2, or
// by decreasing lifetime (d_lifetimes), entirely on `st`: histogram -> device scan -> scatter of
// the packets themselves into the queue out_queue[n][8] (one 64-byte record per packet) with
// out_order[n] = base + local packet index.
// scale > 0: the bin scale; scale <= 0: taken on the device from *d_max (k_speed_max ran before).
// beside_persistent: the launches may have to run next to a persistent kernel that fills every
// CU's LDS, so the histogram uses global atomics instead of LDS bins.
static int order_async(nxc_handle *h, hipStream_t st, const double *soa, int64_t stride, int64_t n,
                       const long long *d_lifetimes, double scale, unsigned long long *d_max,
                       unsigned long long *d_hist, unsigned *out_order, double *out_queue,
                       bool beside_persistent, unsigned base = 0, bool flight_key = false)
{
    const size_t hb = (size_t)NXC_ORDER_BINS * sizeof(unsigned long long);
    const int grid = flat_grid(h, n, NXC_BLOCK);
    const unsigned long long *mx = scale > 0 ? (const unsigned long long *)nullptr : d_max;
    HIPCHK(hipMemsetAsync(d_hist, 0, hb, st));
#define NXC_ORDER_LAUNCH(KERNEL, ...)                                                           \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(NXC_BLOCK), 0, st, soa, d_lifetimes, stride, n,  \
                       scale, mx, __VA_ARGS__)
    if (d_lifetimes) NXC_ORDER_LAUNCH((k_order_hist<1, true>), d_hist);
    else if (flight_key) NXC_ORDER_LAUNCH((k_order_hist<2, true>), d_hist);
    else if (beside_persistent) NXC_ORDER_LAUNCH((k_order_hist<0, false>), d_hist);
    else NXC_ORDER_LAUNCH((k_order_hist<0, true>), d_hist);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(64), 0, st, d_hist);
    HIPCHK(hipGetLastError());
    if (d_lifetimes) NXC_ORDER_LAUNCH(k_order_scatter<1>, d_hist, out_order, base, out_queue);
    else if (flight_key) NXC_ORDER_LAUNCH(k_order_scatter<2>, d_hist, out_order, base, out_queue);
    else NXC_ORDER_LAUNCH(k_order_scatter<0>, d_hist, out_order, base, out_queue);
#undef NXC_ORDER_LAUNCH
    HIPCHK(hipGetLastError());
    return NXC_OK;
}

// Queue order of the resident packets.  k2max: upper bound of |v|^2 when the caller knows it (the
// sampler does), negative = find the largest finite |v|^2 on the device.  d_lifetimes != null:
// sort by those step counts instead (exact lifetimes from a counting pass; max_steps = their
// upper bound).
static int order_on_device(nxc_handle *h, double k2max, const long long *d_lifetimes,
                           int64_t max_steps, bool flight_key)
{
    const int64_t n = h->n_packets;
    const bool by_steps = d_lifetimes != nullptr;
    h->order_key = by_steps ? 1 : flight_key ? 2 : 0;
    if (by_steps) {
        if (n < 2 || n >= (int64_t)0xffffffffll || max_steps < 1) return NXC_OK;   // keep what there is
    } else {
        h->have_order = false;
        if (n < 2 || n >= (int64_t)0xffffffffll || k2max == 0 || std::isnan(k2max) ||
            (k2max > 0 && !std::isfinite(k2max)))
            return NXC_OK;
    }
    int rc = ensure(reinterpret_cast<void **>(&h->d_order), &h->order_cap, (size_t)n * sizeof(unsigned));
    if (rc) return rc;
    const size_t hb = (size_t)NXC_ORDER_BINS * sizeof(unsigned long long);
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_hist), &h->hist_cap, hb + 64))) return rc;
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_queue), &h->queue_cap,
                     (size_t)8 * n * sizeof(double))))
        return rc;
    unsigned long long *d_max = h->d_hist + NXC_ORDER_BINS;
    double scale = by_steps ? (double)(NXC_ORDER_BINS - 1) / (double)max_steps
                            : (k2max > 0 ? (double)(NXC_ORDER_BINS - 1) / k2max : 0.0);
    if (!by_steps && !(k2max > 0)) {
        HIPCHK(hipMemsetAsync(d_max, 0, sizeof(unsigned long long), h->stream));
        if (flight_key)
            hipLaunchKernelGGL(k_speed_max<true>, dim3(flat_grid(h, n, NXC_BLOCK)), dim3(NXC_BLOCK), 0,
                               h->stream, (const double *)h->d_packets, n, n, d_max);
        else
            hipLaunchKernelGGL(k_speed_max<false>, dim3(flat_grid(h, n, NXC_BLOCK)), dim3(NXC_BLOCK), 0,
                               h->stream, (const double *)h->d_packets, n, n, d_max);
        HIPCHK(hipGetLastError());
    }
    h->have_order = false;
    if ((rc = order_async(h, h->stream, h->d_packets, n, n, d_lifetimes, scale, d_max, h->d_hist,
                          h->d_order, h->d_queue, false, 0, flight_key)))
        return rc;
    HIPCHK(stream_sync(h));
    h->have_order = true;
    return NXC_OK;
}

int nxc_packets_upload(nxc_handle *h, int64_t n, const double *soa0)
{
    return guarded([&]() -> int {
    if (!h || n < 0 || (n && !soa0)) return fail(NXC_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const size_t bytes = (size_t)8 * n * sizeof(double);
    int rc = ensure(reinterpret_cast<void **>(&h->d_packets), &h->packets_cap, bytes);
    if (rc) return rc;
    if (n) HIPCHK(hipMemcpyAsync(h->d_packets, soa0, bytes, hipMemcpyHostToDevice, h->stream));
    h->n_packets = n;
    h->first_id = 0;
    h->rows_total = -1;
    // Queue order for the persistent kernels (longest-lived first): counting sort of the packet
    // indices by decreasing |v|^2 on the device, bounded by the largest launch speed found there.
    if ((rc = order_on_device(h, -1.0, nullptr, 0))) return rc;
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}
int nxc_packets_upload_pieces(nxc_handle *h, int32_t n_pieces, const int64_t *counts,
                              const double *const *soa)
{
    return guarded([&]() -> int {
    if (!h || n_pieces < 1 || !counts || !soa) return fail(NXC_ERR_ARG, "bad arguments");
    int64_t n = 0;
    for (int p = 0; p < n_pieces; p++) {
        if (counts[p] < 0 || (counts[p] && !soa[p])) return fail(NXC_ERR_ARG, "bad piece");
        n += counts[p];
    }
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure(reinterpret_cast<void **>(&h->d_packets), &h->packets_cap, (size_t)8 * n * sizeof(double));
    if (rc) return rc;
    // piece p's column c goes behind the same column of the pieces before it: the resident set is
    // the concatenation, without the host ever building it
    int64_t at = 0;
    for (int p = 0; p < n_pieces; p++) {
        for (int c = 0; c < 8 && counts[p]; c++)
            HIPCHK(hipMemcpyAsync(h->d_packets + c * n + at, soa[p] + c * counts[p],
                                  (size_t)counts[p] * sizeof(double), hipMemcpyHostToDevice, h->stream));
        at += counts[p];
    }
    h->n_packets = n;
    h->first_id = 0;
    h->rows_total = -1;
    if ((rc = order_on_device(h, -1.0, nullptr, 0))) return rc;
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

int nxc_packets_sample(nxc_handle *h, const nxc_source_desc *d, int64_t n, double *soa_out)
{
    return guarded([&]() -> int {
    if (!h || !d || n < 1) return fail(NXC_ERR_ARG, "bad arguments");
    if (d->speed_type < 0 || d->speed_type > 2 || d->angular_type < 0 || d->angular_type > 1 ||
        d->spatial_type < 0 || d->spatial_type > 1 || !(d->unit_km > 0) || !(d->exobase > 0))
        return fail(NXC_ERR_ARG, "bad nxc_source_desc");
    const bool tab_speed = d->speed_type == 2, spot = d->spatial_type == 1;
    const bool pcg = d->generator == 1;
    if (d->generator != 0 && d->generator != 1) return fail(NXC_ERR_ARG, "nxc_source_desc: generator must be 0 or 1");
    if (pcg) {
        if (d->spatial_type != 0 || d->speed_type != 0)
            return fail(NXC_ERR_ARG, "generator 1 (PCG64) covers the sources whose every draw is a "
                                     "random(npackets) vector: uniform surface, flat speeds");
        if (d->pcg_n < 1 || d->pcg_row0 < 0 || d->pcg_row0 + n > d->pcg_n ||
            d->pcg_n >= ((int64_t)1 << (NXC_PCG_BITS - 1)) || !(d->pcg_inc[1] & 1ull))
            return fail(NXC_ERR_ARG, "nxc_source_desc: PCG64 window outside its draw vectors");
    }
    if (tab_speed) {
        if (d->n_speed < 2 || d->n_speed > (1 << 24) || !d->speed_cdf || !d->speed_v)
            return fail(NXC_ERR_ARG, "nxc_source_desc: tabulated speeds need n_speed >= 2 and both tables");
        for (int64_t k = 0; k + 1 < d->n_speed; k++)
            if (!(d->speed_cdf[k + 1] >= d->speed_cdf[k]))
                return fail(NXC_ERR_ARG, "nxc_source_desc: speed_cdf must be non-decreasing");
        if (!(d->speed_cdf[d->n_speed - 1] > d->speed_cdf[0]))
            return fail(NXC_ERR_ARG, "nxc_source_desc: speed_cdf is flat");
    }
    double map_max = 0.0, map_sum = 0.0;
    if (spot) {
        if (d->map_nlon < 2 || d->map_nlat < 2 || d->map_nlon > 8192 || d->map_nlat > 8192 || !d->map)
            return fail(NXC_ERR_ARG, "nxc_source_desc: surface spot needs a density map");
        for (int64_t k = 0; k < d->map_nlon * d->map_nlat; k++) {
            if (!(d->map[k] >= 0.0) || !std::isfinite(d->map[k]))
                return fail(NXC_ERR_ARG, "nxc_source_desc: density map values must be finite and >= 0");
            map_max = std::max(map_max, d->map[k]);
            map_sum += d->map[k];
        }
        if (!(map_max > 0.0)) return fail(NXC_ERR_ARG, "nxc_source_desc: density map is all zero");
    }
    HIPCHK(hipSetDevice(h->device));
    const int64_t total = d->dest_total > 0 ? d->dest_total : n;
    const int64_t offset = d->dest_total > 0 ? d->dest_offset : 0;
    if (offset < 0 || offset + n > total) return fail(NXC_ERR_ARG, "nxc_source_desc: piece outside its set");
    const size_t bytes = (size_t)8 * total * sizeof(double);
    if (offset > 0 && (h->packets_cap < bytes || h->n_packets != total))
        return fail(NXC_ERR_STATE, "nxc_packets_sample: pieces of a set must start with dest_offset 0");
    int rc = ensure(reinterpret_cast<void **>(&h->d_packets), &h->packets_cap, bytes);
    if (rc) return rc;
    const size_t n_sp = tab_speed ? (size_t)d->n_speed : 0;
    const size_t n_map = spot ? (size_t)(d->map_nlon * d->map_nlat) : 0;
    const size_t n_pcg = pcg ? (size_t)4 * (NXC_PCG_BITS + NXC_PCG_VECS) : 0;   // doubles' worth
    if (n_sp + n_map + n_pcg) {
        if ((rc = ensure(reinterpret_cast<void **>(&h->d_source), &h->source_cap,
                         (2 * n_sp + n_map + n_pcg) * sizeof(double))))
            return rc;
        if (pcg) {
            const u128 inc = ((u128)d->pcg_inc[0] << 64) | d->pcg_inc[1];
            const std::vector<u128> maps = pcg_tables(inc, d->pcg_n);
            HIPCHK(hipMemcpyAsync(h->d_source, maps.data(), maps.size() * sizeof(u128),
                                  hipMemcpyHostToDevice, h->stream));
            HIPCHK(stream_sync(h));      // the table is a local
        }
        if (n_sp) {
            HIPCHK(hipMemcpyAsync(h->d_source, d->speed_cdf, n_sp * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_source + n_sp, d->speed_v, n_sp * 8, hipMemcpyHostToDevice, h->stream));
        }
        if (n_map)
            HIPCHK(hipMemcpyAsync(h->d_source + 2 * n_sp, d->map, n_map * 8, hipMemcpyHostToDevice, h->stream));
    }
    SourceK K{};
    K.endtime = d->endtime; K.exobase = d->exobase; K.sinlat0 = d->sinlat0; K.sinlat1 = d->sinlat1;
    K.lon0 = d->lon0; K.lon1 = d->lon1; K.vprob = d->vprob; K.vwidth = d->vwidth;
    K.unit_km = d->unit_km; K.sinalt0 = d->sinalt0; K.sinalt1 = d->sinalt1; K.az0 = d->az0;
    K.az1 = d->az1; K.random_time = d->random_time; K.speed_type = d->speed_type;
    K.angular_type = d->angular_type; K.is_planet = d->is_planet; K.seed = d->seed;
    K.first_index = d->first_index;
    K.spatial_type = d->spatial_type; K.n_speed = (int)n_sp;
    K.map_nlon = spot ? (int)d->map_nlon : 0; K.map_nlat = spot ? (int)d->map_nlat : 0;
    K.map_max = map_max;
    K.max_trials = NXC_SPOT_MIN_TRIALS;
    if (spot && map_sum > 0) {
        // acceptance rate of the uniform (lon, lat) proposal = mean / max of the map: a narrow spot
        // (sigma 0.05 rad: 8e-4) needs tens of thousands of trials for the unluckiest of 1e6 packets
        const double accept = map_sum / (double)(d->map_nlon * d->map_nlat) / map_max;
        const double want = 32.0 / accept;
        K.max_trials = want > (double)NXC_SPOT_MAX_TRIALS ? NXC_SPOT_MAX_TRIALS
                       : (want < (double)NXC_SPOT_MIN_TRIALS ? NXC_SPOT_MIN_TRIALS : (int)want);
    }
    K.speed_cdf = h->d_source; K.speed_v = h->d_source + n_sp; K.map = h->d_source + 2 * n_sp;
    K.generator = d->generator;
    if (pcg) {       // (pcg excludes the tabulated sources, so the maps sit at the buffer's start)
        K.pcg.state = ((u128)d->pcg_state[0] << 64) | d->pcg_state[1];
        K.pcg.row0 = d->pcg_row0;
        K.pcg.maps = reinterpret_cast<const nxc_u128 *>(h->d_source);
    }
    K.stride = total;
    K.offset = offset;
    HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(DevCounters), h->stream));
    if ((rc = begin_timed(h))) return rc;
    hipLaunchKernelGGL(k_sample, dim3(flat_grid(h, n, NXC_BLOCK)), dim3(NXC_BLOCK), 0, h->stream, K,
                       n, h->d_packets, h->d_ctr);
    HIPCHK(hipGetLastError());
    if ((rc = end_timed(h))) return rc;
    DevCounters c;
    HIPCHK(hipMemcpyAsync(&c, h->d_ctr, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    if (c.unfinished) {
        h->n_packets = 0;            // never-accepted candidates must not pass for packets
        h->have_order = false;
        return fail(NXC_ERR_ARG, "nxc_packets_sample: " + std::to_string(c.unfinished) +
                                 " packets found no launch point in the density map (is it "
                                 "almost everywhere zero?)");
    }
    h->n_packets = total;
    h->rows_total = -1;
    if (offset == 0) h->first_id = d->first_index;
    h->have_order = false;
    if (soa_out)        // this piece's eight columns
        HIPCHK(hipMemcpy2DAsync(soa_out, (size_t)n * 8, h->d_packets + offset, (size_t)total * 8,
                                (size_t)n * 8, 8, hipMemcpyDeviceToHost, h->stream));
    if (offset + n < total) {        // more pieces to come: the queue order waits for the last
        HIPCHK(stream_sync(h));
        return NXC_OK;
    }
    double vmax;
    if (tab_speed) {
        vmax = 0.0;
        for (size_t k = 0; k < n_sp; k++) vmax = std::max(vmax, std::fabs(d->speed_v[k]));
        vmax /= d->unit_km;
    } else {
        vmax = (d->speed_type == 0 ? std::fabs(d->vprob) + std::fabs(d->vwidth)
                                   : std::fabs(d->vprob) + 6 * std::fabs(d->vwidth)) / d->unit_km;
    }
    // (a set made of pieces may mix sources: let the device find its largest launch speed)
    if ((rc = order_on_device(h, d->dest_total > 0 ? -1.0 : vmax * vmax, nullptr, 0))) return rc;
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

int nxc_integrate_const_async(nxc_handle *h, double step, int64_t n_iter, double outeredge,
                              uint32_t flags)
{
    return guarded([&]() -> int {
    int rc = need_forces(h);
    if (rc) return rc;
    if (h->n_packets < 1) return fail(NXC_ERR_STATE, "no resident packets (nxc_packets_upload)");
    if (!(step > 0) || n_iter < 0) return fail(NXC_ERR_ARG, "step must be > 0, n_iter >= 0");
    const bool image = (flags & NXC_RUN_IMAGE) != 0;
    if (image && !h->have_image) return fail(NXC_ERR_STATE, "NXC_RUN_IMAGE without nxc_set_image");
    if ((rc = speed_order_for_const(h))) return rc;
    return launch_const(h, step, n_iter, outeredge, image, nullptr, nullptr);
    });
}

// Upload and integrate in one pipelined pass (the integrate call including the
// host-to-device copy of X0).  ONE launch of the persistent integrator starts at once and consumes
// a queue that is still being filled: the packets cross PCIe in pieces (copy stream); each piece is
// put into queue order by small kernels on a second stream (global-atomic histogram, one-wave
// scan: no host round trip and no LDS, since they run next to the persistent kernel whose block
// owns the CU's LDS) and then published (k_publish / wait_published in nxc_kernels.hpp).  The
// kernel's waves only ever wait for the first piece: a piece uploads in a fraction of the time it
// takes to integrate.  (Cutting the PASS into one launch per piece does not pipeline: a 12-wave
// workgroup keeps its CU until its last wave ends, so no CU is free for the next launch before
// the previous one is over -- measured: 62 ms against 56 for upload-then-integrate.)
int nxc_integrate_const_streamed(nxc_handle *h, int64_t n, const double *soa0, int32_t pieces,
                                 double step, int64_t n_iter, double outeredge, uint32_t flags)
{
    return guarded([&]() -> int {
    int rc = need_forces(h);
    if (rc) return rc;
    if (n < 1 || !soa0 || pieces < 1 || pieces > 32) return fail(NXC_ERR_ARG, "bad arguments (1..32 pieces)");
    if (!(step > 0) || n_iter < 0) return fail(NXC_ERR_ARG, "step must be > 0, n_iter >= 0");
    const bool image = (flags & NXC_RUN_IMAGE) != 0;
    if (image && !h->have_image) return fail(NXC_ERR_STATE, "NXC_RUN_IMAGE without nxc_set_image");
    if (n >= (int64_t)0xffffffffll) return fail(NXC_ERR_ARG, "too many packets for one resident set");
    if (h->have_bodies || h->have_bounce)
        return fail(NXC_ERR_STATE, "the streamed pass covers the plain force models; with moons or "
                                   "surface re-emission upload first (nxc_packets_upload)");
    if ((int64_t)pieces > n) pieces = (int32_t)n;
    const size_t per_piece = NXC_ORDER_BINS + 8;                    // bins, then the largest |v|^2
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_packets), &h->packets_cap, (size_t)8 * n * 8))) return rc;
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_queue), &h->queue_cap, (size_t)8 * n * 8))) return rc;
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_order), &h->order_cap, (size_t)n * sizeof(unsigned)))) return rc;
    if (!h->d_piece_hist)
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&h->d_piece_hist),
                         (32 * per_piece + 8) * sizeof(unsigned long long)));
    for (int e = 0; e <= 32; e++)
        if (!h->ev_piece[e]) HIPCHK(hipEventCreateWithFlags(&h->ev_piece[e], hipEventDisableTiming));
    unsigned long long *d_avail = h->d_piece_hist + 32 * per_piece;
    h->n_packets = n;
    h->first_id = 0;
    h->rows_total = -1;
    h->have_order = false;

    const size_t tables = image ? h->all_bytes : h->force_bytes;
    const size_t lds = persist_lds(tables);
    if ((rc = upload_step(h, step))) return rc;
    HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(DevCounters), h->stream));
    HIPCHK(hipMemsetAsync(d_avail, 0, sizeof(unsigned long long), h->stream));
    // everything queued on the handle's stream so far (image clear, tables, the zeroed word) first
    HIPCHK(hipEventRecord(h->ev_piece[32], h->stream));
    HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_piece[32], 0));
    HIPCHK(hipStreamWaitEvent(h->copy_stream, h->ev_piece[32], 0));
    FusedJob job;
    job.soa = h->d_queue; job.order = h->d_order; job.n = n; job.first_id = 0; job.ctr = h->d_ctr;
    job.stream = h->stream; job.avail = d_avail;
    if ((rc = pick_fused(h, tables, lds, n_iter, sqrt_threshold(outeredge), image, nullptr, nullptr, job)))
        return rc;
    // from here on the kernel is waiting for pieces: whatever happens, something must be published
    auto give_up = [&](int code) {
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, h->stream2, d_avail, ~0ull);
        (void)hipStreamSynchronize(h->stream2);
        (void)stream_sync(h);
        h->n_packets = 0;
        return code;
    };
    // piece boundaries: a small first piece (enough to occupy every lane a few times over) lets
    // the integrator start after a fraction of a millisecond of copying; the rest in equal parts
    std::vector<int64_t> bound((size_t)pieces + 1, n);
    bound[0] = 0;
    if (pieces > 1) {
        int64_t first = std::max<int64_t>(n / (4 * (int64_t)pieces), (int64_t)h->n_cu * BLOCK_PERSIST * 2);
        first = std::min(first, n / pieces);
        for (int p = 1; p < pieces; p++) bound[(size_t)p] = first + (n - first) * (p - 1) / (pieces - 1);
    }
    // fault injection for the tests of the give-up path: the last piece is never published
    const bool withhold = std::getenv("NXC_TEST_WITHHOLD_LAST_PIECE") != nullptr;
    for (int p = 0; p < pieces; p++) {
        const int64_t p0 = bound[(size_t)p], len = bound[(size_t)p + 1] - p0;
        if (len <= 0) continue;
        if (withhold && p == pieces - 1 && pieces > 1) break;
        hipError_t e = hipSuccess;
        for (int c = 0; c < 8 && e == hipSuccess; c++)
            e = hipMemcpyAsync(h->d_packets + c * n + p0, soa0 + c * n + p0, (size_t)len * 8,
                               hipMemcpyHostToDevice, h->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(h->ev_piece[p], h->copy_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(h->stream2, h->ev_piece[p], 0);
        unsigned long long *hist = h->d_piece_hist + (size_t)p * per_piece, *d_max = hist + NXC_ORDER_BINS;
        if (e == hipSuccess) e = hipMemsetAsync(d_max, 0, sizeof(unsigned long long), h->stream2);
        if (e != hipSuccess)
            return give_up(fail_hip("streamed upload", e));
        hipLaunchKernelGGL(k_speed_max<false>, dim3(flat_grid(h, len, NXC_BLOCK)), dim3(NXC_BLOCK), 0,
                           h->stream2, (const double *)(h->d_packets + p0), n, len, d_max);
        if ((rc = order_async(h, h->stream2, h->d_packets + p0, n, len, nullptr, 0.0, d_max, hist,
                              h->d_order + p0, h->d_queue + 8 * p0, true, (unsigned)p0)))
            return give_up(rc);
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, h->stream2, d_avail,
                           (unsigned long long)(p0 + len));
        if (hipGetLastError() != hipSuccess) return give_up(fail(NXC_ERR_HIP, "streamed upload: launch failed"));
    }
    h->have_order = true;            // pieces sorted one by one: still a valid queue order
    h->order_key = 0;
    h->streamed_pending = true;
    // the handle's stream continues after the ordering stream as well
    HIPCHK(hipEventRecord(h->ev_piece[32], h->stream2));
    HIPCHK(hipStreamWaitEvent(h->stream, h->ev_piece[32], 0));
    return NXC_OK;
    });
}

int nxc_integrate_const(nxc_handle *h, double step, int64_t n_iter, double outeredge,
                        uint32_t flags, double *traj_out, int64_t nrec, double *final_out,
                        int64_t *steps_out)
{
    return guarded([&]() -> int {
    int rc = need_forces(h);
    if (rc) return rc;
    const int64_t n = h->n_packets;
    if (n < 1) return fail(NXC_ERR_STATE, "no resident packets (nxc_packets_upload)");
    if (!(step > 0) || n_iter < 0) return fail(NXC_ERR_ARG, "step must be > 0, n_iter >= 0");
    const bool image = (flags & NXC_RUN_IMAGE) != 0;
    if (image && !h->have_image) return fail(NXC_ERR_STATE, "NXC_RUN_IMAGE without nxc_set_image");
    if (traj_out && nrec < n_iter + 1) return fail(NXC_ERR_ARG, "nrec must be >= n_iter + 1");
    if ((rc = speed_order_for_const(h))) return rc;

    const size_t col = (size_t)n * sizeof(double);
    double *d_final = nullptr;
    long long *d_steps = nullptr;
    if (!traj_out) {
        if (final_out) {
            if ((rc = ensure(reinterpret_cast<void **>(&h->d_scratch), &h->scratch_cap, 8 * col)))
                return rc;
            d_final = h->d_scratch;
        }
        if (steps_out) {
            if ((rc = ensure(reinterpret_cast<void **>(&h->d_steps), &h->steps_cap,
                             (size_t)n * sizeof(long long))))
                return rc;
            d_steps = h->d_steps;
        }
        if ((rc = launch_const(h, step, n_iter, outeredge, image, d_final, d_steps))) return rc;
    } else {
        // the dense `results` array of the reference (Output.py:376,419): pass 1 (persistent
        // integrator, + image) -> row offsets -> pass 2 writes the live records -> k_rows_densify
        // lays them out [column][record][packet] with the death record and the zero padding
        const size_t tbytes = (size_t)8 * (size_t)nrec * col;
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        if (tbytes > free_b && pool_bytes(h)) {
            pool_flush(h);
            HIPCHK(hipMemGetInfo(&free_b, &total_b));
        }
        if (tbytes > free_b)
            return fail(NXC_ERR_ARG, "trajectory buffer does not fit in device memory; run fewer "
                                     "packets per call (the reference chunks too, Input.py:219-222)");
        long long total = 0;
        if ((rc = count_rows(h, step, n_iter, outeredge, image, nullptr, &total))) return rc;
        // the caller's counters are those of pass 1 (work, samples); pass 2 re-does the steps
        DevCounters pass1;
        HIPCHK(hipMemcpyAsync(&pass1, h->d_ctr, sizeof pass1, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(stream_sync(h));
        d_final = h->d_scratch;
        d_steps = h->d_steps;
        if ((rc = write_records(h, false, tbytes))) return rc;
        h->rows_total = -1;
        const double *d_rec = static_cast<const double *>(h->d_rec);
        double *d_traj = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_traj), tbytes);
        if (e == hipSuccess) {
            const dim3 grid((unsigned)((n + NXC_BLOCK - 1) / NXC_BLOCK),
                            (unsigned)((nrec + NXC_DENSIFY_RECORDS - 1) / NXC_DENSIFY_RECORDS));
            hipLaunchKernelGGL(k_rows_densify, grid, dim3(NXC_BLOCK), 0, h->stream, d_rec,
                               (const long long *)h->d_offsets, (const double *)d_final,
                               (const long long *)d_steps, n, nrec, d_traj);
            e = hipGetLastError();
        }
        DevCounters pass2;
        if (e == hipSuccess)
            e = hipMemcpyAsync(&pass2, h->d_ctr, sizeof pass2, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(traj_out, d_traj, tbytes, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = stream_sync(h);
        if (e == hipSuccess) {
            pass1.unfinished += pass2.unfinished;        // rows the two passes disagree on
            e = hipMemcpyAsync(h->d_ctr, &pass1, sizeof pass1, hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) e = stream_sync(h);
        }
        if (d_traj) (void)hipFree(d_traj);
        if (e != hipSuccess)
            return fail_hip("trajectory run", e);
    }
    if (final_out)
        HIPCHK(hipMemcpyAsync(final_out, d_final, 8 * col, hipMemcpyDeviceToHost, h->stream));
    if (steps_out)
        HIPCHK(hipMemcpyAsync(steps_out, d_steps, (size_t)n * sizeof(long long),
                              hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

int nxc_integrate_const_rows(nxc_handle *h, double step, int64_t n_iter, double outeredge,
                             int64_t *lengths_out, int64_t *total_out)
{
    return guarded([&]() -> int {
    int rc = need_forces(h);
    if (rc) return rc;
    if (h->n_packets < 1) return fail(NXC_ERR_STATE, "no resident packets (nxc_packets_upload)");
    if (!(step > 0) || n_iter < 0 || !total_out) return fail(NXC_ERR_ARG, "bad arguments");
    if ((rc = speed_order_for_const(h))) return rc;
    long long total = 0;
    if ((rc = count_rows(h, step, n_iter, outeredge, false, lengths_out, &total))) return rc;
    *total_out = total;
    return NXC_OK;
    });
}

int nxc_rows_build(nxc_handle *h, int narrow, nxc_rows **out)
{
    return guarded([&]() -> int {
    if (!out) return fail(NXC_ERR_ARG, "out is null");
    if (!h || h->rows_total < 0 || h->rows_n != h->n_packets)
        return fail(NXC_ERR_STATE, "nxc_rows_build needs a preceding nxc_integrate_const_rows");
    return rows_build(h, narrow != 0, out);
    });
}

int nxc_rows_info(const nxc_rows *r, int64_t *total, int32_t *is_f32)
{
    if (!r) return fail(NXC_ERR_ARG, "null argument");
    if (total) *total = r->total;
    if (is_f32) *is_f32 = r->f32 ? 1 : 0;
    return NXC_OK;
}

int nxc_rows_free(nxc_handle *h, nxc_rows *r)
{
    if (!r) return NXC_OK;
    (void)hipSetDevice(r->device);
    if (h && h->stream) (void)stream_sync(h);
    if (h && h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    if (h && h->device == r->device) {
        pool_give(h, r->d_cols, r->cols_cap);
        pool_give(h, r->d_index, r->index_cap);
    } else {
        if (r->d_cols) (void)hipFree(r->d_cols);
        if (r->d_index) (void)hipFree(r->d_index);
    }
    delete r;
    return NXC_OK;
}

int nxc_rows_download(nxc_handle *h, const nxc_rows *r, int64_t first, int64_t count,
                      void *cols_out, void *index_out)
{
    return guarded([&]() -> int {
    int rc = rows_check(h, r, first, count);
    if (rc) return rc;
    if (count == 0) return NXC_OK;
    HIPCHK(hipSetDevice(h->device));
    // a finished store is immutable (nxc_rows_build synchronises before it returns), so the copy
    // goes on its own stream and may be issued from a second host thread (a file writer) while
    // the handle's thread launches the next run
    hipStream_t st = h->copy_stream;
    const size_t vsz = r->f32 ? 4 : 8, isz = r->f32 ? 4 : 8;
    if (cols_out)          // nine strided column pieces -> [9][count]
        HIPCHK(hipMemcpy2DAsync(cols_out, (size_t)count * vsz,
                                static_cast<const char *>(r->d_cols) + (size_t)first * vsz,
                                (size_t)r->total * vsz, (size_t)count * vsz, 9,
                                hipMemcpyDeviceToHost, st));
    if (index_out)
        HIPCHK(hipMemcpyAsync(index_out, static_cast<const char *>(r->d_index) + (size_t)first * isz,
                              (size_t)count * isz, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return NXC_OK;
    });
}

int nxc_image_accumulate_rows(nxc_handle *h, const nxc_rows *r, int64_t first, int64_t count)
{
    return guarded([&]() -> int {
    if (!h || !h->have_image) return fail(NXC_ERR_STATE, "nxc_set_image has not been called");
    int rc = rows_check(h, r, first, count);
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(DevCounters), h->stream));
    if (count == 0) return NXC_OK;
    // columns 1, 2, 3, 5, 7 of the store = x, y, z, vy, frac
    if (r->f32) {
        const float *c = static_cast<const float *>(r->d_cols) + first;
        const long long t = r->total;
        return image_run<float>(h, count, c + t, c + 2 * t, c + 3 * t, c + 5 * t, c + 7 * t);
    }
    const double *c = static_cast<const double *>(r->d_cols) + first;
    const long long t = r->total;
    return image_run<double>(h, count, c + t, c + 2 * t, c + 3 * t, c + 5 * t, c + 7 * t);
    });
}

int nxc_los_accumulate_rows(nxc_handle *h, const nxc_los_desc *d, int64_t S, const double *sc,
                            const nxc_rows *r, int64_t first, int64_t count, int64_t index_shift,
                            int64_t n_index,
                            double *radiance, int64_t *npackets, uint8_t *included,
                            int64_t used_cap, int64_t *used_pairs, int64_t *n_used)
{
    return guarded([&]() -> int {
    int rc = los_check(h, d, S, sc, count, radiance, npackets, included, n_index, used_cap,
                       used_pairs, n_used);
    if (rc) return rc;
    if ((rc = rows_check(h, r, first, count))) return rc;
    const long long t = r->total;
    if (r->f32) {
        const float *c = static_cast<const float *>(r->d_cols) + first;
        return los_run<float, int>(h, d, S, sc, count, c + t, c + 2 * t, c + 3 * t, c + 5 * t,
                                   c + 7 * t, static_cast<const int *>(r->d_index) + first, index_shift,
                                   n_index,
                                   radiance, npackets, included, used_cap, used_pairs, n_used);
    }
    const double *c = static_cast<const double *>(r->d_cols) + first;
    return los_run<double, long long>(h, d, S, sc, count, c + t, c + 2 * t, c + 3 * t, c + 5 * t,
                                      c + 7 * t, static_cast<const long long *>(r->d_index) + first,
                                      index_shift, n_index, radiance, npackets, included, used_cap, used_pairs,
                                      n_used);
    });
}

int nxc_rows_fetch(nxc_handle *h, double *rows_out)
{
    return guarded([&]() -> int { return rows_fetch(h, rows_out, false); });
}

int nxc_rows_fetch_f32(nxc_handle *h, float *rows_out)
{
    return guarded([&]() -> int { return rows_fetch(h, rows_out, true); });
}

int nxc_integrate_var(nxc_handle *h, double resolution, double outeredge, int64_t max_steps,
                      double *final_out, double *hstore_out)
{
    return guarded([&]() -> int {
    int rc = need_forces(h);
    if (rc) return rc;
    const int64_t n = h->n_packets;
    if (n < 1) return fail(NXC_ERR_STATE, "no resident packets (nxc_packets_upload)");
    if (!(resolution > 0) || !final_out || max_steps < 1) return fail(NXC_ERR_ARG, "bad arguments");
    if (h->have_bodies)
        return fail(NXC_ERR_STATE, "nxc_integrate_var: moons need the constant-step driver");
    const size_t col = (size_t)n * sizeof(double);
    if ((rc = ensure(reinterpret_cast<void **>(&h->d_scratch), &h->scratch_cap, 9 * col)))
        return rc;
    double *d_final = h->d_scratch, *d_hs = d_final + 8 * n;
    // the adaptive driver's queue: slow packets with much time left first (nxc_kernels.hpp:
    // flight_key); once per resident set
#ifdef NXC_VAR_TRACE      /* experiment: the packets as uploaded are the queue */
    if (std::getenv("NXC_TEST_VAR_NO_ORDER")) { h->have_order = false; } else
#endif
    if (h->order_key != 2 && (rc = order_on_device(h, -1.0, nullptr, 0, true))) return rc;
    HIPCHK(hipMemsetAsync(h->d_ctr, 0, sizeof(DevCounters), h->stream));
    int grid = 1;
    const bool full = h->F.grav && h->F.rad && h->F.loss == LOSS_PHOTO;
    // Two launch forms of the same arithmetic (nxc_kernels.hpp: k_var), by packets per lane: under
    // 24 (4.7e6 packets: the launch is mostly tail) with clock-rotated wave priorities and a merged
    // tail, above it plain (highest throughput).
    const double per_lane = (double)n / ((double)h->n_cu * BLOCK_PERSIST);
    bool fair = per_lane < NXC_VAR_FAIR_PACKETS_PER_LANE;
    if (const char *t = std::getenv("NXC_TEST_VAR_VARIANT"))          // tests: both forms at any size
        fair = t[0] == 'f';                                           // "fair" / "plain"
    int block = BLOCK_PERSIST;
    const size_t lds = ((h->force_bytes + 31) & ~size_t(31)) + (size_t)(block / 64) * NXC_WAVE_LDS_BYTES;
    auto launch = [&](auto kernel) -> int {
        int rc2;
        if ((rc2 = prep_kernel(kernel, lds))) return rc2;
        if ((rc2 = persistent_grid(h, kernel, &block, lds, n, &grid))) return rc2;
#ifdef NXC_VAR_ONE_WG_PER_CU          /* experiment: with -DNXC_BLOCK_PERSIST_N=512, two waves per SIMD */
        if (grid > h->n_cu) grid = h->n_cu;
#endif
        if ((rc2 = begin_timed(h))) return rc2;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, h->stream, h->F, h->d_blob,
                           (int64_t)h->force_bytes, n, h->have_order ? h->d_queue : h->d_packets,
                           h->have_order ? h->d_order : (const unsigned *)nullptr, resolution, outeredge,
                           (long long)max_steps, d_final, d_hs, h->d_ctr);
        return NXC_OK;
    };
    if (fair) rc = full ? launch(k_var<true, true>) : launch(k_var<false, true>);
    else rc = full ? launch(k_var<true, false>) : launch(k_var<false, false>);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    if ((rc = end_timed(h))) return rc;
    HIPCHK(hipMemcpyAsync(final_out, d_final, 8 * col, hipMemcpyDeviceToHost, h->stream));
    if (hstore_out)
        HIPCHK(hipMemcpyAsync(hstore_out, d_hs, col, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

#ifdef NXC_VAR_TRACE
extern "C" int nxc_debug_var_trace(nxc_handle *h, unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_var_trace), sizeof(unsigned long long) * 8 * 4096) == hipSuccess ? 0 : -1;
}
#endif

int nxc_image_accumulate(nxc_handle *h, int64_t p, const double *x, const double *y,
                         const double *z, const double *vy, const double *frac)
{
    return guarded([&]() -> int { return image_accumulate(h, p, x, y, z, vy, frac); });
}

int nxc_image_accumulate_f32(nxc_handle *h, int64_t p, const float *x, const float *y,
                             const float *z, const float *vy, const float *frac)
{
    return guarded([&]() -> int { return image_accumulate(h, p, x, y, z, vy, frac); });
}

int nxc_los_accumulate(nxc_handle *h, const nxc_los_desc *d, int64_t S, const double *sc,
                       int64_t P, const double *x, const double *y, const double *z,
                       const double *vy, const double *frac, const int64_t *index,
                       int64_t n_index, double *radiance, int64_t *npackets, uint8_t *included,
                       int64_t used_cap, int64_t *used_pairs, int64_t *n_used)
{
    return guarded([&]() -> int {
        return los_accumulate(h, d, S, sc, P, x, y, z, vy, frac, index, n_index, radiance, npackets,
                              included, used_cap, used_pairs, n_used);
    });
}

int nxc_los_accumulate_f32(nxc_handle *h, const nxc_los_desc *d, int64_t S, const double *sc,
                           int64_t P, const float *x, const float *y, const float *z,
                           const float *vy, const float *frac, const int64_t *index,
                           int64_t n_index, double *radiance, int64_t *npackets, uint8_t *included,
                           int64_t used_cap, int64_t *used_pairs, int64_t *n_used)
{
    return guarded([&]() -> int {
        return los_accumulate(h, d, S, sc, P, x, y, z, vy, frac, index, n_index, radiance, npackets,
                              included, used_cap, used_pairs, n_used);
    });
}

// ---- RCCL -------------------------------------------------------------------------------------
int nxc_comm_unique_id(uint8_t id[NXC_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == NXC_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id) return fail(NXC_ERR_ARG, "id is null");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return NXC_OK;
}

int nxc_comm_init(nxc_handle *h, const uint8_t id[NXC_UNIQUE_ID_BYTES], int rank, int nranks)
{
    if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(NXC_ERR_ARG, "bad arguments");
    int rc = rccl_load();
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    if (h->comm) { g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    const ncclResult_t r = g_rccl.CommInitRank(&h->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        h->comm = nullptr;
        if (r == ncclInvalidUsage || r == ncclInvalidArgument)
            // what RCCL reports when two ranks of the communicator sit on one device
            return fail(NXC_ERR_ARG, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r) +
                                     " -- every rank needs its own GPU (one process per device)");
        return fail(NXC_ERR_RCCL, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    }
    h->rank = rank;
    h->nranks = nranks;
    h->coll_pending = false;
    h->abort_requested.store(false);
    if (const char *t = std::getenv("NXC_COLLECTIVE_TIMEOUT_S")) {
        const double v = std::atof(t);
        if (v > 0.0) h->coll_timeout_s = v;
    }
    return NXC_OK;
}

int nxc_comm_destroy(nxc_handle *h)
{
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    if (h->comm && g_rccl.ok) {
        HIPCHK(hipSetDevice(h->device));
        NCCLCHK(g_rccl.CommDestroy(h->comm));
    }
    h->comm = nullptr;
    h->nranks = 1;
    h->rank = 0;
    return NXC_OK;
}

int nxc_comm_set_timeout(nxc_handle *h, double seconds)
{
    if (!h || !(seconds > 0.0)) return fail(NXC_ERR_ARG, "bad arguments");
    h->coll_timeout_s = seconds;
    return NXC_OK;
}

// In front of a collective: the event that separates this process's own queued work from it.
static int mark_collective(nxc_handle *h)
{
    if (!h->coll_pending) {
        if (!h->ev_pre_coll) HIPCHK(hipEventCreateWithFlags(&h->ev_pre_coll, hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->ev_pre_coll, h->stream));
    }
    h->coll_pending = true;
    return NXC_OK;
}

int nxc_comm_request_abort(nxc_handle *h)
{
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    h->abort_requested.store(true);
    return NXC_OK;
}

int nxc_comm_test_stall(nxc_handle *h, double seconds)
{
    if (!h || !(seconds >= 0.0) || seconds > 30.0) return fail(NXC_ERR_ARG, "bad arguments");
    if (!h->comm) return fail(NXC_ERR_STATE, "nxc_comm_init has not been called");
    HIPCHK(hipSetDevice(h->device));
    if (int rc = mark_collective(h)) return rc;
    hipLaunchKernelGGL(k_stall, dim3(1), dim3(64), 0, h->stream,
                       (unsigned long long)(seconds * 1e8), (unsigned long long *)nullptr);
    HIPCHK(hipGetLastError());
    return NXC_OK;
}

// A rank that has been told of a peer's failure does not enter another collective.
static int refuse_after_abort_request(nxc_handle *h)
{
    if (!h->abort_requested.load()) return NXC_OK;
    if (h->comm && g_rccl.ok) (void)g_rccl.CommAbort(h->comm);
    h->comm = nullptr;
    h->nranks = 1;
    h->rank = 0;
    h->coll_pending = false;
    return fail(NXC_ERR_RCCL, "a peer rank reported a failure; the communicator was aborted");
}

int nxc_comm_abort(nxc_handle *h)
{
    if (!h) return fail(NXC_ERR_ARG, "null handle");
    if (h->comm && g_rccl.ok) {
        (void)hipSetDevice(h->device);
        (void)g_rccl.CommAbort(h->comm);
    }
    h->comm = nullptr;
    h->nranks = 1;
    h->rank = 0;
    h->coll_pending = false;
    return NXC_OK;
}

int nxc_image_allreduce(nxc_handle *h)
{
    if (!h || !h->have_image) return fail(NXC_ERR_STATE, "nxc_set_image has not been called");
    if (!h->comm) return fail(NXC_ERR_STATE, "nxc_comm_init has not been called");
    HIPCHK(hipSetDevice(h->device));
    if (int rc = refuse_after_abort_request(h)) return rc;
    // one collective: weights and (integer-valued fp64) counts are interleaved in one array
    if (int rc = mark_collective(h)) return rc;
    NCCLCHK(g_rccl.AllReduce(h->d_image, h->d_image, 2 * h->npix, ncclFloat64, ncclSum, h->comm,
                             h->stream));
    return NXC_OK;
}

// In-place all-reduce of n host doubles (n small: scalars of the bench, the S radiances + S
// packet counts of a set of lines of sight): staged through device scratch on the handle's stream.
static int allreduce_host(nxc_handle *h, double *values, int64_t n, ncclRedOp_t op)
{
    if (!h || !values || n < 1) return fail(NXC_ERR_ARG, "bad arguments");
    if (!h->comm) return fail(NXC_ERR_STATE, "nxc_comm_init has not been called");
    HIPCHK(hipSetDevice(h->device));
    double *d = h->d_reduce;
    if (n > 1) {
        int rc = ensure(reinterpret_cast<void **>(&h->d_reduce_n), &h->reduce_n_cap, (size_t)n * 8);
        if (rc) return rc;
        d = h->d_reduce_n;
    }
    if (int rc = refuse_after_abort_request(h)) return rc;
    if (int rc = mark_collective(h)) return rc;
    HIPCHK(hipMemcpyAsync(d, values, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    NCCLCHK(g_rccl.AllReduce(d, d, (size_t)n, ncclFloat64, op, h->comm, h->stream));
    HIPCHK(hipMemcpyAsync(values, d, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
}

int nxc_allreduce_max_f64(nxc_handle *h, double *value) { return allreduce_host(h, value, 1, ncclMax); }

int nxc_allreduce_sum_f64(nxc_handle *h, double *value) { return allreduce_host(h, value, 1, ncclSum); }

int nxc_allreduce_f64(nxc_handle *h, double *values, int64_t n)
{
    return allreduce_host(h, values, n, ncclSum);
}

int nxc_barrier(nxc_handle *h)
{
    double v = 0.0;
    return nxc_allreduce_max_f64(h, &v);
}

int nxc_stream_copy_gbs(nxc_handle *h, int64_t bytes, int reps, double *gbs)
{
    return guarded([&]() -> int {
    if (!h || !gbs || bytes < (1 << 20) || reps < 1) return fail(NXC_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const int64_t n16 = bytes / 16;
    char *buf = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&buf), (size_t)n16 * 32));
    hipError_t e = hipMemsetAsync(buf, 1, (size_t)n16 * 32, h->stream);
    float best = 0.f;
    hipEvent_t a = nullptr, b = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    for (int r = 0; r <= reps && e == hipSuccess; r++) {         // first round warms up
        e = hipEventRecord(a, h->stream);
        const int64_t per_block = (int64_t)NXC_BLOCK * NXC_COPY_UNROLL;
        hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)((n16 + per_block - 1) / per_block)),
                           dim3(NXC_BLOCK), 0, h->stream,
                           reinterpret_cast<const nxc_v2d *>(buf),
                           reinterpret_cast<nxc_v2d *>(buf + (size_t)n16 * 16), n16);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(b, h->stream);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (r > 0 && e == hipSuccess && (best == 0.f || ms < best)) best = ms;
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail_hip("stream copy", e);
    *gbs = 2.0 * (double)n16 * 16.0 / ((double)best * 1e-3) / 1e9;      // bytes read + written
    return NXC_OK;
    });
}

int nxc_shader_clock_mhz(nxc_handle *h, double *mhz)
{
    return guarded([&]() -> int {
    if (!h || !mhz) return fail(NXC_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const int blocks = h->n_cu, waves = blocks * (BLOCK_PERSIST / 64);
    int rc = ensure(reinterpret_cast<void **>(&h->d_scratch), &h->scratch_cap,
                    (size_t)waves * 3 * sizeof(unsigned long long));
    if (rc) return rc;
    unsigned long long *d = reinterpret_cast<unsigned long long *>(h->d_scratch);
    std::vector<unsigned long long> got((size_t)waves * 3);
    for (int round = 0; round < 2; round++) {                    // the first brings the clock up
        hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(BLOCK_PERSIST), 0, h->stream, d,
                           round ? 40000 : 400000, 1.0);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(got.data(), d, got.size() * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    std::vector<double> f;
    for (int w = 0; w < waves; w++)
        if (got[3 * (size_t)w + 1] > 0)
            f.push_back((double)got[3 * (size_t)w] / (double)got[3 * (size_t)w + 1] * 100.0);
    if (f.empty()) return fail(NXC_ERR_HIP, "no clock stamps came back");
    std::nth_element(f.begin(), f.begin() + f.size() / 2, f.end());
    *mhz = f[f.size() / 2];
    return NXC_OK;
    });
}

int nxc_pcg64_uniforms(nxc_handle *h, const uint64_t state[2], const uint64_t inc[2], int64_t n,
                       int64_t row0, int64_t count, int32_t nvec, double *out)
{
    return guarded([&]() -> int {
    if (!h || !state || !inc || !out || n < 1 || row0 < 0 || count < 1 || row0 + count > n ||
        nvec < 1 || nvec > NXC_PCG_VECS || n >= ((int64_t)1 << (NXC_PCG_BITS - 1)) || !(inc[1] & 1ull))
        return fail(NXC_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const std::vector<u128> maps = pcg_tables(((u128)inc[0] << 64) | inc[1], n);
    const size_t map_bytes = maps.size() * sizeof(u128), out_bytes = (size_t)nvec * count * 8;
    int rc = ensure(reinterpret_cast<void **>(&h->d_scratch), &h->scratch_cap, map_bytes + out_bytes);
    if (rc) return rc;
    unsigned char *base = reinterpret_cast<unsigned char *>(h->d_scratch);
    HIPCHK(hipMemcpyAsync(base, maps.data(), map_bytes, hipMemcpyHostToDevice, h->stream));
    PcgK P{};
    P.state = ((u128)state[0] << 64) | state[1];
    P.row0 = row0;
    P.maps = reinterpret_cast<const nxc_u128 *>(base);
    hipLaunchKernelGGL(k_pcg_uniforms, dim3(flat_grid(h, count, NXC_BLOCK)), dim3(NXC_BLOCK), 0,
                       h->stream, P, (int)nvec, count, reinterpret_cast<double *>(base + map_bytes));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, base + map_bytes, out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

int nxc_math_batch(nxc_handle *h, int which, int64_t n, const double *in, const double *in2,
                   double *out)
{
    return guarded([&]() -> int {
    if (!h || n < 0 || (n && (!in || !out)) || which < 0 || which > 4 || (which == 4 && !in2))
        return fail(NXC_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (n == 0) return NXC_OK;
    const size_t col = (size_t)n * sizeof(double);
    int rc = ensure(reinterpret_cast<void **>(&h->d_scratch), &h->scratch_cap, 3 * col);
    if (rc) return rc;
    double *d = h->d_scratch;
    HIPCHK(hipMemcpyAsync(d, in, col, hipMemcpyHostToDevice, h->stream));
    if (in2) HIPCHK(hipMemcpyAsync(d + n, in2, col, hipMemcpyHostToDevice, h->stream));
    if (!h->d_blob && (rc = upload_blob(h))) return rc;          // the header with nxc_log's table
    hipLaunchKernelGGL(k_math, dim3(flat_grid(h, n, 256)), dim3(256), NXC_HEADER_BYTES, h->stream,
                       h->d_blob, which, n, d, d + n, d + 2 * n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d + 2 * n, col, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(stream_sync(h));
    return NXC_OK;
    });
}

}  // extern "C"
