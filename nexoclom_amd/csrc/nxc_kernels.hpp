// nxc_kernels.hpp -- the gfx950 kernels of the nexoclom hot path.
//
//   k_state         a-2  element-wise force/loss model
//   k_rk5_step      a-1  one Dormand-Prince step per packet, per-packet step size
//   k_const_fused   a-3 + a-6..a-8  persistent LANE-REFILL integrator: a lane keeps its packet's
//                        state in registers until the packet dies, then takes the next packet
//                        from a global queue (claimed in chunks, handed out inside the wave by
//                        ballot + prefix rank), so waves stay full although lifetimes differ by
//                        three orders of magnitude.  Every stored record is binned straight into
//                        the image; the (N, 8, nsteps) trajectory tensor is never materialised.
//                        <ROWS>: the same loop writes every live record to its row of the
//                        compact trajectory instead (k_rows_transpose / k_rows_densify shape it).
//   k_var           a-4  adaptive-step driver, same lane-refill structure
//   k_image         a-6..a-8  image of stored samples (HBM-bound: 40 B/sample in)
//
// All lookup tables (radiation acceleration, g-values, bin edges) are staged once per workgroup
// into LDS from one packed blob.
#pragma once
#include "nxc_device.hpp"

struct DevCounters {
    unsigned long long particle_steps, samples, samples_binned, nonfinite, bad_step, neg_frac,
        unfinished, queue_head;
    unsigned long long wave_trips;   // trips of a wave through a persistent step loop (measurement)
#ifdef NXC_STAMPS               // diagnostic build only: summed s_memtime shares of the loop segments
    unsigned long long stamp[8];
#endif
};

#ifdef NXC_STAMPS
// In-kernel stamp (cdna_hip_programming.md, "In-kernel stamps"): one asm statement, fenced by
// scheduling barriers.  Diagnostic builds only; the product build has none.
NXC_DEV unsigned long long nxc_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define NXC_STAMP(k) do { const unsigned long long t_ = nxc_stamp(); seg[k] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define NXC_STAMP(k) do { } while (0)
#endif

constexpr int NXC_BLOCK = 256;      // threads per workgroup of the flat kernels (4 waves)
// The persistent kernels run ONE 12-wave workgroup per CU (3 waves per SIMD, <= 168 VGPRs): the
// waves of a workgroup share a single LDS copy of the tables (~86 KB for Na with a 512^2 image),
// which leaves room for the per-wave packet staging blocks and image queues (4.8 KB per wave)
// inside the CU's 160 KB.
#ifndef NXC_BLOCK_PERSIST_N          // overridable for occupancy experiments (tools/)
#define NXC_BLOCK_PERSIST_N 768
#endif
#ifndef NXC_CHUNK_N
#define NXC_CHUNK_N 32
#endif
constexpr int NXC_BLOCK_PERSIST = NXC_BLOCK_PERSIST_N;
constexpr int NXC_CHUNK = NXC_CHUNK_N;   // packets claimed from the global queue per atomic (<= 64: one per lane)
static_assert(NXC_CHUNK >= 1 && NXC_CHUNK <= 64, "a chunk is loaded by one wave");
constexpr int NXC_WAVE_STAGE_BYTES = NXC_CHUNK * 9 * 8;   // per-wave LDS staging: 8 columns + packet id
// per-wave LDS of the persistent kernels: the packet staging block, then the image queue
constexpr int NXC_WAVE_LDS_BYTES = NXC_WAVE_STAGE_BYTES + NXC_IMGQ_BYTES;
// the ROWS variant stages two more columns (first row, row count) and has no image queue
constexpr int NXC_WAVE_LDS_BYTES_ROWS = NXC_CHUNK * 11 * 8;

// Cooperative copy of the first `bytes` (multiple of 8) of the table blob into LDS, then the
// derived per-launch constants of the header.
NXC_DEV void stage_tables(const unsigned char *__restrict__ blob, int64_t bytes)
{
    // lds_f64 & co. address the block absolutely: it must start at LDS address 0 (true as long as
    // no kernel that uses it declares static __shared__ data; a link-time constant, so the test
    // costs nothing)
    if ((unsigned)(size_t)(NXC_LDS_AS unsigned char *)nxc_lds != 0u) __builtin_trap();
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(blob);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(nxc_lds);
    const int64_t words = bytes >> 3;
    for (int64_t w = threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
    __syncthreads();
    if (threadIdx.x == 0) {
        LdsHeader &H = lds_header_rw();
        H.Wt.rs_1e6 = nxc_recip_seed(1e6);
        H.Wt.rs_apix = nxc_mid_range(H.G.apix_cm2) ? nxc_recip_seed(H.G.apix_cm2)
                                                   : __builtin_nan("");
    }
    __syncthreads();
}

// stage_tables + the loop's rarely used kernel arguments parked in the LDS header (LoopK).
NXC_DEV void stage_tables_and_args(const unsigned char *__restrict__ blob, int64_t bytes,
                                   const double *soa0, const unsigned *order, double *final_out,
                                   long long *steps_out, unsigned long long *head, long long n,
                                   const long long *offsets = nullptr,
                                   const unsigned long long *avail = nullptr)
{
    stage_tables(blob, bytes);
    if (threadIdx.x == 0) {
        LoopK &L = lds_header_rw().L;
        L.soa0 = soa0; L.order = order; L.final_out = final_out; L.steps_out = steps_out;
        L.head = head; L.n = n; L.offsets = offsets; L.avail = avail;
        L.q_rec = order ? 8 : 1; L.q_col = order ? 1 : n;
    }
    __syncthreads();
}

NXC_DEV unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

NXC_DEV void flush_counter(unsigned long long *dst, unsigned long long v)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

// Up to three counters of a whole WORKGROUP: summed in LDS (`lds`: 24 bytes nobody else uses any
// more; every thread of the block must call), then one atomic each.  Returning or not, atomics on
// one address -- one cache line -- go through at about 80 million a second, so a launch of 4096
// waves that each add to two counters as they all finish spends 0.1 ms doing so.
NXC_DEV void flush_counters_wg(unsigned long long *lds, unsigned long long *d0, unsigned long long v0,
                               unsigned long long *d1, unsigned long long v1,
                               unsigned long long *d2, unsigned long long v2)
{
    __syncthreads();
    if (threadIdx.x < 3) lds[threadIdx.x] = 0ull;
    __syncthreads();
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
    if ((threadIdx.x & 63) == 0) {
        if (v0) atomicAdd(&lds[0], v0);
        if (v1) atomicAdd(&lds[1], v1);
        if (v2) atomicAdd(&lds[2], v2);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (lds[0]) atomicAdd(d0, lds[0]);
        if (lds[1]) atomicAdd(d1, lds[1]);
        if (lds[2]) atomicAdd(d2, lds[2]);
    }
}

NXC_DEV long long wave_bcast0(long long v)
{
    int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NXC_BLOCK)
k_state(ForceK F, const unsigned char *__restrict__ blob, int64_t stage_bytes, int64_t n,
        const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
        const double *__restrict__ vy, double *__restrict__ ax, double *__restrict__ ay,
        double *__restrict__ az, double *__restrict__ ion)
{
    stage_tables(blob, stage_bytes);
    const LutView T = lut_view(F.tab);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double a0, a1, a2, l;
        state_eval<false>(F, T, x[i], y[i], z[i], vy[i], a0, a1, a2, l);
        ax[i] = a0; ay[i] = a1; az[i] = a2; ion[i] = l;
    }
}

template <bool DELTA>
__global__ void __launch_bounds__(NXC_BLOCK)
k_rk5_step(ForceK F, const unsigned char *__restrict__ blob, int64_t stage_bytes, int64_t n,
           const double *__restrict__ in, const double *__restrict__ hstep,
           double *__restrict__ out, double *__restrict__ delta)
{
    stage_tables(blob, stage_bytes);
    const LutView T = lut_view(F.tab);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double s[8], d[8];
#pragma unroll
        for (int c = 0; c < 8; c++) s[c] = in[c * n + i];
        rk5_step<DELTA, false>(F, T, s, hstep[i], StepW{}, d);
#pragma unroll
        for (int c = 0; c < 8; c++) out[c * n + i] = s[c];
        if (DELTA) {
#pragma unroll
            for (int c = 0; c < 8; c++) delta[c * n + i] = d[c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Trajectory records.  The trajectory-producing runs (the reference's default user flow:
// Output.py:435-449 builds X from every record, save() keeps those with frac > 0, :523-524) use
// the same persistent lane-refill integrator as the fused image pass (k_const_fused<ROWS>): a
// first pass counts every packet's live records, the host turns the counts into row offsets, and
// the second pass writes record k of packet i at row offsets[i] + k.  A lane owns its packet for
// the packet's whole life, so its records are consecutive rows; they are written as 80-byte
// array-of-structures records {t, x, y, z, vx, vy, vz, frac, lossfrac, packet index} (five
// aligned 16-byte stores per record; a 128-byte line is completed by two consecutive steps of
// one lane), and k_rows_transpose turns the finished block into the columns the host wants
// (struct-of-arrays, optionally narrowed to float32 / int32: save()'s down-cast, Output.py:528-543).
// ROWS = 1: records of ten doubles; ROWS = 2: the same record narrowed on the way out (nine floats
// and an int32: save()'s down-cast, Output.py:528-543, which every catalogued Output goes
// through) -- half the store requests and half the bytes the transposition has to read.
constexpr int NXC_REC_DOUBLES = 10;
constexpr int NXC_REC_FLOATS = 10;
typedef float nxc_v4f __attribute__((ext_vector_type(4)));
typedef float nxc_v2f __attribute__((ext_vector_type(2)));
typedef float nxc_v4f_a8 __attribute__((ext_vector_type(4), aligned(8)));

template <int ROWS>
NXC_DEV void store_record(void *__restrict__ rec, long long row, const double (&s)[8],
                          double lossfrac, long long id)
{
    if (ROWS == 2) {
        float *p = static_cast<float *>(rec) + row * NXC_REC_FLOATS;        // 8-byte aligned
        nxc_v4f a, b;
        nxc_v2f c;
        a.x = (float)s[0]; a.y = (float)s[1]; a.z = (float)s[2]; a.w = (float)s[3];
        b.x = (float)s[4]; b.y = (float)s[5]; b.z = (float)s[6]; b.w = (float)s[7];
        c.x = (float)lossfrac; c.y = __int_as_float((int)id);
        *reinterpret_cast<nxc_v4f_a8 *>(p) = a;
        *reinterpret_cast<nxc_v4f_a8 *>(p + 4) = b;
        *reinterpret_cast<nxc_v2f *>(p + 8) = c;
    } else {
        nxc_v2d *p = reinterpret_cast<nxc_v2d *>(static_cast<double *>(rec) + row * NXC_REC_DOUBLES);
        nxc_v2d a, b, c, d, e;                                              // 16-byte aligned
        a.x = s[0]; a.y = s[1]; b.x = s[2]; b.y = s[3]; c.x = s[4]; c.y = s[5]; d.x = s[6]; d.y = s[7];
        e.x = lossfrac; e.y = __longlong_as_double(id);
        p[0] = a; p[1] = b; p[2] = c; p[3] = d; p[4] = e;
    }
}

// rec[total][10] -> cols[9][total] and index[total].  <double, long long>: fp64 records to fp64 /
// int64 columns; <float, int>: narrowed records to float32 / int32 columns.  A block moves 256
// records through LDS: coalesced 8-byte reads, coalesced column writes.
constexpr int NXC_TR_ROWS = 256;
template <typename T, typename I>
__global__ void __launch_bounds__(NXC_TR_ROWS)
k_rows_transpose(const void *__restrict__ rec_, long long total, T *__restrict__ cols,
                 I *__restrict__ index)
{
    static_assert(sizeof(T) == sizeof(I), "a record is ten equal-sized slots");
    typedef T pair_t __attribute__((ext_vector_type(2)));
    __shared__ pair_t tile[NXC_TR_ROWS * 5];
    const T *rec = static_cast<const T *>(rec_);
    for (long long base = (long long)blockIdx.x * NXC_TR_ROWS; base < total;
         base += (long long)gridDim.x * NXC_TR_ROWS) {
        const int nrow = (int)((total - base) < NXC_TR_ROWS ? (total - base) : NXC_TR_ROWS);
        const pair_t *src = reinterpret_cast<const pair_t *>(rec + base * 10);
        for (int w = threadIdx.x; w < nrow * 5; w += NXC_TR_ROWS) tile[w] = src[w];
        __syncthreads();
        if ((int)threadIdx.x < nrow) {
            const T *r = reinterpret_cast<const T *>(tile) + threadIdx.x * 10;
            const long long row = base + threadIdx.x;
#pragma unroll
            for (int c = 0; c < 9; c++) cols[c * total + row] = r[c];
            I id;
            __builtin_memcpy(&id, &r[9], sizeof id);
            index[row] = id;
        }
        __syncthreads();
    }
}

// The reference's dense `results` array (Output.py:376,419), transposed: traj[c][k][i] for
// c < 8, k < nrec.  Record k of packet i is its live record k (k < len), the state in which it
// died (k == steps[i] when that record is not live: frac = 0 and t = 0 but the position stays,
// Output.py:413-419), and zero afterwards.  Thread i walks its own consecutive records; every
// store is coalesced across the packets.  blockIdx.y cuts the record axis.
constexpr int NXC_DENSIFY_RECORDS = 32;
__global__ void __launch_bounds__(NXC_BLOCK)
k_rows_densify(const double *__restrict__ rec, const long long *__restrict__ offsets,
               const double *__restrict__ final_soa, const long long *__restrict__ steps,
               int64_t n, int64_t nrec, double *__restrict__ traj)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long off = offsets[i], len = offsets[i + 1] - off, last = steps[i];
    const int64_t k0 = (int64_t)blockIdx.y * NXC_DENSIFY_RECORDS;
    const int64_t k1 = k0 + NXC_DENSIFY_RECORDS < nrec ? k0 + NXC_DENSIFY_RECORDS : nrec;
    for (int64_t k = k0; k < k1; k++) {
        double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (k < len) {
            const nxc_v2d *r = reinterpret_cast<const nxc_v2d *>(rec + (off + k) * NXC_REC_DOUBLES);
#pragma unroll
            for (int c = 0; c < 4; c++) { const nxc_v2d t = r[c]; v[2 * c] = t.x; v[2 * c + 1] = t.y; }
        } else if (k == last) {
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = final_soa[c * n + i];
        }
#pragma unroll
        for (int c = 0; c < 8; c++) traj[((int64_t)c * nrec + k) * n + i] = v[c];
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-level packet queue.  A wave claims NXC_CHUNK consecutive queue positions with one atomicAdd
// on the global head and copies their 8 state columns into its own LDS staging block (coalesced:
// lane l loads position base + l).  Lanes whose packet has died are then served from that block in
// lane order (prefix rank over the ballot) with LDS reads only, so the steady-state loop contains
// no global load and never has to wait (in-order vmcnt) behind its own in-flight image atomics.
// All control flow here is wave-uniform.  Returns the packet index or -1; on success the lane's
// state is in s[].
// The queue holds the packets in decreasing launch speed (nxc_api.hip: order_on_device sorts the
// indices and writes a permuted copy of the state columns, so a chunk is contiguous in memory;
// `ids` maps queue position -> packet index, null = identity): long-lived packets start first and
// the short-lived ones fill the lanes at the end, which shortens the tail where few lanes still
// hold a packet (longest-processing-time first).  The order changes nothing in any packet's
// result.
// The claim for the NEXT chunk is issued as soon as the current one is loaded, so the returning
// atomic's round trip runs under that chunk's packets; a wave leaves only after the claim it
// holds turned out to lie beyond the queue's end, so no claimed chunk is ever dropped.
// Streamed upload (nxc_integrate_const_streamed): the queue is still being filled while the kernel
// runs -- pieces of it cross PCIe, are put into queue order by small kernels on another stream, and
// are then PUBLISHED: *avail = number of queue positions that are ready (k_publish: release store at
// agent scope after the producing kernels have completed).  A wave that claimed positions beyond
// that waits here (acquire load at agent scope, s_sleep between polls).  Bounded: the host
// publishes ~0 when it gives up, and a wave that has waited three seconds of s_memrealtime
// (100 MHz) gives up itself; either way the wave treats the queue as drained and the launch is
// reported as unfinished.
NXC_DEV bool wait_published(const unsigned long long *avail, unsigned long long need)
{
    unsigned long long t0 = 0;
    bool timing = false;
    for (;;) {
        const unsigned long long v =
            __hip_atomic_load(avail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (v == ~0ull) return false;
        if (v >= need) return true;
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (!timing) { t0 = now; timing = true; }
        else if (now - t0 > 300000000ull) return false;
        __builtin_amdgcn_s_sleep(32);
    }
}

__global__ void k_publish(unsigned long long *avail, unsigned long long value)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        __hip_atomic_store(avail, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

struct WaveQueue {
    long long pending = 0;      // lane 0: queue position claimed for the next reload
    int c_pos = 0, c_cnt = 0;
    bool drained = false, stalled = false;

    // The head pointer comes out of the LDS header as a generic pointer; as such the returning
    // atomic would be a FLAT instruction, which counts in lgkmcnt as well -- and the step loop's
    // next wait for an LDS read would then wait out the atomic's trip to memory.
    static NXC_DEV long long claim(unsigned long long *head)
    {
        return (long long)__hip_atomic_fetch_add((NXC_GLOBAL_AS unsigned long long *)head,
                                                 (unsigned long long)NXC_CHUNK, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
    }

    NXC_DEV void start()
    {
        if ((threadIdx.x & 63) == 0) pending = claim(lds_header().L.head);
    }

    // ROWS: the packet's first row and row count (offsets[id], offsets[id + 1] - offsets[id]) are
    // staged with it and returned in row0 / nrow.
    // STREAMED: the queue is still being filled (wait_published above); a template flag, so that
    // the resident passes' code -- the bench kernel among them -- carries none of it.
    template <bool ROWS = false, bool STREAMED = false>
    NXC_DEV long long refill(bool need, int stage_off, double (&s)[8], long long *row0 = nullptr,
                             long long *nrow = nullptr)
    {
        const unsigned long long mask = __ballot(need);
        long long mine = -1;
        if (mask == 0 || drained) return mine;
        const int want = __popcll(mask);
        const int lane = threadIdx.x & 63;
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        NXC_LDS_AS double *stage = (NXC_LDS_AS double *)(unsigned)stage_off;   // the block starts at LDS address 0
        int served = 0;
        while (served < want) {
            if (c_pos >= c_cnt) {
                const LoopK &L = lds_header().L;
                const long long n = L.n;
                const double *__restrict__ soa0 = L.soa0;
                const unsigned *__restrict__ ids = L.order;
                const long long b = wave_bcast0(pending);
                if (b >= n) { drained = true; break; }
                c_cnt = (b + NXC_CHUNK <= n) ? NXC_CHUNK : (int)(n - b);
                c_pos = 0;
                if (STREAMED) {
                    if (!wait_published(L.avail, (unsigned long long)(b + c_cnt))) {
                        drained = true; stalled = true; c_cnt = 0;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (lane < c_cnt) {
                    // All nine loads are issued before the first LDS store, through pointers in
                    // the GLOBAL address space: as generic pointers (they come out of the LDS
                    // header) the compiler must assume that they may alias the staging block and
                    // emits load, wait, store, load, wait, store ... -- nine memory round trips in
                    // a row for every chunk (round 3: seen in the ISA, 3 % of a wave's time).
                    const long long q_col = L.q_col;
                    const NXC_GLOBAL_AS double *g = (const NXC_GLOBAL_AS double *)soa0 + (b + lane) * L.q_rec;
                    const NXC_GLOBAL_AS unsigned *gid = (const NXC_GLOBAL_AS unsigned *)ids;
                    double col[8];
#pragma unroll
                    for (int c = 0; c < 8; c++) col[c] = g[c * q_col];
                    const long long pid = gid ? (long long)gid[b + lane] : b + lane;
                    long long o0 = 0, o1 = 0;
                    if (ROWS) {
                        const NXC_GLOBAL_AS long long *offs = (const NXC_GLOBAL_AS long long *)L.offsets;
                        o0 = offs[pid]; o1 = offs[pid + 1];
                    }
#pragma unroll
                    for (int c = 0; c < 8; c++) stage[c * NXC_CHUNK + lane] = col[c];
                    stage[8 * NXC_CHUNK + lane] = __longlong_as_double(pid);
                    if (ROWS) {
                        stage[9 * NXC_CHUNK + lane] = __longlong_as_double(o0);
                        stage[10 * NXC_CHUNK + lane] = __longlong_as_double(o1 - o0);
                    }
                }
                if (lane == 0) pending = claim(L.head);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const int room = c_cnt - c_pos;
            const int take = room < (want - served) ? room : want - served;
            if (need && rank >= served && rank < served + take) {
                const int slot = c_pos + (rank - served);
                mine = __double_as_longlong(stage[8 * NXC_CHUNK + slot]);
#pragma unroll
                for (int c = 0; c < 8; c++) s[c] = stage[c * NXC_CHUNK + slot];
                if (ROWS) {
                    *row0 = __double_as_longlong(stage[9 * NXC_CHUNK + slot]);
                    *nrow = __double_as_longlong(stage[10 * NXC_CHUNK + slot]);
                }
            }
            c_pos += take;
            served += take;
        }
        return mine;
    }
};

// One sample per lane off the image queue, located: pixel, radial velocity and masked fraction.
// IMAGE == 1: the queue holds located samples; IMAGE == 2: it holds the float32 samples that passed
// image_frame_test, and the rest of image_locate happens here, on full waves.
template <int IMAGE>
NXC_DEV bool image_pop(ImageQueue &queue, int imgq_off, const ImageRegs &IR, int &p, double &rv,
                       double &fw, unsigned long long &nonfinite)
{
    if (IMAGE == 2) {
        double x = 0, y = 0, z = 0, vy = 0, frac = 0;
        const bool ok = queue.pop_sample(imgq_off, x, y, z, vy, frac);
        if (ok) p = image_locate_core(lds_header().G, IR, x, y, z, vy, frac, rv, fw, nonfinite);
        return ok && p >= 0;
    }
    return queue.pop(imgq_off, p, rv, fw);
}

// Persistent lane-refill constant-step integrator (+ fused image).  The grid is sized to the
// machine (blocks = CUs x resident blocks), not to n; every wave leaves its loop when the queue
// is drained and none of its lanes holds a live packet.
// ROWS != 0 (never with IMAGE): every live record -- the initial state and the state after each step
// while frac > 0 -- goes to row offsets[id] + k of rec[total][10] (doubles, or floats for ROWS = 2) with lossfrac accumulated as
// (lossfrac + frac_before) - frac_after per step (Output.py:420-421) from 0.
template <int IMAGE, bool BOUNCE, bool FULL, bool NBODY = false, int ROWS = 0, bool STREAMED = false>
__global__ void __launch_bounds__(NXC_BLOCK_PERSIST)
k_const_fused(ForceK F, const unsigned char *__restrict__ blob,
              int64_t stage_bytes, int64_t n, const double *__restrict__ soa0,
              const unsigned *__restrict__ order, int64_t first_id, int64_t n_iter,
              double edge2, double *__restrict__ final_out,
              long long *__restrict__ steps_out, double *__restrict__ acc2,
              DevCounters *__restrict__ ctr, const double *__restrict__ moon_pos = nullptr,
              const long long *__restrict__ offsets = nullptr, void *__restrict__ rec = nullptr,
              const unsigned long long *__restrict__ avail = nullptr)
{
    static_assert(!(IMAGE && ROWS), "the rows pass has no image");
    static_assert(!(STREAMED && (ROWS || BOUNCE || NBODY)), "the streamed pass is the plain one");
    stage_tables_and_args(blob, stage_bytes, soa0, order, final_out, steps_out, &ctr->queue_head, n,
                          offsets, STREAMED ? avail : nullptr);
    const LutView T = lut_view(F.tab);
    ImageRegs IR{};
    if (IMAGE) IR = image_regs(lds_header().G);
    ImageQueue queue;
    // per-lane tallies fit 32 bits (a lane makes < 2^31 trips); widened when flushed
    unsigned my_steps = 0, my_samples = 0, my_binned = 0;
    unsigned long long my_nonfinite = 0;
    WaveQueue q;
    q.start();
    // per-packet outputs are rare (parity tests, the rows protocol): one wave-uniform flag keeps
    // the bench's loop from looking the pointers up every time a packet ends
    const bool want_out = __builtin_amdgcn_readfirstlane(
        (final_out != nullptr || steps_out != nullptr) ? 1 : 0) != 0;
    const int wave_off = (int)((stage_bytes + 31) & ~31ll) +
                         (threadIdx.x >> 6) * (ROWS ? NXC_WAVE_LDS_BYTES_ROWS : NXC_WAVE_LDS_BYTES);
    const int stage_off = wave_off, imgq_off = wave_off + NXC_WAVE_STAGE_BYTES;
    bool has = false, fresh = false;
    long long id = -1, row0 = 0, nrow = 0;
    double lossfrac = 0.0;
    unsigned my_overrun = 0, my_trips = 0;          // my_trips is wave-uniform: a scalar add per trip
    int k = 0, nbounce = 0;
    const int n_it = n_iter > 0x7fffffffll ? 0x7fffffff : (int)n_iter;
    double s[8], d[8];
    // Per lane: a packet taken from the queue spends its first trip through the loop only being
    // binned (record 0), every later trip advancing one step and being binned again, so that the
    // step and the image code each appear once and run with (nearly) full waves.  A lane that is
    // neither fresh nor free holds a live packet with k < n_iter.
    // Image: every trip locates its samples (rotation, bins, masks) and queues the ones inside
    // the image; whenever 64 are waiting they are weighted and added by a full wave.
#ifdef NXC_STAMPS
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = nxc_stamp();
#endif
    for (;;) {
#ifdef NXC_REFILL_MIN      /* experiment: hand out packets only once this many lanes are free */
        const unsigned long long free_ = __ballot(!has);
        const bool serve = __popcll(free_) >= NXC_REFILL_MIN || free_ == ~0ull;
        const long long got = serve ? q.refill<ROWS != 0, STREAMED>(!has, stage_off, s, &row0, &nrow) : -1;
#else
        const long long got = q.refill<ROWS != 0, STREAMED>(!has, stage_off, s, &row0, &nrow);
#endif
        if (got >= 0) { id = got; k = 0; has = true; fresh = true; nbounce = 0; lossfrac = 0.0; }
        if (__ballot(has) == 0) break;
        my_trips++;
        NXC_STAMP(0);                                  // refill
        int p = -1;
        double rv = 0.0, fw = 0.0;
        bool inframe = false;                          // IMAGE == 2: the sample as save() stores it
        float qx = 0.f, qy = 0.f, qz = 0.f;
        if (has) {
            const double before = s[7];
            if (!fresh) {
                if (NBODY) {
                    const BodyK *Bd = &lds_header().Bd;
                    const double *mp = moon_pos + (long long)k * (2 * Bd->n_moons);
                    rk5_step<false, true, FULL, true>(F, T, s, 0.0, lds_header().W, d, Bd, mp);
                    apply_fate<false, true>(s, edge2, 0ull, nbounce, Bd, mp);
                } else {
                    rk5_step<false, true, FULL>(F, T, s, 0.0, lds_header().W, d);
                    apply_fate<BOUNCE>(s, edge2, (unsigned long long)(first_id + id), nbounce);
                }
#ifdef NXC_PROBE_VMOV      /* what binds the loop? N independent cheap / fp64 instructions per trip */
                {
                    int junk;
#pragma unroll
                    for (int j = 0; j < NXC_PROBE_VMOV; j++) asm volatile("v_mov_b32 %0, 0x12345" : "=v"(junk));
                }
#endif
#ifdef NXC_PROBE_FMA
                {
                    double j0 = s[1], j1 = s[2], j2 = s[3], j3 = s[4];
#pragma unroll
                    for (int j = 0; j < NXC_PROBE_FMA / 4; j++) {
                        asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(j0));
                        asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(j1));
                        asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(j2));
                        asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(j3));
                    }
                }
#endif
                k++; my_steps++;
                if (ROWS) lossfrac = (lossfrac + before) - s[7];
            }
            NXC_STAMP(1);                              // step + fate
            fresh = false;
            const bool live = s[7] > 0.0;
            if (ROWS && live) {
                if (k < nrow) store_record<ROWS>(rec, row0 + k, s, lossfrac, id);
                else my_overrun++;                     // the two passes disagree: reported, never written
            }
            if (IMAGE && live) {
                my_samples++;
                if (IMAGE == 2)
                    inframe = image_frame_test(lds_header().G, IR, s[1], s[2], s[3], s[5], s[7], qx,
                                               qy, qz, my_nonfinite);
                else
                    p = image_locate(lds_header().G, IR, s[1], s[2], s[3], s[5], s[7], rv, fw,
                                     my_nonfinite);
            }
            NXC_STAMP(2);                              // locate
            if (!live || k >= n_it) {
                if (want_out) {
                    const LoopK &L = lds_header().L;
                    if (NXC_GLOBAL_AS double *fo = (NXC_GLOBAL_AS double *)L.final_out) {
                        const long long np = L.n;
#pragma unroll
                        for (int c = 0; c < 8; c++) fo[c * np + id] = s[c];
                    }
                    if (NXC_GLOBAL_AS long long *so = (NXC_GLOBAL_AS long long *)L.steps_out) so[id] = k;
                }
                has = false;
            }
        }
        NXC_STAMP(3);                                  // final-state bookkeeping
        if (IMAGE) {                                  // wave-cooperative: outside `if (has)`
#ifdef NXC_EXPERIMENT_KNOBS
            // timing experiments (tools/gpu_exp_dbg.py): 3 = locate only, 2 = + queue, no weight,
            // 1 = everything but the atomics
            if (IR.dbg == 3) { my_binned += IMAGE == 2 ? inframe : p >= 0; continue; }
#endif
            if (IMAGE == 2) queue.push_sample(inframe, qx, qy, qz, (float)s[5], (float)s[7], imgq_off);
            else queue.push(p >= 0, p, rv, fw, imgq_off);
            NXC_STAMP(4);                              // push
            if (queue.waiting() >= 64) {
                double w = 0.0;
                bool ok = image_pop<IMAGE>(queue, imgq_off, IR, p, rv, fw, my_nonfinite);
#ifdef NXC_EXPERIMENT_KNOBS
                if (IR.dbg == 2) { my_binned += ok; continue; }
#endif
                if (ok && !image_weight(lds_header().G, IR, rv, fw, w)) { my_nonfinite++; ok = false; }
                my_binned += ok;
                NXC_STAMP(5);                          // pop + weight
#ifdef NXC_EXPERIMENT_KNOBS
                if (IR.dbg == 1) { my_nonfinite += (w == 12345.678); continue; }
#endif
                image_add_pairs(ok, p, w, acc2);
                NXC_STAMP(6);                          // atomics
            }
        }
    }
#ifdef NXC_STAMPS
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 8; c++) atomicAdd(&ctr->stamp[c], seg[c]);
#endif
    if (IMAGE) {
        while (queue.waiting() > 0) {
            int p = -1;
            double rv = 0.0, fw = 0.0, w = 0.0;
            bool ok = image_pop<IMAGE>(queue, imgq_off, IR, p, rv, fw, my_nonfinite);
            if (ok && !image_weight(lds_header().G, IR, rv, fw, w)) { my_nonfinite++; ok = false; }
            my_binned += ok;
            image_add_pairs(ok, p, w, acc2);
        }
    }
    flush_counter(&ctr->particle_steps, my_steps);
    if ((threadIdx.x & 63) == 0) atomicAdd(&ctr->wave_trips, (unsigned long long)my_trips);
    if (ROWS) flush_counter(&ctr->unfinished, my_overrun);
    if (STREAMED && q.stalled && (threadIdx.x & 63) == 0) atomicAdd(&ctr->unfinished, 1ull);
    if (IMAGE) {
        flush_counter(&ctr->samples, my_samples);
        flush_counter(&ctr->samples_binned, my_binned);
        flush_counter(&ctr->nonfinite, my_nonfinite);
    }
}

// ---------------------------------------------------------------------------------------------
// Adaptive-step driver (Output.py:221-359), one rk5 attempt per loop trip, lane-refill as above.
//   h = min(t_remaining, stored step)                                   :253
//   errmax = max_c delta_c / (res_c + |y_c| res_c), res_v = 0.1 res     :235-238,271-281
//   frac-increase guard -> errmax = 1.1                                  :291
//   errmax < 1e-7 -> errmax = 1 and h*10, which lands in the REJECT branch (errmax >= 1)  :294-300
//   accept (errmax < 1): fate tests on r^2; the grown step is never stored  :302-327
//   reject: stored step = max(0.95 h errmax^-0.25, 0.1 h)               :333-342
#ifdef NXC_VAR_TRACE          /* experiment (tools/gpu_exp_var_trace.py): per-wave times of k_var, 10 ns ticks */
__device__ unsigned long long g_var_trace[8 * 4096];
#endif
// FAIR SHARES AND A MERGED TAIL (the 768-thread form; NXC_VAR_FAIR).  A SIMD issues for its oldest
// wave first, so of the three waves of a SIMD one makes a trip every 4.5 us and another every
// 13.6 (tools/gpu_exp_var_trace.py) -- and the launch ends with the long chains that sat in a slow
// wave while the queue lasted.  Two measures: (1) s_setprio, rotated by the clock (every wave of a
// SIMD leads a third of the time, slices of 41 us): all waves at 6.6-7.3 us per trip; (2) once the
// queue is drained the waves only lose lanes, and three sparse waves share the SIMD's issue slots:
// the first wave of each SIMD to register is its KEEPER, every other wave a DONOR that -- queue
// drained, at most NXC_VAR_MERGE_MAX lanes live -- reserves that many lanes of a keeper (`room`, one
// LDS atomic, undone if it overdraws), writes its packets {state, stored step, attempts, id}, bit
// for bit, into its own LDS block, publishes {count, keeper} and ends; the keeper adopts them on
// its next trip.  A keeper leaves when every wave of the workgroup has published (a wave that
// ends on its own publishes 0) and nothing addressed to it is left.  Tallies stay with the lane
// that made the attempt: totals do not move.
#ifndef NXC_VAR_FAIR
#define NXC_VAR_FAIR 1
#endif
#ifndef NXC_VAR_MERGE_MAX_N
#define NXC_VAR_MERGE_MAX_N 48
#endif
#ifndef NXC_VAR_PRIO_SHIFT
#define NXC_VAR_PRIO_SHIFT 12          /* slices of 2^12 x 10 ns */
#endif
constexpr int NXC_VAR_MERGE_MAX = NXC_VAR_MERGE_MAX_N;      // x 11 doubles, from the start of the wave's block
constexpr int NXC_VAR_HDR_OFF = NXC_WAVE_LDS_BYTES - 64;    // the block's last 64 bytes (image queue: unused here)
constexpr unsigned NXC_VAR_PENDING = 0xffffffffu;
static_assert(NXC_VAR_MERGE_MAX * 11 * 8 <= NXC_VAR_HDR_OFF && NXC_VAR_HDR_OFF >= NXC_WAVE_STAGE_BYTES,
              "a donor's packets and the header share the wave's block with the staging area");
struct VarMerge {
    int wave0;             // LDS byte offset of wave 0's block
    NXC_DEV NXC_LDS_AS double *seg(int w) const
    {
        return (NXC_LDS_AS double *)(unsigned)(wave0 + w * NXC_WAVE_LDS_BYTES);
    }
    // wave w: [0] published count, [1] keeper addressed, [2] adopted; wave 0 also: [4..7] room, [8..11] registrations
    NXC_DEV NXC_LDS_AS unsigned *hdr(int w) const
    {
        return (NXC_LDS_AS unsigned *)(unsigned)(wave0 + w * NXC_WAVE_LDS_BYTES + NXC_VAR_HDR_OFF);
    }
    NXC_DEV NXC_LDS_AS int *room(int k) const { return (NXC_LDS_AS int *)(hdr(0) + 4 + k); }
    NXC_DEV static unsigned ld(NXC_LDS_AS unsigned *a)
    {
        return __hip_atomic_load(a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    NXC_DEV void publish(int w, unsigned count) const
    {
        __hip_atomic_store(hdr(w), count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// Why: a SIMD issues for its OLDEST wave first (tools/gpu_exp_var_trace.py, 1e6 packets: 4.5 / 7.0 /
// 13.6 us per trip for the three waves of a SIMD; 3.4 for a wave alone).  The queue is empty after
// 11-12 ms; the launch then ends with the 3 000-5 000-attempt chains, and those that sat in a slow
// wave have made a third of the trips they would have made in a fast one.  The order of the queue
// cannot put them first: longest first would end at 20-22 ms (tools/gpu_exp_var_order.py), but
// the reference's controller only ever shrinks the stored step (Output.py:333-342), so a packet's
// attempts follow from whether it survives its first returns to the surface -- 10 % of the long
// chains misplaced and the gain is gone (same tool).  Plain (FAIR_ = false):
// the highest throughput, for launches with dozens of packets per lane (1e7: 151 ms against 156).
// Fair: 1e6 packets 33.0 -> 28.8-29.6 ms, 2e6 45.8 -> 42.8, 4e6 72.1 -> 69.5.
template <bool FULL, bool FAIR_>     // FULL: gravity + radiation pressure + photo-loss known at compile time
__global__ void __launch_bounds__(NXC_BLOCK_PERSIST)
k_var(ForceK F, const unsigned char *__restrict__ blob, int64_t stage_bytes, int64_t n,
      const double *__restrict__ soa0, const unsigned *__restrict__ order, double resolution, double outeredge, long long max_steps,
      double *__restrict__ final_out, double *__restrict__ hstore_out,
      DevCounters *__restrict__ ctr)
{
    constexpr bool FAIR = FAIR_ && NXC_VAR_FAIR;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;                          // the host launches fewer waves for few packets
    const VarMerge M{(int)((stage_bytes + 31) & ~31ll)};
    if (FAIR && lane == 0) {                                 // before the staging barrier
        M.hdr(wave)[0] = NXC_VAR_PENDING; M.hdr(wave)[1] = 0u; M.hdr(wave)[2] = 0u;
        if (wave == 0)
            for (int k = 4; k < 12; k++) M.hdr(0)[k] = 0u;
    }
    stage_tables_and_args(blob, stage_bytes, soa0, order, final_out, nullptr, &ctr->queue_head, n);
    bool keeper = false, published = false, settled = false;
    int prev_live = -1, prio = 0;
    // HW_REG_HW_ID (4): WAVE_ID = bits 3:0 (the wave's slot on its SIMD), SIMD_ID = bits 5:4 -- read
    // where needed (a wave stays where it is; two registers less to carry through the step)
    auto hw_simd = []() { return (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); };
    auto hw_slot = []() { return (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4); };
    if (FAIR) {
        const int simd = hw_simd();
        unsigned r = 0;
        if (lane == 0) r = __hip_atomic_fetch_add(M.hdr(0) + 8 + simd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        keeper = __builtin_amdgcn_readfirstlane(r) == 0;
        if (keeper && lane == 0) M.publish(wave, 0u);
    }
    const LutView T = lut_view(F.tab);
    const double resx = resolution, resv = 0.1 * resolution, resf = resolution;
    unsigned long long my_steps = 0, my_nonfinite = 0, my_bad = 0, my_neg = 0, my_unfinished = 0;
    WaveQueue q;
    q.start();
    const int stage_off = M.wave0 + wave * NXC_WAVE_LDS_BYTES;
    bool has = false;
    long long id = -1, it = 0;
    unsigned my_trips = 0;
    double s[8], hs = 1000.0;
#ifdef NXC_VAR_TRACE
    const unsigned long long tr_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long tr_drain = 0, tr_low = 0;
    unsigned tr_trips_drain = 0, tr_trips_low = 0;
#endif
    for (;;) {
        const long long got = q.refill(!has, stage_off, s);
#ifdef NXC_VAR_TRACE
        if (q.drained && tr_drain == 0) { tr_drain = __builtin_amdgcn_s_memrealtime(); tr_trips_drain = my_trips; }
        if (q.drained && tr_low == 0 && __popcll(__ballot(has)) <= 8) { tr_low = __builtin_amdgcn_s_memrealtime(); tr_trips_low = my_trips; }
#endif
        if (got >= 0) {
            id = got; it = 0; hs = 1000.0; has = true;
        }
        if (FAIR && q.drained && !settled) {
            const int simd = hw_simd();
            const unsigned long long livem = __ballot(has);
            int live = __popcll(livem);
            if (!keeper) {
                if (live > 0 && live <= NXC_VAR_MERGE_MAX) {
                    int r = 0;
                    if (lane < 4) r = __hip_atomic_load(M.room(lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned long long can = __ballot(lane < 4 && r >= live);
                    if (can != 0) {
                        // the keeper of the own SIMD if it can take them (that SIMD then loses a wave), else the first that can
                        const int k = (can >> simd) & 1ull ? simd : (int)__builtin_ctzll(can);
                        int old = 0;
                        if (lane == 0) {
                            old = __hip_atomic_fetch_sub(M.room(k), live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (old < live) __hip_atomic_fetch_add(M.room(k), live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (__builtin_amdgcn_readfirstlane(old) >= live) {
                            if (has) {
                                NXC_LDS_AS double *e = M.seg(wave) + __popcll(livem & ((1ull << lane) - 1ull));
#pragma unroll
                                for (int c = 0; c < 8; c++) e[c * NXC_VAR_MERGE_MAX] = s[c];
                                e[8 * NXC_VAR_MERGE_MAX] = hs;
                                e[9 * NXC_VAR_MERGE_MAX] = __longlong_as_double(it);
                                e[10 * NXC_VAR_MERGE_MAX] = __longlong_as_double(id);
                                has = false;
                            }
                            if (lane == 0) { M.hdr(wave)[1] = (unsigned)k; M.publish(wave, (unsigned)live); }
                            published = true;
                        }
                    }
                }
            } else {
                // room: what the lanes freed since the last look (all free lanes at the first)
                const int freed = prev_live < 0 ? 64 - live : prev_live - live;
                if (freed > 0 && lane == 0)
                    __hip_atomic_fetch_add(M.room(simd), freed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // adopt what is addressed to this keeper: lane w looks at wave w's header
                bool mine = false;
                unsigned c = 0;
                if (lane < nw) {
                    c = M.ld(M.hdr(lane));
                    mine = c != NXC_VAR_PENDING && c != 0u && M.hdr(lane)[1] == (unsigned)simd && M.hdr(lane)[2] == 0u;
                }
                unsigned long long offer = __ballot(mine);
                // every wave has published and nothing is addressed to this keeper: it is on its
                // own from here (no more looks, no more turns at the priority)
                if (offer == 0 && __ballot(lane < nw && c == NXC_VAR_PENDING) == 0) {
                    settled = true;
                    __builtin_amdgcn_s_setprio(0);
                }
                while (offer != 0) {
                    const int w = __builtin_ctzll(offer);
                    offer &= offer - 1;
                    const int got = __shfl((int)c, w, 64);       // reserved: fits
                    const unsigned long long freem = __ballot(!has);
                    const int rank = __popcll(freem & ((1ull << lane) - 1ull));
                    if (!has && rank < got) {
                        const NXC_LDS_AS double *e = M.seg(w) + rank;
#pragma unroll
                        for (int c2 = 0; c2 < 8; c2++) s[c2] = e[c2 * NXC_VAR_MERGE_MAX];
                        hs = e[8 * NXC_VAR_MERGE_MAX];
                        it = __double_as_longlong(e[9 * NXC_VAR_MERGE_MAX]);
                        id = __double_as_longlong(e[10 * NXC_VAR_MERGE_MAX]);
                        has = true;
                    }
                    if (lane == 0) M.hdr(w)[2] = 1u;
                    live += got;
                }
                prev_live = live;
            }
        }
        if (__ballot(has) == 0) {
            if (!FAIR || !keeper || settled) break;
            // every wave has published, and nothing published to this keeper is left
            const int simd = hw_simd();
            bool open = false;
            if (lane < nw) {
                const unsigned c = M.ld(M.hdr(lane));
                open = c == NXC_VAR_PENDING || (c != 0u && M.hdr(lane)[1] == (unsigned)simd && M.hdr(lane)[2] == 0u);
            }
            if (__ballot(open) == 0) break;
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        my_trips++;
        if (FAIR && !settled && (my_trips & 3u) == 0) {
            // every wave of the SIMD leads for a slice of the clock in turn (looked up every fourth
            // trip: a slice lasts six)
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            const int pr = (int)(((t >> NXC_VAR_PRIO_SHIFT) + (unsigned long long)hw_slot()) % 3ull);
            if (pr != prio) {
                prio = pr;
                if (pr == 2) __builtin_amdgcn_s_setprio(2);
                else if (pr == 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
        }
        if (has) {
            bool done = !(s[0] > resolution && s[7] > 0.0);
            if (!done && it >= max_steps) { my_unfinished++; done = true; }
            if (!done) {
                const double h = __builtin_fmin(s[0], hs);
                if (!(h > 0.0)) { my_bad++; done = true; }
                else {
                    double t[8], d[8];
#pragma unroll
                    for (int c = 0; c < 8; c++) t[c] = s[c];
                    rk5_step<true, false, FULL>(F, T, t, h, StepW{}, d);
                    my_steps++; it++;
                    const double fscale = resf + __builtin_fabs(t[7]) * resf;
                    // max over the columns; a NaN quotient must not be dropped by fmax (the
                    // reference asserts on non-finite errmax, Output.py:284 -- and would spin
                    // forever on a row whose max() skipped the NaN)
                    double e = d[0];
                    bool finite = true;
#pragma unroll
                    for (int c = 1; c <= 7; c++) {
                        const double scale = c <= 3 ? resx + __builtin_fabs(t[c]) * resx
                                           : c <= 6 ? resv + __builtin_fabs(t[c]) * resv : fscale;
                        const double q = nxc_div(d[c], scale);
                        finite = finite && (__builtin_fabs(q) <= 1.7976931348623157e308);
                        e = __builtin_fmax(e, q);
                    }
                    if (!finite) e = __builtin_nan("");
                    if (!(__builtin_fabs(e) <= 1.7976931348623157e308)) { my_nonfinite++; done = true; }
                    else {
                        if (t[7] < 0.0 && e < 1.0) my_neg++;
                        if ((t[7] - s[7] > fscale) && (e > 1.0)) e = 1.1;
                        double hold = h;
                        if (e < 1e-7) { e = 1.0; hold = h * 10; }
                        if (e < 1.0) {
                            { int nb_ = 0; apply_fate<false>(t, outeredge, 0ull, nb_); }
#pragma unroll
                            for (int c = 0; c < 8; c++) s[c] = t[c];
                        } else {
                            const double hn = 0.95 * hold * nxc_pow_m025(e);
                            if (!(__builtin_fabs(hn) <= 1.7976931348623157e308)) { my_bad++; done = true; }
                            else hs = __builtin_fmax(hn, 0.1 * hold);
                        }
                    }
                }
            }
            if (done) {
#pragma unroll
                for (int c = 0; c < 8; c++) final_out[c * n + id] = s[c];
#ifdef NXC_VAR_TRACE
                if (hstore_out) hstore_out[id] = (double)it;   // experiment: the packet's attempts
#else
                if (hstore_out) hstore_out[id] = hs;
#endif
                has = false;
            }
        }
    }
    if (FAIR && !keeper && !published && lane == 0) M.publish(wave, 0u);
#ifdef NXC_VAR_TRACE
    if ((threadIdx.x & 63) == 0) {
        // HW_REG_HW_ID (4): SIMD_ID = bits 5:4, WAVE_ID = bits 3:0
        const unsigned hw = __builtin_amdgcn_s_getreg((5 << 11) | (0 << 6) | 4);
        unsigned long long *tr = g_var_trace + 8 * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 4095);
        tr[0] = tr_start; tr[1] = tr_drain; tr[2] = tr_low; tr[3] = __builtin_amdgcn_s_memrealtime();
        tr[4] = my_trips; tr[5] = tr_trips_drain; tr[6] = tr_trips_low; tr[7] = hw;
    }
#endif
    flush_counter(&ctr->particle_steps, my_steps);
    if ((threadIdx.x & 63) == 0) atomicAdd(&ctr->wave_trips, (unsigned long long)my_trips);
    flush_counter(&ctr->nonfinite, my_nonfinite);
    flush_counter(&ctr->bad_step, my_bad);
    flush_counter(&ctr->neg_frac, my_neg);
    flush_counter(&ctr->unfinished, my_unfinished);
}

// ---------------------------------------------------------------------------------------------
// T = double, or float for samples handed over as the reference stores them (Output.py:528-543);
// widening a float is exact, so both give the same image.
constexpr int NXC_IMAGE_BLOCK = 1024;
template <typename T>
__global__ void __launch_bounds__(NXC_IMAGE_BLOCK)
k_image(const unsigned char *__restrict__ blob, int64_t stage_bytes, int64_t p,
        const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ z,
        const T *__restrict__ vy, const T *__restrict__ frac,
        double *__restrict__ acc2, DevCounters *__restrict__ ctr)
{
    stage_tables(blob, stage_bytes);
    const ImageRegs IR = image_regs(lds_header().G);
    unsigned long long my_samples = 0, my_binned = 0, my_nonfinite = 0;
    // wave-uniform trip count (the accumulation is wave-cooperative); the last trip is ragged
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < p;
         base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const bool has = i < p;
        my_samples += has;
        double sx = 0, sy = 0, sz = 0, svy = 0, sf = 0;
        if (has) { sx = x[i]; sy = y[i]; sz = z[i]; svy = vy[i]; sf = frac[i]; }
        image_sample(lds_header().G, IR, has, sx, sy, sz, svy, sf, acc2, my_binned, my_nonfinite);
    }
    // (the tables are not read any more: their first bytes carry the workgroup's sums)
    flush_counters_wg(reinterpret_cast<unsigned long long *>(nxc_lds), &ctr->samples, my_samples,
                      &ctr->samples_binned, my_binned, &ctr->nonfinite, my_nonfinite);
}

// ---------------------------------------------------------------------------------------------
// The same image through LDS-privatised tiles (the histogram design BASELINE.json names), for
// sample sets large enough to pay for two launches.  k_image is bound by the chip's memory-side
// atomic request rate (2.1-2.4e10/s: one request per binned sample, whatever the pixel), a fifth of
// what HBM could stream; the samples of a cloud have no pixel locality a wave could merge on, so
// locality is MADE: pass 1 (k_image_bin) locates and weighs every sample as k_image does and
// appends the binned ones -- {weight fp64, pixel-in-tile u16} -- to the staging block of the
// sample's TILE in LDS; a full block of 256 leaves for HBM as one chunk (2.5 KB of contiguous
// stores) in the workgroup's own scratch region, so no global atomic is involved.  Pass 2
// (k_image_tiles): a workgroup owns one tile as 8192 x {fp64 sum, u32 count} in LDS, adds the
// chunks tagged with its tile by LDS atomics (ds_add_f64 / ds_add_u32), and hands the tile to the
// resident image with one pair of global atomics per touched pixel.  Tile b holds the image rows
// ix = b (mod nb): every tile sees the same cut through the cloud, so the static assignment of
// tiles to workgroups is balanced.  Packet counts are exact as before; the weight sums differ from
// k_image's by the order of fp64 additions only (both are unordered).
//
// Image size.  Pass 1 keeps one staging block per tile in LDS, so the number of tiles fixes the
// chunk: nb x CAP = 8192 staged entries (80 KB) whatever the image -- 32 tiles of 256-entry chunks
// up to 512^2 pixels (262 144), 64 of 128 up to 524 288, 128 of 64 up to 1024^2 (the reference's
// default 800 x 800, ModelImage.py:53, takes 128 tiles of 7 image rows).  A chunk of 64 still
// leaves as 512 + 128 contiguous bytes.  Larger images stay with k_image.
#ifndef NXC_TILE_STAGE_N          // (overridable: tools/gpu_exp_tile_occupancy.sh)
#define NXC_TILE_STAGE_N 8192
#endif
constexpr int NXC_TILE_STAGE = NXC_TILE_STAGE_N;   // staged entries per workgroup of pass 1 (all tiles)
constexpr int NXC_TILE_MAX = 128;           // tiles per image at most
constexpr int NXC_TILE_PIXELS = 8192;       // pixels per tile at most (96 KB of LDS)
#ifndef NXC_TILE_UNROLL_N
#define NXC_TILE_UNROLL_N 4
#endif
// samples per thread between two workgroup barriers: 4 float32 samples (1.42 / 1.26 / 1.20 / 1.19 ms
// per 1.34e8 rows for 1 / 2 / 3 / 4), 2 of 64 bits (4 would spill)
template <typename T> constexpr int nxc_tile_unroll() { return sizeof(T) == 4 ? NXC_TILE_UNROLL_N : 2; }

// Workgroup barrier that orders LDS traffic only: the global loads of the next samples stay in
// flight across it (__syncthreads waits for vmcnt(0) as well).
NXC_DEV void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// image_add_pairs for a pixel that carries a whole count (pass 2's hand-over).
NXC_DEV void image_add_pairs_n(bool has, int pix, double w, double cnt, double *__restrict__ acc2)
{
    if (__ballot(has) == 0) return;
    const bool upper = (threadIdx.x & 32) != 0;
    const int ppix = half_swap(has ? pix : -1, upper);
    const auto pc = __builtin_amdgcn_permlane32_swap(__double2loint(cnt), __double2loint(cnt), false, false);
    const auto ph = __builtin_amdgcn_permlane32_swap(__double2hiint(cnt), __double2hiint(cnt), false, false);
    const double pcnt = __hiloint2double(upper ? (int)ph[0] : (int)ph[1], upper ? (int)pc[0] : (int)pc[1]);
    const bool own = has && w != 0.0;
    const bool partner = ppix >= 0;
    {
        const bool act = upper ? partner : own;
        if (act)
            unsafeAtomicAdd(&acc2[2ll * (upper ? ppix : pix) + (upper ? 1 : 0)], upper ? pcnt : w);
    }
    {
        const bool act = upper ? own : partner;
        if (act)
            unsafeAtomicAdd(&acc2[2ll * (upper ? pix : ppix) + (upper ? 0 : 1)], upper ? w : pcnt);
    }
}

// Pass 1.  Workgroup k takes the samples [k span, (k + 1) span) and owns the chunks
// [k mc, (k + 1) mc) of the scratch arrays: sw / sl hold a chunk's 8-byte payloads / pixels-in-tile,
// tag[c] = tile << 8 | (entries - 1), nchunks[k] the chunks it wrote (at most span / 256 full ones
// + one partial per tile = mc).
// DEFER: the payload is {vy, masked frac} as two floats and pass 2 forms the weight -- on full
// waves of binned samples, where this pass would run the g-value lookups and divisions with the
// half of its lanes whose sample fell inside the image; possible whenever the samples are float32
// values (stored rows, or the down-cast image), since the masked fraction is frac or a zero.
// Otherwise the payload is the fp64 weight itself.
#ifndef NXC_TILE_BIN_BLOCK_N
#define NXC_TILE_BIN_BLOCK_N 1024
#endif
constexpr int NXC_TILE_BIN_BLOCK = NXC_TILE_BIN_BLOCK_N;
template <typename T, bool DEFER, int CAP>
__global__ void __launch_bounds__(NXC_TILE_BIN_BLOCK)
k_image_bin(const unsigned char *__restrict__ blob, int64_t stage_bytes, int64_t p, int64_t span,
            int mc, int nb_log2, const T *__restrict__ x, const T *__restrict__ y,
            const T *__restrict__ z, const T *__restrict__ vy, const T *__restrict__ frac,
            double *__restrict__ sw, unsigned short *__restrict__ sl,
            unsigned short *__restrict__ list, unsigned *__restrict__ nlist,
            DevCounters *__restrict__ ctr)
{
    constexpr int U = nxc_tile_unroll<T>();
    stage_tables(blob, stage_bytes);
    const ImageRegs IR = image_regs(lds_header().G);
    const int nb = 1 << nb_log2;
    const int s0 = ((int)stage_bytes + 15) & ~15;
    double *const stw = reinterpret_cast<double *>(nxc_lds + s0);                        // [nb][CAP]
    unsigned short *const stl = reinterpret_cast<unsigned short *>(nxc_lds + s0 + nb * CAP * 8);
    unsigned *const cnt = reinterpret_cast<unsigned *>(nxc_lds + s0 + nb * CAP * 10);    // [nb]
    unsigned *const nch = cnt + nb;                   // chunks written so far
    unsigned *const again = nch + 1;                  // [2]: somebody still holds an entry
    unsigned *const bc = again + 2;                   // [nb]: chunks written per tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nwaves = blockDim.x >> 6;
    if (tid < 2 * nb + 3) cnt[tid] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * span;
    const int64_t hi = lo + span < p ? lo + span : p;
    const size_t chunk0 = (size_t)blockIdx.x * (size_t)mc;

    // a wave moves staging block b (n entries) to the workgroup's next chunk and enters the chunk
    // in tile b's list (one wave serves a tile at any time: bc[b] needs no atomic)
    unsigned short *const my_list = list + ((size_t)blockIdx.x * nb) * (size_t)mc;
    auto flush = [&](int b, int n) {
        unsigned j = 0;
        if (lane == 0) {
            j = atomicAdd(nch, 1u);
            const unsigned i = bc[b];
            bc[b] = i + 1;
            my_list[(size_t)b * mc + i] = (unsigned short)j;
        }
        j = (unsigned)__builtin_amdgcn_readfirstlane((int)j);
        const size_t c = (chunk0 + j) * CAP;
#pragma unroll
        for (int t = 0; t < (CAP + 63) / 64; t++) {
            const int e = lane + 64 * t;
            if (e < n) { sw[c + e] = stw[b * CAP + e]; sl[c + e] = stl[b * CAP + e]; }
        }
    };

    unsigned long long my_samples = 0, my_binned = 0, my_nonfinite = 0;
    unsigned round = 0;
    // the samples of the first trip; every trip loads the next one's before it works on its own
    T nx_[U][5];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t i = lo + (int64_t)u * blockDim.x + tid;
#pragma unroll
        for (int c = 0; c < 5; c++) nx_[u][c] = T(0);
        if (i < hi) { nx_[u][0] = x[i]; nx_[u][1] = y[i]; nx_[u][2] = z[i]; nx_[u][3] = vy[i]; nx_[u][4] = frac[i]; }
    }
    for (int64_t base = lo; base < hi; base += (int64_t)U * blockDim.x) {    // block-uniform trips
        T cur[U][5];
        bool has[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            has[u] = base + (int64_t)u * blockDim.x + tid < hi;
#pragma unroll
            for (int c = 0; c < 5; c++) cur[u][c] = nx_[u][c];
            const int64_t i = base + (int64_t)(U + u) * blockDim.x + tid;
            if (i < hi) { nx_[u][0] = x[i]; nx_[u][1] = y[i]; nx_[u][2] = z[i]; nx_[u][3] = vy[i]; nx_[u][4] = frac[i]; }
        }
        int bk[U], loc[U];
        double w[U];
        unsigned pend = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            my_samples += has[u];
            int ix = 0, iz = 0;
            double radvel = 0.0, fw = 0.0;
            double s[5];
#pragma unroll
            for (int c = 0; c < 5; c++) s[c] = (double)cur[u][c];
            if (sizeof(T) == 8 && IR.downcast) {        // a widened float is its own round trip
#pragma unroll
                for (int c = 0; c < 5; c++) s[c] = f32_round_trip(s[c]);
            }
            bool ok = has[u] && image_locate_core_xz(lds_header().G, IR, s[0], s[1], s[2], s[3], s[4],
                                                     radvel, fw, my_nonfinite, ix, iz);
            if (DEFER) {
                w[u] = __hiloint2double(__float_as_int((float)fw), __float_as_int((float)s[3]));
            } else {
                w[u] = 0.0;
                if (ok && !image_weight(lds_header().G, IR, radvel, fw, w[u])) { my_nonfinite++; ok = false; }
                my_binned += ok;
            }
            bk[u] = ix & (nb - 1);
            loc[u] = (ix >> nb_log2) * IR.nz + iz;
            pend |= ok ? 1u << u : 0u;
        }
        for (;;) {
#pragma unroll
            for (int u = 0; u < U; u++)
                if (pend >> u & 1u) {
                    const unsigned pos = atomicAdd(&cnt[bk[u]], 1u);
                    if (pos < (unsigned)CAP) {
                        stw[bk[u] * CAP + pos] = w[u];
                        stl[bk[u] * CAP + pos] = (unsigned short)loc[u];
                        pend &= ~(1u << u);
                    }
                }
            lds_barrier();
            if (tid == 0) again[(round + 1) & 1] = 0;
            for (int b = wid; b < nb; b += nwaves)
                if (cnt[b] >= (unsigned)CAP) {                     // full (later arrivals try again)
                    flush(b, CAP);
                    if (lane == 0) cnt[b] = 0;
                }
            if (__ballot(pend != 0) != 0 && lane == 0) again[round & 1] = 1;
            lds_barrier();
            const bool more = again[round & 1] != 0;
            round++;
            if (!more) break;
        }
    }
    // the partial blocks; nlist = chunks of the tile << 16 | entries of the last one
    for (int b = wid; b < nb; b += nwaves) {
        const int n = (int)cnt[b];
        if (n > 0) flush(b, n);
        if (lane == 0) nlist[(size_t)blockIdx.x * nb + b] = bc[b] << 16 | (unsigned)(n > 0 ? n : CAP);
    }
    // (the staging blocks are empty now: their first bytes carry the workgroup's sums)
    flush_counters_wg(reinterpret_cast<unsigned long long *>(stw), &ctr->samples, my_samples,
                      &ctr->samples_binned, my_binned, &ctr->nonfinite, my_nonfinite);
}

// Pass 2.  Workgroup (tile b, group g) adds the chunks of tile b written by the producers
// k = g (mod ng) into its LDS tile and hands the touched pixels to the resident image.  WEIGH: the
// chunks hold {vy, masked frac} (k_image_bin<DEFER>) and the weight is formed here, so the image
// tables are staged in front of the tile.
template <bool WEIGH, int CAP>
__global__ void __launch_bounds__(NXC_IMAGE_BLOCK)
k_image_tiles(const unsigned char *__restrict__ blob, int64_t stage_bytes, int n_prod, int mc,
              int nb_log2, int ng, int tile_used, int nz,
              const double *__restrict__ sw, const unsigned short *__restrict__ sl,
              const unsigned short *__restrict__ list, const unsigned *__restrict__ nlist,
              double *__restrict__ acc2, DevCounters *__restrict__ ctr)
{
    constexpr int E = (CAP + 63) / 64, NF = 512 / CAP;      // 512 entries in flight per wave
    int t0 = 0;
    ImageRegs IR = {};
    if (WEIGH) {
        stage_tables(blob, stage_bytes);
        IR = image_regs(lds_header().G);
        t0 = ((int)stage_bytes + 15) & ~15;
    }
    double *const tw = reinterpret_cast<double *>(nxc_lds + t0);
    unsigned *const tc = reinterpret_cast<unsigned *>(nxc_lds + t0 + 8 * NXC_TILE_PIXELS);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nwaves = blockDim.x >> 6;
    const int nb = 1 << nb_log2;
    const int b = blockIdx.x & (nb - 1), g = blockIdx.x >> nb_log2;
    for (int i = tid; i < tile_used; i += blockDim.x) { tw[i] = 0.0; tc[i] = 0u; }
    __syncthreads();
    unsigned long long my_binned = 0, my_nonfinite = 0;
    // one chunk: E entries per lane
    auto fetch = [&](size_t c, int n, double (&pay)[E], int (&loc)[E]) {
#pragma unroll
        for (int t = 0; t < E; t++) {
            const int e = lane + 64 * t;
            pay[t] = 0.0; loc[t] = 0;
            if (e < n) { pay[t] = sw[c + e]; loc[t] = sl[c + e]; }
        }
    };
    auto add = [&](int n, const double (&pay)[E], const int (&loc)[E]) {
#pragma unroll
        for (int t = 0; t < E; t++) {
            if (lane + 64 * t < n) {
                double w = pay[t];
                bool ok = true;
                if (WEIGH) {
                    const double vy = (double)__int_as_float(__double2loint(pay[t]));
                    const double fw = (double)__int_as_float(__double2hiint(pay[t]));
                    ok = image_weight(lds_header().G, IR, vy + IR.vrplanet, fw, w);
                    my_binned += ok;
                    my_nonfinite += !ok;
                }
                if (ok) {
                    if (w != 0.0) unsafeAtomicAdd(&tw[loc[t]], w);
                    atomicAdd(&tc[loc[t]], 1u);
                }
            }
        }
    };
    // a wave takes whole producers; their lists name the chunks of this tile, so nothing is
    // searched, and two chunks are in flight per wave
    for (int k = g + wid * ng; k < n_prod; k += nwaves * ng) {
        const unsigned nl = nlist[(size_t)k * nb + b];
        const int n = (int)(nl >> 16), last = (int)(nl & 0xffffu);
        const unsigned short *const L = list + ((size_t)k * nb + b) * (size_t)mc;
        const size_t c0 = (size_t)k * (size_t)mc;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int id = i0 + lane < n ? (int)L[i0 + lane] : 0;
            const int m = n - i0 < 64 ? n - i0 : 64;
            for (int l = 0; l < m; l += NF) {              // NF chunks (512 entries) in flight
                double pay[NF][E];
                int loc[NF][E], cn[NF];
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    const bool there = l + f < m;
                    const int j = __builtin_amdgcn_readlane(id, there ? l + f : l);
                    cn[f] = there ? (i0 + l + f == n - 1 ? last : CAP) : 0;
                    fetch((c0 + j) * CAP, cn[f], pay[f], loc[f]);
                }
#pragma unroll
                for (int f = 0; f < NF; f++) add(cn[f], pay[f], loc[f]);
            }
        }
    }
    __syncthreads();
    for (int i0 = 0; i0 < tile_used; i0 += blockDim.x) {           // block-uniform: pairs cooperate
        const int i = i0 + tid;
        const bool has = i < tile_used && tc[i] > 0u;
        const int lrow = i / nz, iz = i - lrow * nz;
        const int pix = ((lrow << nb_log2) + b) * nz + iz;
        image_add_pairs_n(has, pix, has ? tw[i] : 0.0, has ? (double)tc[i] : 0.0, acc2);
    }
    if (WEIGH)      // (the tile has been handed over: its first bytes carry the workgroup's sums)
        flush_counters_wg(reinterpret_cast<unsigned long long *>(tw), &ctr->samples_binned, my_binned,
                          &ctr->nonfinite, my_nonfinite, &ctr->nonfinite, 0ull);
}

// ---- measurement helpers (bench.py's roofline object) --------------------------------------------
// Streaming copy, 16 bytes per lane: the box's own HBM ceiling next to the 8 TB/s of the data sheet.
// One workgroup copies four consecutive 4 KB runs, all four loads of a lane in flight before its
// first store, non-temporal both ways: 6.25 TB/s on the boxes of this pool, which is the 6.29 of
// MI355X_MICROARCH.md's float4 copy (tools/ubench_copy.hip: a persistent grid-stride loop -- what
// this kernel was until round 4 -- stops at 4.5-5.0, hipMemcpyAsync at 4.8).
constexpr int NXC_COPY_UNROLL = 4;
__global__ void __launch_bounds__(NXC_BLOCK)
k_stream_copy(const nxc_v2d *__restrict__ src, nxc_v2d *__restrict__ dst, int64_t n16)
{
    const int64_t base = (int64_t)blockIdx.x * (NXC_BLOCK * NXC_COPY_UNROLL) + threadIdx.x;
    nxc_v2d r[NXC_COPY_UNROLL];
#pragma unroll
    for (int u = 0; u < NXC_COPY_UNROLL; u++) {
        const int64_t i = base + (int64_t)u * NXC_BLOCK;
        if (i < n16) r[u] = __builtin_nontemporal_load(src + i);
    }
#pragma unroll
    for (int u = 0; u < NXC_COPY_UNROLL; u++) {
        const int64_t i = base + (int64_t)u * NXC_BLOCK;
        if (i < n16) __builtin_nontemporal_store(r[u], dst + i);
    }
}

// Shader clock under fp64 load: every wave runs `iters` rounds of eight independent fp64 fma
// chains between two pairs of stamps; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
// (MI355X_MICROARCH.md, DVFS item 6).  Diagnostic launch of its own: no product kernel carries stamps.
__global__ void __launch_bounds__(NXC_BLOCK_PERSIST)
k_clock(unsigned long long *__restrict__ out, int iters, double seed)
{
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = seed + threadIdx.x + j;
    const double m = 0.999999, c = 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = __builtin_fma(a[j], m, c);
    }
    double sink = 0.0;
#pragma unroll
    for (int j = 0; j < 8; j++) sink += a[j];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[3 * w] = t1 - t0;
        out[3 * w + 1] = r1 - r0;
        out[3 * w + 2] = (unsigned long long)__double_as_longlong(sink);   // keeps the chains alive
    }
}

// Fault injection for the bounded wait on a collective (nxc_comm_test_stall): one wave that holds
// the stream for `ticks` of the 100 MHz s_memrealtime counter and then leaves -- it ends by itself
// whether or not anybody still waits for it.
__global__ void __launch_bounds__(64)
k_stall(unsigned long long ticks, unsigned long long *__restrict__ out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long now = t0;
    while (now - t0 < ticks) {
        __builtin_amdgcn_s_sleep(127);
        now = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0 && out) *out = now - t0;
}

__global__ void k_math(const unsigned char *__restrict__ blob, int which, int64_t n,
                       const double *__restrict__ in, const double *__restrict__ in2,
                       double *__restrict__ out)
{
    stage_tables(blob, NXC_HEADER_BYTES);              // nxc_log reads its table from the header
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double v = in[i];
        double r;
        switch (which) {
        case 0: r = nxc_exp(v); break;
        case 1: r = nxc_log(v); break;
        case 2: r = nxc_cube(v); break;
        case 3: r = nxc_sqrt(v); break;
        default: r = nxc_div(v, in2[i]); break;
        }
        out[i] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// Spacecraft line-of-sight cones (data_simulation/compute_iteration.py:98-232): for every
// (stored sample, spectrum) pair decide whether the sample lies in the view cone of half-angle
// dphi in front of the planet cut-off, and add its weight / Apix to that spectrum's radiance.
// The reference narrows the pairs with a KD-tree (balls of radius t_k sin(2 dphi) around the ladder
// points t_k along the line of sight); here all pairs are tested: a spectra tile sits in LDS and
// is broadcast to the lanes, each lane owns samples.  The ball pre-selection is reproduced for
// the (rare) pairs that pass the cone test, so the selected set is the reference's.
struct LosK {
    double sin_dphi, sin_2dphi, cos_thr, cos_thr2_lo, vrplanet, unit_cm2;
    double log1p_s_inv, t0;        // ladder: t_k = t0 (1 + sin_dphi)^k; only to seed the ball search
    double tan_dphi;               // block culling: a cone's radius per unit distance along its axis
    int cull, tile_cap;            // cull = 1: every boresight is a unit vector (checked on the host);
                                   // tile_cap: spectra per LDS tile of k_los (<= NXC_LOS_TILE)
    int n_lines, n_ladder;
    int64_t index_shift;           // subtracted from the index column: packet number inside its Output
    int64_t row_base;              // row number of the first sample of this launch (slabs)
    int64_t tile_off;              // byte offset of the spectra tile inside the LDS block
    LutDesc line[4];
};

constexpr int NXC_LOS_TILE = 512;  // spectra per launch tile at most, all of them in LDS (LosK.tile_cap:
                                   // fewer when large g-value tables leave less room)
constexpr int NXC_LOS_SP = 7;      // doubles per spectrum there: position, boresight, cut-off
#ifndef NXC_LOS_BLOCK_N             // (overridable: tools/ experiments)
#define NXC_LOS_BLOCK_N 8
#endif
constexpr int NXC_LOS_BLOCK = NXC_LOS_BLOCK_N;   // consecutive stored samples tested together first

// The cheap part of the pair test: in front of the spacecraft, this side of the planet cut-off,
// inside a cone a hair wider than the real one.  What passes goes on to los_pair (one pair in
// 1.4e4 for the bench cloud).  A non-finite coordinate fails every comparison, as in the reference.
NXC_DEV bool los_pair_maybe(const LosK &K, const double *__restrict__ sp, double px, double py, double pz)
{
    const double rx = px - sp[0], ry = py - sp[1], rz = pz - sp[2];
    const double q = (rx * sp[3] + ry * sp[4]) + rz * sp[5];
    if (!(q > 0.0) || !(q < sp[6])) return false;
    const double d2 = (rx * rx + ry * ry) + rz * rz;
    return q * q >= K.cos_thr2_lo * d2;
}

// One (stored sample, spectrum) pair, exactly as the reference decides and weighs it
// (compute_iteration.py:177-213).  sp: the spectrum's eight tile values.
// (vy_p, frac_p, idx_p: the sample's radial velocity, fraction and slot in `included`, loaded by the
// caller together with its coordinates; ladder: global or LDS)
NXC_DEV void los_pair(const LosK &K, const double *__restrict__ sp, int64_t spectrum, int64_t p,
                      double px, double py, double pz, double vy_p, double frac_p, long long idx_p,
                      const double *ladder, double *radiance, unsigned long long *npackets,
                      unsigned char *__restrict__ included, long long used_cap,
                      long long *__restrict__ used_pairs, unsigned long long *__restrict__ n_used,
                      double rs_1e6, unsigned long long &my_pairs, unsigned long long &my_nonfinite)
{
    const double xs = sp[0], ys = sp[1], zs = sp[2], bx = sp[3], by = sp[4], bz = sp[5];
    const double rx = px - xs, ry = py - ys, rz = pz - zs;
    const double q = (rx * bx + ry * by) + rz * bz;                // losrad   :178
    if (!(q > 0.0) || !(q < sp[6])) return;                        // cone in front; planet cut :185
    const double d2 = (rx * rx + ry * ry) + rz * rz;
    if (!(q * q >= K.cos_thr2_lo * d2)) return;                    // coarse cone test
#if defined(NXC_LOS_EXPERIMENT) && NXC_LOS_EXPERIMENT >= 1   /* timing experiments: see tools/gpu_exp_los_stages.sh */
    my_pairs++;
    return;
#endif
    const double dist = nxc_sqrt(d2);                              // :177
    double cosang = nxc_div(q, dist);                              // :179
    cosang = cosang > 1.0 ? 1.0 : cosang;                          // :180
    if (!(cosang >= K.cos_thr)) return;                            // ang <= dphi  :181-185
    // KD-tree pre-selection (:164-173): inside any ball |X - (x_sc + bore t_k)| <= t_k sin(2 dphi)
    const int nk = (int)sp[7];
    int kc = (int)(nxc_log(q / K.t0) * K.log1p_s_inv);
    kc = kc < 0 ? 0 : kc;
    bool cand = false;
    for (int k = kc - 3; k <= kc + 3; k++) {
        if (k < 0 || k >= nk) continue;
        const double t = ladder[k];
        const double cx = xs + bx * t, cy = ys + by * t, cz = zs + bz * t;
        const double ex = cx - px, ey = cy - py, ez = cz - pz;
        const double r = t * K.sin_2dphi;
        cand = cand || (((ex * ex + ey * ey) + ez * ez) <= r * r);
    }
    if (!cand) return;
    // the weight of the sample (ModelResult.py:150-161, out_of_shadow = 1); pairs that get here
    // are one in 1e4 of those tested, so it is formed per pair rather than kept per sample
    const double radvel = vy_p + K.vrplanet;
    double gg = K.n_lines > 0 ? lut_interp(lut_view(K.line[0]), radvel) : 0.0;
#pragma unroll
    for (int l = 1; l < 4; l++)
        if (l < K.n_lines) gg += lut_interp(lut_view(K.line[l]), radvel);
    const double weight = nxc_div_const(frac_p * gg, 1e6, rs_1e6);   // 1e6 is mid-range
    if (!(__builtin_fabs(weight) <= 1.7976931348623157e308) || radvel != radvel) my_nonfinite++;
    const double ds = dist * K.sin_dphi;
    const double apix = (3.141592653589793 * (ds * ds)) * K.unit_cm2;         // :194-195
    double wtemp = nxc_div(weight, apix);
    const double hx = xs + bx * q, hy = ys + by * q, hz = zs + bz * q;          // :202-206
    const bool lit = ((hx * hx + hz * hz) > 0x1.0000000000001p+0) || (hy < 0.0);
    wtemp = lit ? wtemp : wtemp * 0.0;
    if (wtemp != 0.0) unsafeAtomicAdd(&radiance[spectrum], wtemp);     // (global, or a workgroup's LDS copy)
    atomicAdd(&npackets[spectrum], 1ull);
    my_pairs++;
    if (included) included[idx_p] = 1;
    if (used_pairs && wtemp > 0.0) {
        const unsigned long long slot = atomicAdd(n_used, 1ull);
        if ((long long)slot < used_cap) {
            used_pairs[slot] = spectrum;
            used_pairs[used_cap + slot] = p + K.row_base;
        }
    }
}

// Per-wave queue of (block, spectrum) candidates between the sphere tests and the pair tests: a
// ring in LDS, filled in lane order (ballot + prefix rank), emptied 64 at a time, so that the pair
// tests run with full waves although only one sphere test in a thousand passes.  All calls are
// wave-uniform.  An entry names its block as (first row << 4 | rows).
constexpr int NXC_LOSQ_SLOTS = 128;                   // < 64 waiting + at most 64 new
constexpr int NXC_LOSQ_BYTES = NXC_LOSQ_SLOTS * (8 + 4);
struct LosQueue {
    int head = 0, tail = 0;
    NXC_DEV void push(bool has, long long blk, int j, int qoff)
    {
        const unsigned long long m = __ballot(has);
        if (m == 0) return;
        if (has) {
            const int lane = threadIdx.x & 63;
            const int slot = (tail + __popcll(m & ((1ull << lane) - 1ull))) & (NXC_LOSQ_SLOTS - 1);
            *reinterpret_cast<long long *>(nxc_lds + qoff + 8 * slot) = blk;
            *reinterpret_cast<int *>(nxc_lds + qoff + 8 * NXC_LOSQ_SLOTS + 4 * slot) = j;
        }
        tail += __popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    NXC_DEV int waiting() const { return tail - head; }
    NXC_DEV bool pop(int qoff, long long &blk, int &j)
    {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n = waiting() < 64 ? waiting() : 64;
        const int lane = threadIdx.x & 63;
        const bool mine = lane < n;
        if (mine) {
            const int slot = (head + lane) & (NXC_LOSQ_SLOTS - 1);
            blk = *reinterpret_cast<const long long *>(nxc_lds + qoff + 8 * slot);
            j = *reinterpret_cast<const int *>(nxc_lds + qoff + 8 * NXC_LOSQ_SLOTS + 4 * slot);
        }
        head += n;
        __builtin_amdgcn_wave_barrier();
        return mine;
    }
};

// Wave-wide exclusive prefix sum / prefix maximum of one int per lane (six shuffle steps);
// `total` receives the reduction over all 64 lanes.
NXC_DEV int wave_excl_sum(int v, int &total)
{
    const int lane = threadIdx.x & 63;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    total = __builtin_amdgcn_readlane(incl, 63);
    return incl - v;
}
NXC_DEV int wave_excl_max(int v, int &total)
{
    const int lane = threadIdx.x & 63;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl = incl > o ? incl : o;
    }
    total = __builtin_amdgcn_readlane(incl, 63);
    const int before = __shfl_up(incl, 1, 64);
    return lane == 0 ? -1 : before;
}

// Can any point of the sphere (centre c, radius R) lie in the cone of spectrum sp (apex a, unit
// axis b, half-angle dphi, between the apex and the cut-off sp[6])?  Only if the centre's distance
// from the axis is at most R + (q_c + R) tan dphi, q_c = (c - a).b, and q_c + R > 0,
// q_c - R < cut-off (the distance to a line and q are 1-Lipschitz).  Conservative: with slack for
// the rounding of perp2, whose terms cancel when the centre is near the line; R = +inf (a block
// with a non-finite coordinate, or culling switched off) passes every cone, R < 0 (no block) none.
NXC_DEV bool los_sphere_hits(const double *__restrict__ sp, double cx, double cy, double cz, double R,
                             double tan_dphi)
{
    // (fused multiply-adds: nothing here is the reference's arithmetic -- the test only has to be
    // conservative, and an fma rounds less than the mul + add it replaces; 19 instructions, not 26)
    const double rx = cx - sp[0], ry = cy - sp[1], rz = cz - sp[2];
    const double qc = __builtin_fma(rz, sp[5], __builtin_fma(ry, sp[4], rx * sp[3]));
    const double perp2 = __builtin_fma(-qc, qc, __builtin_fma(rz, rz, __builtin_fma(ry, ry, rx * rx)));
    const double lim = __builtin_fma(qc + R, tan_dphi, R);
    return (R >= 0.0) && (qc + R > 0.0) && (qc - R < sp[6]) &&
           !(perp2 > __builtin_fma(lim * lim, 1.0 + 1e-6, 1e-9));
}

// T: double, or float for samples as Output.save() stores them (widened exactly, like restore());
// I: the type of the packet-index column (int64, or int32 as save() stores it).
//
// Two levels of culling in front of the exact pair test, both on bounding spheres, prepared by a
// pass of its own over the samples (k_los_blocks) and used by k_los for every spectrum.
//
// BLOCKS.  The rows of an Output are packet-major (Output.py:435-449), so consecutive rows are
// consecutive steps of one packet -- until the packet ends.  A block is at most NXC_LOS_BLOCK
// consecutive rows OF ONE PACKET (the index column says where packets end): a short piece of one
// trajectory, a small sphere.  (Blocks cut from the row numbers alone straddle packet ends -- one
// block in six for the bench cloud -- and such a block is as large as the distance between the two
// packets: 2.5 % of all (block, cone) tests passed, against 0.15 % for blocks that respect the
// packets.)  Without an index column the rows are one packet.
//
// GROUPS.  Eight consecutive block SLOTS form a group with a bounding sphere of its own, and a
// packet of two blocks or more starts on a fresh group (the slots skipped stay empty: 16 % of
// them for the bench cloud), so that a group is a piece of ONE trajectory as well: 10 % of the
// (group, cone) tests pass, 25 % when groups are cut from the block numbers alone.
//
// k_los_blocks: a wave owns a contiguous range of rows; it reads the index column 256 rows at a
// time, finds the packet starts with a prefix maximum across its lanes, places the blocks with a
// prefix scan that knows about the fresh-group rule, and writes {first row << 4 | rows} and the
// bounding sphere of every slot to its region of the scratch arrays (at most one slot per row).
// k_los: persistent waves take the stream 64 slots at a time, one block per lane.  Three levels:
// a lane holds ONE spectrum (64 at a time) and meets the 8 group spheres; the (group, spectrum)
// pairs that pass meet the group's two HALVES (4 blocks each), 32 pairs per wave instruction; the
// (half, spectrum) pairs that pass meet the half's four blocks, 16 per wave instruction.  Sphere
// tests per block and spectrum: one before, 0.24 now.  What passes the block test -- one (block,
// spectrum) pair in 700 -- goes through the per-wave LDS queue, so that its rows meet los_pair
// with full waves.  All the spectra of a launch (up to NXC_LOS_TILE = 512) sit in LDS, so blocks
// and spheres are read once for all of them.  (The kernel is bound by the latency of its
// wave-synchronous stages -- LDS list, ballot, LDS list -- at 16 waves per CU, which the 150 KB of
// tables, spectra and per-wave lists allow: halving the block tests did not change its time.)
// K.cull = 0 (boresights that are not unit vectors) gives every block an infinite radius.
constexpr int NXC_LOS_FORM = 256;        // rows whose packet ids are examined at a time: 4 per lane

// Partner values inside groups of 4 and 8 lanes without going through the LDS crossbar (`__shfl_xor`
// is a ds_bpermute): data-parallel-primitive moves.  quad_xor1 / quad_xor2 pair the lanes of a
// quad; oct_other hands a lane a value from the OTHER quad of its group of eight (for values that
// are already uniform across each quad, e.g. after the two quad steps of a min / max reduction).
template <int CTRL>
NXC_DEV double dpp_move(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
NXC_DEV double quad_xor1(double v) { return dpp_move<0xB1>(v); }      // quad_perm [1,0,3,2]
NXC_DEV double quad_xor2(double v) { return dpp_move<0x4E>(v); }      // quad_perm [2,3,0,1]
NXC_DEV double oct_other(double v) { return dpp_move<0x141>(v); }     // row_half_mirror: lane i <-> 7 - i

// The placement scan.  A lane's blocks move the write position like pos -> pos + a (no fresh
// group among them) or pos -> roundup8(pos + a) + c (a blocks, then a fresh group, then c more
// slots with the later round-ups resolved: they start from a multiple of 8).  Composition is
// associative, so the positions of all lanes come out of six shuffle steps.
struct LosPlace {
    int fresh, a, c;
};
NXC_DEV LosPlace los_place_then(const LosPlace &f, const LosPlace &g)   // first f, then g
{
    if (!f.fresh) return LosPlace{g.fresh, f.a + g.a, g.c};
    if (!g.fresh) return LosPlace{1, f.a, f.c + g.a};
    return LosPlace{1, f.a, ((f.c + g.a + 7) & ~7) + g.c};
}
NXC_DEV long long los_place_apply(const LosPlace &f, long long pos)
{
    return f.fresh ? ((pos + f.a + 7) & ~7ll) + f.c : pos + f.a;
}

// One wave per REGION of NXC_LOS_FORM rows: packet starts from the index column, blocks, placement,
// spheres -- everything from one round of loads of the ids and one of the coordinates.  A region
// starts a packet of its own (a packet cut by a region's edge continues as another "packet" in the
// next region: one block in thirty more than strictly needed).  The region's slots -- a whole
// number of groups -- go to the next free place of ONE stream of slots (a returning atomic per
// region; the order of the regions in the stream does not matter), so that k_los reads dense trips.
#ifndef NXC_LOS_BLOCKS_THREADS_N
#define NXC_LOS_BLOCKS_THREADS_N 1024
#endif
constexpr int NXC_LOS_BLOCKS_THREADS = NXC_LOS_BLOCKS_THREADS_N;     // 16 regions per workgroup, one atomic for all of them
template <typename T, typename I>
// (capped at 64 registers so that two workgroups share a CU: measured no faster, 113 against 116 us)
__global__ void __launch_bounds__(NXC_LOS_BLOCKS_THREADS)
k_los_blocks(int64_t P, int cull, const T *__restrict__ x, const T *__restrict__ y,
             const T *__restrict__ z, const I *__restrict__ index,
             unsigned long long *__restrict__ bdesc, double *__restrict__ bsph,
             unsigned long long *__restrict__ n_slots)
{
    static_assert(NXC_LOS_BLOCK == 8, "groups of 8 slots of 8 rows");
    const int lane = threadIdx.x & 63;
    const int64_t region = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t lo = region * NXC_LOS_FORM < P ? region * NXC_LOS_FORM : P;       // (past the end: empty)
    const int64_t hi = lo + NXC_LOS_FORM < P ? lo + NXC_LOS_FORM : P;
    long long stream0 = 0;             // where this region's slots start in the stream
    auto put = [&](int slot, unsigned long long desc, double cx, double cy, double cz, double R) {
        bdesc[stream0 + slot] = desc;
        nxc_v2d *q = reinterpret_cast<nxc_v2d *>(bsph + 4 * (stream0 + slot));
        nxc_v2d u, v;
        u.x = cx; u.y = cy; v.x = cz; v.y = R;
        q[0] = u; q[1] = v;
    };
    const int64_t r0 = lo + 4 * lane;
    long long id[4] = {0, 0, 0, 0};
    long long prev = 0;
    if (index) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (r0 + k < hi) id[k] = (long long)index[r0 + k];
        if (lane > 0 && r0 < hi) prev = (long long)index[r0 - 1];
    }
    // positions inside the region: rel = 4 lane + k; a row past the end counts as a packet start
    bool bnd[4];
    int last = -1, first = NXC_LOS_FORM;       // my last / first packet start (none: -1 / beyond)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int rel = 4 * lane + k;
        bnd[k] = r0 + k >= hi || rel == 0 || (index && id[k] != (k ? id[k - 1] : prev));
        if (bnd[k]) last = rel;
    }
#pragma unroll
    for (int k = 3; k >= 0; k--)
        if (bnd[k]) first = 4 * lane + k;
    int unused;
    const int before = wave_excl_max(last, unused);          // last start in the lanes below mine
    // first start in the lanes above mine: a suffix minimum
    int after = first;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_down(after, off, 64);
        if (lane + off < 64) after = after < o ? after : o;
    }
    after = __shfl_down(after, 1, 64);
    if (lane == 63) after = NXC_LOS_FORM;
    int start = before;                        // start of the packet of the row before mine
    unsigned long long d[4] = {0, 0, 0, 0};
    bool emit[4], fresh[4];
    LosPlace mine{0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int rel = 4 * lane + k;
        if (bnd[k]) start = rel;
        emit[k] = r0 + k < hi && ((rel - start) & (NXC_LOS_BLOCK - 1)) == 0;
        fresh[k] = false;
        if (emit[k]) {
            int nextb = after;                 // the next packet start after this row
#pragma unroll
            for (int m = 3; m > 0; m--)
                if (m > k && bnd[m]) nextb = 4 * lane + m;
            const int run = nextb - rel;       // rows of this packet from here on (the region's end is a start)
            const int n = run < NXC_LOS_BLOCK ? run : NXC_LOS_BLOCK;
            d[k] = (unsigned long long)(r0 + k) << 4 | (unsigned long long)n;
            fresh[k] = rel == start && run > NXC_LOS_BLOCK;  // a packet of two blocks or more opens a group
            mine = los_place_then(mine, fresh[k] ? LosPlace{1, 0, 1} : LosPlace{0, 1, 0});
        }
    }
    LosPlace incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        LosPlace o;
        o.fresh = __shfl_up(incl.fresh, off, 64);
        o.a = __shfl_up(incl.a, off, 64);
        o.c = __shfl_up(incl.c, off, 64);
        if (lane >= off) incl = los_place_then(o, incl);
    }
    LosPlace all, excl;
    all.fresh = __builtin_amdgcn_readlane(incl.fresh, 63);
    all.a = __builtin_amdgcn_readlane(incl.a, 63);
    all.c = __builtin_amdgcn_readlane(incl.c, 63);
    excl.fresh = __shfl_up(incl.fresh, 1, 64);
    excl.a = __shfl_up(incl.a, 1, 64);
    excl.c = __shfl_up(incl.c, 1, 64);
    int pos = lane == 0 ? 0 : (int)los_place_apply(excl, 0);
    // the last group is completed with empty slots
    const int tail = (int)los_place_apply(all, 0);
    const int full = (tail + 7) & ~7;
    // The blocks change hands: a lane found the blocks that START in its four rows (none, one,
    // rarely more), but the spheres are computed one block per lane -- the descriptors go through
    // LDS by slot number, empty slots stay zero.
    __shared__ unsigned long long slot_desc[NXC_LOS_BLOCKS_THREADS / 64][NXC_LOS_FORM + 8];
    const int wid = threadIdx.x >> 6;
    unsigned long long *const mine_desc = slot_desc[wid];
    for (int s_ = lane; s_ < full; s_ += 64) mine_desc[s_] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!emit[k]) continue;
        if (fresh[k]) pos = (pos + 7) & ~7;
        mine_desc[pos++] = d[k];
    }
    // one returning atomic per WORKGROUP (a single address takes about 80 million a second: one
    // per region would cost half a millisecond for 1e7 rows)
    __shared__ long long wg_base[NXC_LOS_BLOCKS_THREADS / 64 + 1];
    if (lane == 0) wg_base[wid + 1] = full;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long sum = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) { const long long c = wg_base[w + 1]; wg_base[w + 1] = sum; sum += c; }
        wg_base[0] = (long long)atomicAdd(n_slots, (unsigned long long)sum);
    }
    __syncthreads();
    stream0 = wg_base[0] + wg_base[wid + 1];
    for (int s_ = lane; s_ < full; s_ += 64) {
        const unsigned long long desc = mine_desc[s_];
        if (desc == 0ull) { put(s_, 0ull, 0.0, 0.0, 0.0, -1.0); continue; }   // a slot left empty
        const int64_t p0 = (int64_t)(desc >> 4);
        const int nb = (int)(desc & 15ull);
        // bounding sphere: centre of the bounding box, largest distance from it
        double px[NXC_LOS_BLOCK], py[NXC_LOS_BLOCK], pz[NXC_LOS_BLOCK];
#pragma unroll
        for (int r = 0; r < NXC_LOS_BLOCK; r++) {
            const int64_t p = p0 + (r < nb ? r : 0);         // all the loads in flight together
            px[r] = (double)x[p]; py[r] = (double)y[p]; pz[r] = (double)z[p];
        }
        double lox = px[0], hix = lox, loy = py[0], hiy = loy, loz = pz[0], hiz = loz;
#pragma unroll
        for (int r = 1; r < NXC_LOS_BLOCK; r++) {
            lox = __builtin_fmin(lox, px[r]); hix = __builtin_fmax(hix, px[r]);
            loy = __builtin_fmin(loy, py[r]); hiy = __builtin_fmax(hiy, py[r]);
            loz = __builtin_fmin(loz, pz[r]); hiz = __builtin_fmax(hiz, pz[r]);
        }
        double cx = 0.5 * (lox + hix), cy = 0.5 * (loy + hiy), cz = 0.5 * (loz + hiz);
        double R2 = 0.0;
        bool nan_seen = false;
#pragma unroll
        for (int r = 0; r < NXC_LOS_BLOCK; r++) {
            const double ex = px[r] - cx, ey = py[r] - cy, ez = pz[r] - cz;
            const double e2 = (ex * ex + ey * ey) + ez * ez;
            nan_seen = nan_seen || e2 != e2;   // (fmin / fmax drop a NaN: the box alone would not show it)
            R2 = __builtin_fmax(R2, e2);
        }
        // a non-finite coordinate must reach los_pair (it decides such samples like the reference)
        const bool finite = !nan_seen && (R2 <= 1.7976931348623157e308) && (cx == cx) && (cy == cy) && (cz == cz);
        double R = __builtin_sqrt(R2) * (1.0 + 1e-12);
        if (!finite || !cull) { cx = cy = cz = 0.0; R = __builtin_inf(); }
        put(s_, desc, cx, cy, cz, R);
    }
}

constexpr int NXC_LOS_PAIRS = 512;       // (group, spectrum) survivors of 8 tests per lane
constexpr int NXC_LOS_PAIRS2 = 256;      // (half, spectrum) survivors waiting for the block tests
// per wave: candidate queue | 64 block | 16 half-group | 8 group spheres | pairs | (half, spectrum) pairs
constexpr int NXC_LOS_WAVE_BYTES = NXC_LOSQ_BYTES + 64 * 32 + 16 * 32 + 8 * 32 + NXC_LOS_PAIRS * 2 + NXC_LOS_PAIRS2 * 2;
#ifndef NXC_LOS_THREADS
#define NXC_LOS_THREADS 1024
#endif
template <typename T, typename I>
__global__ void __launch_bounds__(NXC_LOS_THREADS)
k_los(LosK K, const unsigned char *__restrict__ blob, int64_t stage_bytes, int64_t S,
      const double *__restrict__ sc, const unsigned long long *__restrict__ n_slots,
      const unsigned long long *__restrict__ bdesc, const double *__restrict__ bsph,
      unsigned long long *__restrict__ pair_list, unsigned *__restrict__ pair_fill,
      unsigned *__restrict__ pair_used, int pair_chunks,
      const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ z,
      const T *__restrict__ vy, const T *__restrict__ frac, const I *__restrict__ index,
      const double *__restrict__ ladder, double *__restrict__ radiance,
      unsigned long long *__restrict__ npackets, unsigned char *__restrict__ included,
      long long used_cap, long long *__restrict__ used_pairs,
      unsigned long long *__restrict__ n_used, DevCounters *__restrict__ ctr)
{
    static_assert(NXC_LOS_TILE <= 512, "a (group, spectrum) pair is 3 + 9 bits");
    stage_tables(blob, stage_bytes);
    const int64_t s0 = (int64_t)blockIdx.y * K.tile_cap;
    const int ns = (int)((S - s0) < K.tile_cap ? (S - s0) : K.tile_cap);
    double *tile = reinterpret_cast<double *>(nxc_lds + K.tile_off);
    // seven doubles per spectrum (the eighth, the ladder length, is only read where a pair is
    // decided): a stride of 14 banks, so the 32 different spectra a wave instruction of the
    // pair stages reads lie in different banks (with a stride of 16 they fell on four)
    for (int w = threadIdx.x; w < ns * NXC_LOS_SP; w += blockDim.x) {
        const int c = w / ns, j = w - c * ns;
        tile[j * NXC_LOS_SP + c] = sc[c * S + s0 + j];
    }
    __syncthreads();
    const double rs_1e6 = nxc_recip_seed(1e6);
    unsigned long long my_pairs = 0, my_nonfinite = 0;
    unsigned long long wave_tests = 0;     // sphere tests of this wave (wave-uniform: scalar adds)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int qoff = (int)K.tile_off + K.tile_cap * NXC_LOS_SP * 8 + wid * NXC_LOS_WAVE_BYTES;
    double *const sph = reinterpret_cast<double *>(nxc_lds + qoff + NXC_LOSQ_BYTES);
    double *const hsph = sph + 64 * 4;
    double *const gsph = hsph + 16 * 4;
    unsigned short *const pairs = reinterpret_cast<unsigned short *>(gsph + 8 * 4);
    unsigned short *const pairs2 = pairs + NXC_LOS_PAIRS;
    const long long count = (long long)*n_slots;              // a multiple of 8
    LosQueue queue;
    // The rows of the queued blocks meet the cheap part of the pair test; what passes -- one
    // (row, spectrum) pair in 25 of those, a few lanes at a time -- is not decided here (the exact
    // test, the ball search, two table lookups, three divisions and the atomics would run for one
    // or two live lanes per wave instruction: a quarter of this kernel's time) but written to a
    // list in memory, 64 to a chunk, which k_los_pairs works through with full waves.  A wave
    // fills chunks of its own (one returning atomic per chunk); when the list is full the pair is
    // decided on the spot.
    int cur_chunk = -1, cur_fill = 0;                 // wave-uniform
    auto drain = [&]() {               // up to 64 queued candidates, one per lane
        long long qb = 0;
        int qj = 0;
        const bool mine = queue.pop(qoff, qb, qj);
        const int64_t q0 = qb >> 4;
        const int qn = mine ? (int)(qb & 15) : 0;
        for (int s_ = 0; s_ < NXC_LOS_BLOCK; s_++) {
            if (__ballot(s_ < qn) == 0) break;
            const int64_t p = q0 + s_;
            double px = 0, py = 0, pz = 0;
            bool hit = false;
            if (s_ < qn) {
                px = (double)x[p]; py = (double)y[p]; pz = (double)z[p];
                hit = los_pair_maybe(K, tile + qj * NXC_LOS_SP, px, py, pz);
            }
            const unsigned long long m = __ballot(hit);
            if (m == 0) continue;
            const int n = __popcll(m);
            if (cur_chunk < 0 || cur_fill + n > 64) {
                if (cur_chunk >= 0 && cur_chunk < pair_chunks && lane == 0) pair_fill[cur_chunk] = (unsigned)cur_fill;
                unsigned c = 0;
                if (lane == 0) c = atomicAdd(pair_used, 1u);
                cur_chunk = __builtin_amdgcn_readfirstlane((int)c);
                cur_fill = 0;
            }
            if (cur_chunk < pair_chunks) {
                if (hit)
                    pair_list[(size_t)cur_chunk * 64 + cur_fill + __popcll(m & ((1ull << lane) - 1ull))] =
                        (unsigned long long)p << 32 | (unsigned long long)(s0 + qj);
                cur_fill += n;
            } else if (hit) {
                double sp8[8];
#pragma unroll
                for (int c = 0; c < NXC_LOS_SP; c++) sp8[c] = tile[qj * NXC_LOS_SP + c];
                sp8[7] = sc[7 * S + s0 + qj];
                los_pair(K, sp8, s0 + qj, p, px, py, pz, (double)vy[p], (double)frac[p],
                         index ? (long long)index[p] - K.index_shift : p + K.row_base, ladder, radiance,
                         npackets, included, used_cap, used_pairs, n_used, rs_1e6, my_pairs, my_nonfinite);
            }
        }
    };
    // a trip: 64 consecutive slots of the stream, one per lane.  The trips differ in how much of
    // them lies near a line of sight, so the waves of a workgroup take theirs dynamically -- from a
    // counter in LDS over the workgroup's own trips (every gridDim.x-th: a returning atomic on ONE
    // global address runs at 80 million a second, which for 3e4 trips was two thirds of the kernel)
    // (the counter lives in the dynamic block, behind the last wave's lists: the kernels address
    // LDS from its start, so nothing static may sit in front)
    unsigned *const wg_next = reinterpret_cast<unsigned *>(
        nxc_lds + K.tile_off + K.tile_cap * NXC_LOS_SP * 8 + (blockDim.x >> 6) * NXC_LOS_WAVE_BYTES);
    if (threadIdx.x == 0) *wg_next = 0;
    __syncthreads();
    // (the next trip's spheres are on their way from memory while this one is worked through)
    auto take = [&](long long &base_, double &x_, double &y_, double &z_, double &r_) {
        unsigned k = 0;
        if (lane == 0) k = atomicAdd(wg_next, 1u);
        k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
        base_ = 64ll * ((long long)blockIdx.x + (long long)k * gridDim.x);
        x_ = y_ = z_ = 0.0; r_ = -1.0;
        if (base_ + lane < count) {
            const nxc_v2d *q = reinterpret_cast<const nxc_v2d *>(bsph + 4 * (base_ + lane));
            const nxc_v2d u = q[0], v = q[1];
            x_ = u.x; y_ = u.y; z_ = v.x; r_ = v.y;
        }
    };
    long long base = 0, nbase;
    int n2 = 0;                                       // (half, spectrum) pairs waiting in pairs2
    auto block_tests = [&]() {
        if (n2 == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int b0 = 0; b0 < n2; b0 += 16) {
            const int bt = b0 + (lane >> 2);
            bool bhit = false;
            int j = 0, slot = 0;
            if (bt < n2) {
                const unsigned e = pairs2[bt];
                slot = (int)(e >> 9) * 4 + (lane & 3);
                j = (int)(e & 511u);
                const double *q = sph + 4 * slot;
                bhit = los_sphere_hits(tile + j * NXC_LOS_SP, q[0], q[1], q[2], q[3], K.tan_dphi);
            }
            wave_tests += 4 * (n2 - b0 < 16 ? n2 - b0 : 16);
            // (one block test in 80 passes: its descriptor comes from memory then)
            const long long bd = bhit ? (long long)bdesc[base + slot] : 0ll;
#if defined(NXC_LOS_EXPERIMENT) && NXC_LOS_EXPERIMENT >= 2
            my_pairs += bhit && bd != 0;
#else
            queue.push(bhit, bd, j, qoff);
            if (queue.waiting() >= 64) drain();
#endif
        }
        __builtin_amdgcn_wave_barrier();              // pairs2 is rewritten from its start
        n2 = 0;
    };
    double cx, cy, cz, R, ncx, ncy, ncz, nR;
    take(nbase, ncx, ncy, ncz, nR);
    for (;;) {
        base = nbase; cx = ncx; cy = ncy; cz = ncz; R = nR;
        if (base >= count) break;
#if defined(NXC_LOS_EXPERIMENT) && NXC_LOS_EXPERIMENT >= 5
        break;
#endif
        take(nbase, ncx, ncy, ncz, nR);
        const bool has = R >= 0.0;
        sph[4 * lane] = cx; sph[4 * lane + 1] = cy; sph[4 * lane + 2] = cz; sph[4 * lane + 3] = R;
        // Bounding spheres of the HALF groups (4 lanes) and of the groups (8 lanes): centre of the
        // box of the blocks' centres, radius to the farthest point of any block (butterfly steps
        // inside the 4, then the 8 lanes)
        const double big = 1.7976931348623157e308;
        double blx = has ? cx : big, bhx = has ? cx : -big, bly = has ? cy : big, bhy = has ? cy : -big,
               blz = has ? cz : big, bhz = has ? cz : -big;
        auto widen = [&](double (*other)(double)) {
            blx = __builtin_fmin(blx, other(blx)); bhx = __builtin_fmax(bhx, other(bhx));
            bly = __builtin_fmin(bly, other(bly)); bhy = __builtin_fmax(bhy, other(bhy));
            blz = __builtin_fmin(blz, other(blz)); bhz = __builtin_fmax(bhz, other(bhz));
        };
        auto reach = [&](double ox, double oy, double oz) {     // from (ox, oy, oz) to my block's far side
            if (!has) return -1.0;
            const double ex = cx - ox, ey = cy - oy, ez = cz - oz;
            return (__builtin_sqrt((ex * ex + ey * ey) + ez * ez) + R) * (1.0 + 1e-12);
        };
        widen(quad_xor1); widen(quad_xor2);                     // uniform across each quad now
        const double hx_ = 0.5 * (blx + bhx), hy_ = 0.5 * (bly + bhy), hz_ = 0.5 * (blz + bhz);
        double HR = reach(hx_, hy_, hz_);
        HR = __builtin_fmax(HR, quad_xor1(HR));
        HR = __builtin_fmax(HR, quad_xor2(HR));
        if ((lane & 3) == 0) {
            double *q = hsph + 4 * (lane >> 2);
            q[0] = hx_; q[1] = hy_; q[2] = hz_; q[3] = HR;
        }
        widen(oct_other);                                       // both quads of the group of eight
        const double gx_ = 0.5 * (blx + bhx), gy_ = 0.5 * (bly + bhy), gz_ = 0.5 * (blz + bhz);
        double GR = reach(gx_, gy_, gz_);
        GR = __builtin_fmax(GR, quad_xor1(GR));
        GR = __builtin_fmax(GR, quad_xor2(GR));
        GR = __builtin_fmax(GR, oct_other(GR));
        // the eight group spheres: in LDS, read back with a wave-uniform address (a broadcast;
        // as 64 scalar registers they spilled)
        if ((lane & 7) == 0) {
            double *q = gsph + 4 * (lane >> 3);
            q[0] = gx_; q[1] = gy_; q[2] = gz_; q[3] = GR;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- 64 spectra at a time.  Level 1: a lane holds ONE spectrum (read from LDS once) and
        //      meets the eight group spheres.  Level 2: the surviving (group, spectrum) pairs meet
        //      the group's two halves, 32 pairs per wave instruction.  Level 3: the surviving
        //      (half, spectrum) pairs meet the half's four blocks, 16 per wave instruction. -------
#if defined(NXC_LOS_EXPERIMENT) && NXC_LOS_EXPERIMENT >= 4
        my_pairs += (unsigned long long)(gsph[3] > 0.0);
        if (ns > 0) continue;
#endif
        for (int c0 = 0; c0 < ns; c0 += 64) {
            const int jm = c0 + lane;
            double spj[NXC_LOS_SP];
#pragma unroll
            for (int c = 0; c < NXC_LOS_SP; c++) spj[c] = tile[(jm < ns ? jm : 0) * NXC_LOS_SP + c];
            int npair = 0;
            // the eight tests first (independent chains the scheduler can interleave), then the
            // eight compactions
            bool ghit[8];
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const double *G = gsph + 4 * g;
                ghit[g] = jm < ns && los_sphere_hits(spj, G[0], G[1], G[2], G[3], K.tan_dphi);
            }
            wave_tests += 8 * (ns - c0 < 64 ? ns - c0 : 64);
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const unsigned long long mask = __ballot(ghit[g]);
                if (ghit[g]) pairs[npair + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)(g << 9 | jm);
                npair += __popcll(mask);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if defined(NXC_LOS_EXPERIMENT) && NXC_LOS_EXPERIMENT >= 3
            my_pairs += npair; npair = 0;
#endif
            for (int at0 = 0; at0 < npair; at0 += 32) {
                const int at = at0 + (lane >> 1);
                bool hit = false;
                unsigned e2 = 0;
                if (at < npair) {
                    const unsigned e = pairs[at];
                    const int hs = (int)(e >> 9) * 2 + (lane & 1);
                    const int j = (int)(e & 511u);
                    const double *q = hsph + 4 * hs;
                    hit = los_sphere_hits(tile + j * NXC_LOS_SP, q[0], q[1], q[2], q[3], K.tan_dphi);
                    e2 = (unsigned)hs << 9 | (unsigned)j;
                }
                wave_tests += 2 * (npair - at0 < 32 ? npair - at0 : 32);
                const unsigned long long mask = __ballot(hit);
                if (hit) pairs2[n2 + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)e2;
                n2 += __popcll(mask);
                // the block tests wait until the list holds enough for full instructions (16 pairs
                // each): taken after every 32 pairs they ran a third full
                if (n2 > NXC_LOS_PAIRS2 - 64) block_tests();
            }
            __builtin_amdgcn_wave_barrier();          // `pairs` is rewritten by the next 64 spectra
        }
        block_tests();                                // before the spheres change
    }
    while (queue.waiting() > 0) drain();
    if (cur_chunk >= 0 && cur_chunk < pair_chunks && lane == 0) pair_fill[cur_chunk] = (unsigned)cur_fill;
    // the counters of the workgroup in one go (wave_tests is wave-uniform: lane 0 carries it)
    flush_counters_wg(reinterpret_cast<unsigned long long *>(wg_next + 2), &ctr->samples,
                      lane == 0 ? wave_tests : 0ull, &ctr->samples_binned, my_pairs, &ctr->nonfinite,
                      my_nonfinite);
}

// The pairs k_los found near a cone, decided and weighed as the reference does it
// (compute_iteration.py:177-213) -- 64 of them per wave instruction.
template <typename T, typename I>
__global__ void __launch_bounds__(NXC_BLOCK)
k_los_pairs(LosK K, const unsigned char *__restrict__ blob, int64_t stage_bytes, int64_t S,
            const double *__restrict__ sc, const unsigned long long *__restrict__ pair_list,
            const unsigned *__restrict__ pair_fill, const unsigned *__restrict__ pair_used,
            int pair_chunks, int lds_sums, const T *__restrict__ x, const T *__restrict__ y,
            const T *__restrict__ z, const T *__restrict__ vy, const T *__restrict__ frac,
            const I *__restrict__ index, const double *__restrict__ ladder,
            double *__restrict__ radiance, unsigned long long *__restrict__ npackets,
            unsigned char *__restrict__ included, long long used_cap,
            long long *__restrict__ used_pairs, unsigned long long *__restrict__ n_used,
            DevCounters *__restrict__ ctr)
{
    stage_tables(blob, stage_bytes);                   // g-value tables; nxc_log's table in the header
    // the ladder of ball centres next to them: a pair walks seven of its rungs
    double *const lad = reinterpret_cast<double *>(nxc_lds + ((stage_bytes + 31) & ~31ll));
    for (int k = threadIdx.x; k < K.n_ladder; k += blockDim.x) lad[k] = ladder[k];
    // The sums of a workgroup's pairs are formed in LDS and handed over once per touched spectrum
    // (lds_sums: S of them fit).  A line of sight through the densest part of the cloud collects
    // tens of thousands of pairs, and atomics on ONE global address run at 80 million a second:
    // added straight to memory they took 0.5 ms for 3.9e5 pairs.
    double *const racc = lad + ((K.n_ladder + 3) & ~3);
    unsigned long long *const nacc = reinterpret_cast<unsigned long long *>(racc + (lds_sums ? S : 0));
    if (lds_sums)
        for (int k = threadIdx.x; k < S; k += blockDim.x) { racc[k] = 0.0; nacc[k] = 0ull; }
    __syncthreads();
    const double rs_1e6 = nxc_recip_seed(1e6);
    unsigned long long my_pairs = 0, my_nonfinite = 0;
    const int lane = threadIdx.x & 63;
    const unsigned used = *pair_used;
    const long long chunks = used < (unsigned)pair_chunks ? used : pair_chunks;
    const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long c = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); c < chunks; c += waves) {
        if (lane < (int)pair_fill[c]) {
            const unsigned long long e = pair_list[c * 64 + lane];
            const int64_t p = (int64_t)(e >> 32), spectrum = (int64_t)(e & 0xffffffffull);
            // everything the pair needs, in one round of loads
            double sp[8];
#pragma unroll
            for (int k = 0; k < 8; k++) sp[k] = sc[k * S + spectrum];
            const double px = (double)x[p], py = (double)y[p], pz = (double)z[p];
            const double vy_p = (double)vy[p], frac_p = (double)frac[p];
            const long long idx_p = index ? (long long)index[p] - K.index_shift : p + K.row_base;
            los_pair(K, sp, spectrum, p, px, py, pz, vy_p, frac_p, idx_p, lad, lds_sums ? racc : radiance,
                     lds_sums ? nacc : npackets, included, used_cap, used_pairs, n_used, rs_1e6, my_pairs,
                     my_nonfinite);
        }
    }
    if (lds_sums) {
        __syncthreads();
        for (int k = threadIdx.x; k < S; k += blockDim.x) {
            if (nacc[k]) atomicAdd(&npackets[k], nacc[k]);
            if (racc[k] != 0.0) unsafeAtomicAdd(&radiance[k], racc[k]);
        }
    }
    flush_counter(&ctr->samples_binned, my_pairs);
    flush_counter(&ctr->nonfinite, my_nonfinite);
}

// ---------------------------------------------------------------------------------------------
// On-device initial states: the uniform-surface / flat|gaussian speed /
// isotropic|radial direction source of initial_state/source_distribution.py:47-62,141-171,198-252,
// one thread per packet, written straight into the resident SoA.  Draws come from a counter-based
// generator (Philox-4x32-10; counter = packet index, draw block, stream; key = seed) so packet i
// is the same whatever the launch geometry or the number of GPUs.  The reference's sampler uses
// NumPy's PCG64 stream, so parity with it is statistical (KS tests); parity with the oracle's
// NumPy Philox restatement is to rounding of sin/cos/asin/log.
// NumPy's PCG64 on the device (numpy/random/src/pcg64/pcg64.h: 128-bit LCG, XSL-RR output; a
// double is (next64 >> 11) * 2^-53 and consumes exactly one output).  The reference draws whole
// vectors one after the other (Output.py:138-139; source_distribution.py:51-62,169-171,202-212),
// so element `row` of draw `vec` of a chunk of `n` packets is output number vec*n + row of the
// stream, i.e. the state after vec*n + row + 1 steps.  A thread jumps there in two moves: to
// row + 1 with the precomputed affine maps of 2^b steps (one 128-bit multiply-add per set bit),
// then by vec*n with that vector's map.  All maps come from the host (nxc_api.hip: pcg_tables).
typedef unsigned __int128 nxc_u128;
constexpr int NXC_PCG_BITS = 40;       // rows below 2^40
constexpr int NXC_PCG_VECS = 8;        // draws per packet
struct PcgK {
    nxc_u128 state;                    // after seeding: PCG64(seed).state['state']['state']
    long long row0;                    // row of this call's first packet inside its chunk's draws
    const nxc_u128 *maps;              // [NXC_PCG_BITS + NXC_PCG_VECS][2]: {multiplier, increment}
};

NXC_DEV nxc_u128 pcg_row_state(const PcgK &P, long long row)
{
    nxc_u128 s = P.state;
    unsigned long long d = (unsigned long long)(P.row0 + row) + 1ull;
    for (int b = 0; d != 0; b++, d >>= 1)
        if (d & 1ull) s = P.maps[2 * b] * s + P.maps[2 * b + 1];
    return s;
}

// the uniform of draw `vec` for the packet whose row state is s
NXC_DEV double pcg_uniform(const PcgK &P, nxc_u128 s, int vec)
{
    const nxc_u128 *m = P.maps + 2 * (NXC_PCG_BITS + vec);
    const nxc_u128 t = m[0] * s + m[1];
    const unsigned long long hi = (unsigned long long)(t >> 64), lo = (unsigned long long)t;
    const unsigned rot = (unsigned)(hi >> 58);                      // state >> 122
    const unsigned long long x = hi ^ lo;
    const unsigned long long r = (x >> rot) | (x << ((64u - rot) & 63u));
    return (double)(r >> 11) * 0x1p-53;
}

struct SourceK {
    double endtime, exobase, sinlat0, sinlat1, lon0, lon1, vprob, vwidth, unit_km;
    double sinalt0, sinalt1, az0, az1;
    int random_time, speed_type, angular_type, is_planet;
    unsigned long long seed;
    long long first_index;
    int spatial_type, n_speed, map_nlon, map_nlat;
    double map_max;                      // accept/reject ceiling = max of the density map
    const double *speed_cdf, *speed_v, *map;
    int generator, max_trials;                 // 0 = Philox-4x32-10, 1 = NumPy's PCG64 stream
    PcgK pcg;
    long long stride, offset;            // the packets go to soa[c * stride + offset + i]
};

// diagnostics: out[vec][i] = uniform of draw `vec` for row i (the parity test compares them with
// numpy.random.default_rng(seed).random(n) bit for bit)
__global__ void __launch_bounds__(NXC_BLOCK)
k_pcg_uniforms(PcgK P, int nvec, int64_t count, double *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const nxc_u128 s = pcg_row_state(P, i);
        for (int v = 0; v < nvec; v++) out[v * count + i] = pcg_uniform(P, s, v);
    }
}

// np.interp(x, xp, fp) for a non-decreasing xp in global memory: bisection for the last node
// <= x, then slope*(x - xp[j]) + fp[j] (numpy compiled_base.c arr_interp).
NXC_DEV double interp_global(const double *__restrict__ xp, const double *__restrict__ fp, int n,
                             double x)
{
    if (!(x > xp[0])) return fp[0];
    if (!(x < xp[n - 1])) return fp[n - 1];
    int lo = 0, hi = n - 1;                      // xp[lo] <= x < xp[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (x >= xp[mid]) lo = mid; else hi = mid;
    }
    const double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
    return slope * (x - xp[lo]) + fp[lo];
}

// Bilinear value of the density map at (lon, lat) inside its grid (the linear interpn of
// math/randomdeviates.py:66); weights in the order ((1-tx)(1-ty) f00 + (1-tx) ty f01) +
// (tx (1-ty) f10 + tx ty f11).
NXC_DEV double map_bilinear(const SourceK &K, double lon, double lat)
{
    const double TWO_PI = 6.283185307179586, HALF_PI = 1.5707963267948966;
    const double gx = lon / (TWO_PI / (K.map_nlon - 1));
    const double gy = (lat + HALF_PI) / (3.141592653589793 / (K.map_nlat - 1));
    int i = (int)gx, j = (int)gy;
    i = i < 0 ? 0 : (i > K.map_nlon - 2 ? K.map_nlon - 2 : i);
    j = j < 0 ? 0 : (j > K.map_nlat - 2 ? K.map_nlat - 2 : j);
    const double tx = gx - (double)i, ty = gy - (double)j;
    const double *row0 = K.map + (long long)i * K.map_nlat + j, *row1 = row0 + K.map_nlat;
    return ((1.0 - tx) * (1.0 - ty) * row0[0] + (1.0 - tx) * ty * row0[1]) +
           (tx * (1.0 - ty) * row1[0] + tx * ty * row1[1]);
}

// Rejection trials per packet: the host sizes the budget to the map (32 / acceptance rate, so that
// a packet fails to find a launch point with probability e^-32) between these bounds; a packet
// that never passes is reported, and the call fails.
constexpr int NXC_SPOT_MIN_TRIALS = 4096, NXC_SPOT_MAX_TRIALS = 1 << 18;
constexpr unsigned NXC_SPOT_BLOCK0 = 16;    // Philox draw blocks 16 + 2t, 17 + 2t of trial t

__global__ void __launch_bounds__(NXC_BLOCK)
k_sample(SourceK K, int64_t n, double *__restrict__ soa, DevCounters *__restrict__ ctr)
{
    const double TWO_PI = 6.283185307179586;
    unsigned long long my_unfinished = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long gi = (unsigned long long)(K.first_index + i);
        double ut = 0.0, ulat, ulon, uspd, ualt = 0.0, uaz = 0.0;
        if (K.generator == 1) {
            // the reference's draw order: [launch time] sin(latitude), longitude, speed,
            // [sin(altitude), azimuth] -- each a whole vector of the stream
            const nxc_u128 st = pcg_row_state(K.pcg, i);
            int v = 0;
            if (K.random_time) ut = pcg_uniform(K.pcg, st, v++);
            ulat = pcg_uniform(K.pcg, st, v++);
            ulon = pcg_uniform(K.pcg, st, v++);
            uspd = pcg_uniform(K.pcg, st, v++);
            if (K.angular_type != 0) {
                ualt = pcg_uniform(K.pcg, st, v++);
                uaz = pcg_uniform(K.pcg, st, v++);
            }
        } else {
            philox_pair(gi, 0, NXC_STREAM_SOURCE, K.seed, ut, ulat);
            philox_pair(gi, 1, NXC_STREAM_SOURCE, K.seed, ulon, uspd);
            philox_pair(gi, 2, NXC_STREAM_SOURCE, K.seed, ualt, uaz);
        }
        const double time = K.random_time ? ut * K.endtime : K.endtime;       // Output.py:136-139
        double lat, lon;
        if (K.spatial_type == 0) {                                             // uniform :51-62
            lat = asin(K.sinlat0 + (K.sinlat1 - K.sinlat0) * ulat);
            lon = fmod(K.lon0 + (K.lon1 - K.lon0) * ulon, TWO_PI);
        } else {                                                               // surface spot :96-118
            bool accepted = false;
            lon = 0.0; lat = 0.0;
            for (int t = 0; t < K.max_trials && !accepted; t++) {
                double ux, uy, uf, unused;
                philox_pair(gi, NXC_SPOT_BLOCK0 + 2u * (unsigned)t, NXC_STREAM_SOURCE, K.seed, ux, uy);
                philox_pair(gi, NXC_SPOT_BLOCK0 + 2u * (unsigned)t + 1u, NXC_STREAM_SOURCE, K.seed,
                            uf, unused);
                lon = ux * TWO_PI;
                lat = uy * 3.141592653589793 - 1.5707963267948966;
                accepted = uf * K.map_max < map_bilinear(K, lon, lat);
            }
            my_unfinished += !accepted;
        }
        const double clat = cos(lat);
        const double x0 = (K.is_planet ? 1.0 : -1.0) * K.exobase * sin(lon) * clat;   // :12-28
        const double y0 = -K.exobase * cos(lon) * clat;
        const double z0 = K.exobase * sin(lat);
        double v;
        if (K.speed_type == 0) {                                               // flat :169-171
            v = uspd * 2 * K.vwidth + K.vprob - K.vwidth;
        } else if (K.speed_type == 1) {                                        // gaussian :141-147
            double g0, g1;
            philox_pair(gi, 3, NXC_STREAM_SOURCE, K.seed, g0, g1);
            const double zn = sqrt(-2.0 * log(1.0 - g0)) * cos(TWO_PI * g1);
            v = K.vwidth == 0.0 ? K.vprob : zn * K.vwidth + K.vprob;
        } else {                                                               // tabulated :148-168
            v = interp_global(K.speed_cdf, K.speed_v, K.n_speed, uspd);
        }
        v = v / K.unit_km;                                                     // :184
        double alt, az;
        if (K.angular_type == 0) {                                             // radial :198-201
            alt = 1.5707963267948966; az = 0.0;
        } else {                                                               // isotropic :202-212
            alt = asin(ualt * (K.sinalt1 - K.sinalt0) + K.sinalt0);
            az = K.az0 + (K.az1 - K.az0) * uaz;
        }
        const double v_rad = sin(alt), v_t0 = cos(alt) * cos(az), v_t1 = cos(alt) * sin(az);
        const double rn = sqrt((x0 * x0 + y0 * y0) + z0 * z0);                 // :236-245
        const double en = sqrt(y0 * y0 + x0 * x0);
        const double n0 = -z0 * x0, n1 = -z0 * y0, n2 = x0 * x0 + y0 * y0;
        const double nn = sqrt((n0 * n0 + n1 * n1) + n2 * n2);
        const double dx = (v_t0 * (n0 / nn) + v_t1 * (y0 / en)) + v_rad * (x0 / rn);   // :247-248
        const double dy = (v_t0 * (n1 / nn) + v_t1 * (-x0 / en)) + v_rad * (y0 / rn);
        const double dz = (v_t0 * (n2 / nn) + v_t1 * 0.0) + v_rad * (z0 / rn);
        double *__restrict__ dst = soa + K.offset + i;
        dst[0] = time;
        dst[1 * K.stride] = x0; dst[2 * K.stride] = y0; dst[3 * K.stride] = z0;
        dst[4 * K.stride] = dx * v; dst[5 * K.stride] = dy * v; dst[6 * K.stride] = dz * v;
        dst[7 * K.stride] = 1.0;
    }
    flush_counter(&ctr->unfinished, my_unfinished);
}

// Queue order on the device: counting sort of the packet indices by decreasing |v|^2.
// The packets are `n` rows of an array whose columns are `stride` apart (a whole resident set:
// stride = n; a piece of one that is still being uploaded: stride = the set's size).
constexpr int NXC_ORDER_BINS = 4096;

// BY_STEPS: the key is the packet's known number of steps (after a counting pass the lifetimes are
// exact, so the queue becomes longest-processing-time-first proper and the lanes of a wave, which
// hold neighbours of the sorted order, finish together).
NXC_DEV int steps_bin(const long long *__restrict__ steps, int64_t i, double scale)
{
    const double f = (double)steps[i] * scale;
    int b = f < (double)NXC_ORDER_BINS ? (int)f : NXC_ORDER_BINS - 1;
    return NXC_ORDER_BINS - 1 - (b < 0 ? 0 : b);
}

NXC_DEV int speed_bin(const double *__restrict__ soa, int64_t stride, int64_t i, double scale)
{
    const double vx = soa[4 * stride + i], vy = soa[5 * stride + i], vz = soa[6 * stride + i];
    const double f = (vx * vx + vy * vy + vz * vz) * scale;
    int b = (f >= 0.0 && f < (double)NXC_ORDER_BINS) ? (int)f : (f >= (double)NXC_ORDER_BINS ? NXC_ORDER_BINS - 1 : 0);
    return NXC_ORDER_BINS - 1 - b;
}

// The queue key of the ADAPTIVE driver: remaining time over launch speed, descending.  A packet's
// attempts run one after the other in one lane, so the launch lasts at least as long as its
// longest chain, and the long chains belong to SLOW packets with much time left (bound packets
// that keep a small step for the rest of their flight), not to the fast ones: measured on 1e5
// packets of the bench workload (tests/tools/var_schedule.py), the lane-refill schedule's makespan
// is 6965 attempts for fastest-first, 5791 as sampled, 5069 for this key and 5017 for
// longest-first, which needs the answer (at 50 packets per lane: 27 818 / 26 482 / 24 863 against
// a mean load of 24 319).
NXC_DEV double flight_key(const double *__restrict__ soa, int64_t stride, int64_t i)
{
    const double vx = soa[4 * stride + i], vy = soa[5 * stride + i], vz = soa[6 * stride + i];
    return soa[i] / __builtin_sqrt(vx * vx + vy * vy + vz * vz);
}
NXC_DEV int flight_bin(const double *__restrict__ soa, int64_t stride, int64_t i, double scale)
{
    const double f = flight_key(soa, stride, i) * scale;       // NaN (0 / 0) goes last, +inf first
    int b = (f >= 0.0 && f < (double)NXC_ORDER_BINS) ? (int)f : (f >= (double)NXC_ORDER_BINS ? NXC_ORDER_BINS - 1 : 0);
    return NXC_ORDER_BINS - 1 - b;
}
// KEY: 0 = |v|^2, 1 = known step counts, 2 = the adaptive driver's flight key
template <int KEY>
NXC_DEV int order_bin(const double *__restrict__ soa, const long long *__restrict__ steps, int64_t stride,
                      int64_t i, double scale)
{
    return KEY == 1 ? steps_bin(steps, i, scale) : KEY == 2 ? flight_bin(soa, stride, i, scale)
                                                              : speed_bin(soa, stride, i, scale);
}

// The bin scale: given by the host, or (max_bits != null) derived on the device from the largest
// key that k_speed_max left there, so that a piece can be ordered without a host round trip.
NXC_DEV double order_scale(double scale, const unsigned long long *__restrict__ max_bits)
{
    if (!max_bits) return scale;
    const double m = __longlong_as_double((long long)*max_bits);
    return m > 0.0 ? (double)(NXC_ORDER_BINS - 1) / m : 0.0;
}

// Largest finite |v|^2 of the packets (bit pattern of a non-negative double: its order as an
// unsigned integer is its order as a number).
template <bool FLIGHT = false>      // FLIGHT: the largest finite flight key instead
__global__ void __launch_bounds__(NXC_BLOCK)
k_speed_max(const double *__restrict__ soa, int64_t stride, int64_t n, unsigned long long *__restrict__ out)
{
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double vx = soa[4 * stride + i], vy = soa[5 * stride + i], vz = soa[6 * stride + i];
        const double f = FLIGHT ? flight_key(soa, stride, i) : vx * vx + vy * vy + vz * vz;
        if (f <= 1.7976931348623157e308 && f > m) m = f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 0.0)
        atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// LDS: privatised bins (the launches that order a whole resident set); !LDS: straight global
// atomics, for the launches that run BESIDE a persistent kernel whose block leaves no LDS free.
template <int KEY, bool LDS>
__global__ void __launch_bounds__(NXC_BLOCK)
k_order_hist(const double *__restrict__ soa, const long long *__restrict__ steps, int64_t stride,
             int64_t n, double scale_, const unsigned long long *__restrict__ max_bits,
             unsigned long long *__restrict__ hist)
{
    const double scale = order_scale(scale_, max_bits);
    if (LDS) {
        __shared__ unsigned lh[NXC_ORDER_BINS];
        for (int b = threadIdx.x; b < NXC_ORDER_BINS; b += blockDim.x) lh[b] = 0;
        __syncthreads();
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            atomicAdd(&lh[order_bin<KEY>(soa, steps, stride, i, scale)], 1u);
        __syncthreads();
        for (int b = threadIdx.x; b < NXC_ORDER_BINS; b += blockDim.x)
            if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            atomicAdd(&hist[order_bin<KEY>(soa, steps, stride, i, scale)], 1ull);
    }
}

// hist[NXC_ORDER_BINS] -> exclusive running sum in place: the scatter cursors.  One wave, no LDS
// (it may have to run on a CU whose LDS a persistent kernel owns): lane l sums its 64 bins, the
// lanes' totals are scanned with shuffles.
__global__ void __launch_bounds__(64)
k_order_scan(unsigned long long *__restrict__ hist)
{
    constexpr int PER = NXC_ORDER_BINS / 64;
    const int lane = threadIdx.x;
    unsigned long long sum = 0;
    for (int j = 0; j < PER; j++) sum += hist[lane * PER + j];
    unsigned long long incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    unsigned long long acc = incl - sum;
    for (int j = 0; j < PER; j++) {
        const unsigned long long v = hist[lane * PER + j];
        hist[lane * PER + j] = acc;
        acc += v;
    }
}

template <int KEY>
__global__ void __launch_bounds__(NXC_BLOCK)
k_order_scatter(const double *__restrict__ soa, const long long *__restrict__ steps, int64_t stride,
                int64_t n, double scale_, const unsigned long long *__restrict__ max_bits,
                unsigned long long *__restrict__ cursor, unsigned *__restrict__ order, unsigned base,
                double *__restrict__ queue)
{
    // Every packet takes the next free position of its bin and goes there whole: the eight columns
    // are read along the packets (coalesced) and leave as ONE 64-byte record per packet -- the
    // queue the persistent kernels read front to back.  (Until round 3 the positions were
    // scattered first and the columns gathered by a second kernel, whose 8-byte reads at random
    // rows fetched six times the bytes they used.)
    const double scale = order_scale(scale_, max_bits);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = soa[c * stride + i];
        const unsigned long long pos =
            atomicAdd(&cursor[order_bin<KEY>(soa, steps, stride, i, scale)], 1ull);
        order[pos] = base + (unsigned)i;
        nxc_v2d *rec = reinterpret_cast<nxc_v2d *>(queue + 8 * pos);
#pragma unroll
        for (int c = 0; c < 4; c++) { nxc_v2d t; t.x = v[2 * c]; t.y = v[2 * c + 1]; rec[c] = t; }
    }
}

// Sum of the per-piece counters of a streamed pass (each piece has its own queue head).
__global__ void k_sum_counters(const DevCounters *__restrict__ pieces, int n, DevCounters *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        DevCounters t{};
        for (int p = 0; p < n; p++) {
            t.particle_steps += pieces[p].particle_steps; t.samples += pieces[p].samples;
            t.samples_binned += pieces[p].samples_binned; t.nonfinite += pieces[p].nonfinite;
            t.bad_step += pieces[p].bad_step; t.neg_frac += pieces[p].neg_frac;
            t.unfinished += pieces[p].unfinished; t.wave_trips += pieces[p].wave_trips;
        }
        *out = t;
    }
}
