// Micro-benchmark: device-to-device streaming copy on gfx950 -- which form reaches the ~6.3 TB/s that
// MI355X_MICROARCH.md measures for a float4 copy?  (The denominator of bench.py's
// roofline.frac_of_measured_peak is nxc_stream_copy_gbs; round 3's kernel reached 5.1-5.2 TB/s.)
// hipcc -O3 --offload-arch=gfx950 tools/ubench_copy.hip -o tools/ubench_copy.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

// one float4 per thread, grid covers the array
__global__ void __launch_bounds__(256) k_one(const v4 *__restrict__ s, v4 *__restrict__ d, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}

// U float4 per thread, block-contiguous tiles: a block copies U consecutive 4 KB runs
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_tile(const v4 *__restrict__ s, v4 *__restrict__ d, int64_t n)
{
    const int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
    v4 r[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t i = base + (int64_t)u * 256;
        if (i < n) r[u] = NT ? __builtin_nontemporal_load(s + i) : s[i];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t i = base + (int64_t)u * 256;
        if (i < n) { if (NT) __builtin_nontemporal_store(r[u], d + i); else d[i] = r[u]; }
    }
}

// persistent grid-stride, U loads in flight (round 3's k_stream_copy is <4, true> of this)
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_stride(const v4 *__restrict__ s, v4 *__restrict__ d, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4 r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) { if (NT) __builtin_nontemporal_store(r[u], d + i + u * stride); else d[i + u * stride] = r[u]; }
    }
    for (; i < n; i += stride) d[i] = s[i];
}

template <class F>
static int run(const char *name, F launch, int64_t n16, hipStream_t st)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 6; r++) {
        CK(hipEventRecord(a, st));
        launch();
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-34s %8.3f ms  %7.1f GB/s (read + written)\n", name, best, 2.0 * n16 * 16 / (best * 1e-3) / 1e9);
    return 0;
}

int main(int argc, char **argv)
{
    const int64_t bytes = argc > 1 ? atoll(argv[1]) : (int64_t)1 << 31;
    const int64_t n = bytes / 16;
    v4 *s, *d;
    CK(hipMalloc((void **)&s, n * 16)); CK(hipMalloc((void **)&d, n * 16));
    CK(hipMemset(s, 1, n * 16)); CK(hipMemset(d, 0, n * 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cu = p.multiProcessorCount;
    printf("%s, %d CUs, %lld bytes each way\n", p.gcnArchName, cu, (long long)bytes);
    run("hipMemcpyAsync D2D", [&] { (void)hipMemcpyAsync(d, s, n * 16, hipMemcpyDeviceToDevice, st); }, n, st);
    run("one float4 per thread", [&] { hipLaunchKernelGGL(k_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, d, n); }, n, st);
#define TILE(U, NT) run("tile U=" #U " nt=" #NT, [&] { hipLaunchKernelGGL((k_tile<U, NT>), dim3((unsigned)((n + 256 * U - 1) / (256 * U))), dim3(256), 0, st, s, d, n); }, n, st)
    TILE(2, false); TILE(4, false); TILE(8, false); TILE(4, true); TILE(8, true);
#define STRIDE(U, NT, G) run("stride U=" #U " nt=" #NT " grid=cu*" #G, [&] { hipLaunchKernelGGL((k_stride<U, NT>), dim3((unsigned)(cu * G)), dim3(256), 0, st, s, d, n); }, n, st)
    STRIDE(4, true, 32); STRIDE(4, false, 32); STRIDE(4, false, 8); STRIDE(8, false, 8); STRIDE(8, false, 16);
    STRIDE(8, true, 8); STRIDE(2, false, 64);
    return 0;
}
