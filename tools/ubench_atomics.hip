// Micro-benchmark: scattered global atomic throughput on gfx950 for the image accumulation design.
// hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/ubench_atomics.hip -o /tmp/ubench_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ inline uint32_t rng(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int MODE>
__global__ void k(void *buf, uint32_t npix_mask, int iters, float hot)
{
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    double *d = (double *)buf; unsigned long long *u = (unsigned long long *)buf; uint32_t *w = (uint32_t *)buf;
    for (int i = 0; i < iters; i++) {
        uint32_t r = rng(s);
        uint32_t pix = r & npix_mask;
        if ((rng(s) & 0xffff) < (uint32_t)(hot * 65536.f)) pix &= 0x3fff;   // hot region: 16k pixels
        if (MODE == 0) atomicAdd(&w[pix], 1u);
        else if (MODE == 1) atomicAdd(&u[pix], 1ull);
        else if (MODE == 2) unsafeAtomicAdd(&d[pix], 1.0);
        else if (MODE == 3) __hip_atomic_fetch_add(&u[pix], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 4) __hip_atomic_fetch_add(&d[pix], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 5) { unsafeAtomicAdd(&d[2 * pix], 1.0); atomicAdd(&u[2 * pix + 1], 1ull); }   // interleaved pair
        else if (MODE == 6) __hip_atomic_fetch_add(&u[pix], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else if (MODE == 7) d[pix] += 1.0;   // non-atomic RMW (for reference; racy)
        else if (MODE == 8) {   // lane pairs (2k, 2k+1) add to the two halves of one 16-B pixel record
            uint32_t ppix = __shfl(pix, (threadIdx.x & 63) & ~1u, 64);
            unsafeAtomicAdd(&d[2 * ppix + (threadIdx.x & 1)], 1.0);
        }
        else if (MODE == 9) {   // lanes l and l+32 add to the two halves of one 16-B pixel record
            uint32_t ppix = __shfl(pix, (threadIdx.x & 31), 64);
            unsafeAtomicAdd(&d[2 * ppix + ((threadIdx.x >> 5) & 1)], 1.0);
        }
        else if (MODE == 10) {  // groups of 4 lanes add to 4 consecutive doubles (one 32-B half line)
            uint32_t ppix = __shfl(pix, (threadIdx.x & 63) & ~3u, 64);
            unsafeAtomicAdd(&d[(2 * ppix & ~3u) + (threadIdx.x & 3)], 1.0);
        }
    }
}

template <int MODE>
int run(const char *name, void *buf, uint32_t mask, float hot)
{
    const int iters = 2048, blocks = 256 * 8, threads = 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<blocks, threads>>>(buf, mask, 16, hot); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); k<MODE><<<blocks, threads>>>(buf, mask, iters, hot); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)iters * blocks * threads * (MODE == 5 ? 2 : 1);
    printf("%-34s hot=%.2f  %8.2f ms  %8.2f G atomics/s\n", name, hot, ms, ops / ms / 1e6);
    return 0;
}

int main()
{
    void *buf; size_t bytes = (size_t)(1 << 18) * 16;   // 256k pixels x 16 B
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    uint32_t mask = (1 << 18) - 1;
    for (float hot : {0.0f, 0.9f}) {
        run<0>("u32 agent", buf, mask, hot);
        run<1>("u64 agent", buf, mask, hot);
        run<2>("f64 unsafeAtomicAdd", buf, mask, hot);
        run<3>("u64 workgroup scope", buf, mask, hot);
        run<4>("f64 workgroup scope", buf, mask, hot);
        run<5>("f64+u64 interleaved pair", buf, mask, hot);
        run<6>("u64 wavefront scope", buf, mask, hot);
        run<7>("f64 plain RMW (racy)", buf, mask, hot);
        run<8>("f64 pair, adjacent lanes", buf, mask, hot);
        run<9>("f64 pair, lanes l / l+32", buf, mask, hot);
        run<10>("f64 quad, adjacent lanes", buf, mask, hot);
    }
    return 0;
}
