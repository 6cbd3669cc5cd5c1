// Micro-benchmark: issue rate of the fp64 VALU operations the step loop is made of (gfx950).
// hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(double *out, int iters, double seed)
{
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __builtin_fma(a[i], 1.0000001, 1e-9);
            else if (OP == 1) a[i] = __builtin_amdgcn_rcp(a[i]);
            else if (OP == 2) a[i] = __builtin_amdgcn_rsq(a[i]);
            else if (OP == 3) a[i] = a[i] * 1.0000001;
            else if (OP == 4) a[i] = a[i] + 1e-9;
            else if (OP == 5) a[i] = (double)__builtin_amdgcn_rcpf((float)a[i]);
            else if (OP == 6) a[i] = a[i] > 1.5 ? a[i] - 0.5 : a[i] + 0.25;       // compare + select
            else if (OP == 7) a[i] = (double)(float)a[i];                           // f32 round trip
            else if (OP == 8) a[i] = __builtin_sqrt(a[i]);                          // full IEEE sqrt sequence
            else if (OP == 9) a[i] = 1.0 / a[i];                                    // full IEEE division
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
int run(const char *name, double *buf, int per_op)
{
    const int iters = 4096, blocks = 256 * 12, threads = 256;    // 12 waves per SIMD-quad: 3 per SIMD
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<OP><<<blocks, threads>>>(buf, 16, 1.25); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); k<OP><<<blocks, threads>>>(buf, iters, 1.25); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double waveops = (double)iters * 8 * blocks * threads / 64;          // wave-level operations
    double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;                       // at the nominal clock
    printf("%-28s %8.3f ms  %7.2f nominal cycles per wave-op per SIMD (%d instr per op)\n", name, ms,
           simd_cycles / waveops, per_op);
    return 0;
}

int main()
{
    double *buf; CK(hipMalloc(&buf, sizeof(double) * 256 * 12 * 256));
    run<0>("v_fma_f64", buf, 1);
    run<3>("v_mul_f64", buf, 1);
    run<4>("v_add_f64", buf, 1);
    run<1>("v_rcp_f64", buf, 1);
    run<2>("v_rsq_f64", buf, 1);
    run<5>("cvt + v_rcp_f32 + cvt", buf, 3);
    run<6>("v_cmp + 2 v_cndmask + add/sub", buf, 5);
    run<7>("f32 round trip (2 cvt)", buf, 2);
    run<8>("IEEE sqrt sequence", buf, 0);
    run<9>("IEEE division sequence", buf, 0);
    return 0;
}
