// Microbenchmark (diagnostic, not part of the product): issue rate of v_mfma_f64_16x16x4_f64 against v_fma_f64 on gfx950 -- the
// arithmetic behind the decision NOT to accumulate the Hessian as dense J^T diag(d) J tiles on the matrix cores:
// a dense 32 x 32 tile update costs 1024 multiply-adds per pixel where the thresholded sparse accumulation needs ~25.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma(double *out, int iters)
{
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0.x + acc1.y + acc2.z + acc3.w;
}

__global__ __launch_bounds__(256) void k_fma(double *out, int iters)
{
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
        c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
        c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

int main()
{
    const int grid = 256 * 8, iters = 20000;
    double *out; CHECK(hipMalloc(&out, sizeof(double) * grid * 256));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = 2.0 * 16 * 16 * 4 * 4.0 * iters * (double)grid * 4;      // 4 MFMAs per iteration, 4 wavefronts per workgroup
        if (rep) printf("v_mfma_f64_16x16x4_f64: %.1f TFLOP/s  (%.1f cycles per MFMA and SIMD at 2.4 GHz, 8 workgroups of 4 wavefronts per CU)\n", flop / ms / 1e9,
                        ms * 1e-3 * 2.4e9 / (4.0 * iters * 8.0 * 4 / 4));
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flop2 = 2.0 * 8.0 * iters * (double)grid * 256;
        if (rep) printf("v_fma_f64:              %.1f TFLOP/s\n", flop2 / ms / 1e9);
    }
    return 0;
}
