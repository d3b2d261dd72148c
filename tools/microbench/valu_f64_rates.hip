// Microbenchmark (diagnostic, not part of the product): issue cost of the vector instructions the pixel pass of sdsm_k_solve is made of,
// relative to v_fma_f64, on gfx950: conversions f32 -> f64, f64 add / mul / fma, 32-bit integer and select instructions, v_rcp_f64 / v_rsq_f64.
// One workgroup of 256 threads per compute unit x 8, eight independent chains per lane: cycles per wavefront instruction and SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_f64_rates valu_f64_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_op(double *out, const float *in, int iters)
{
    double c[8];
    float f[8];
    int u[8];
    for (int k = 0; k < 8; k++) { c[k] = 1.0 + k + threadIdx.x * 1e-9; f[k] = in[(threadIdx.x + k) & 255]; u[k] = threadIdx.x + k; }
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (OP == 0) c[k] = fma(a, b, c[k]);
            else if (OP == 1) c[k] = c[k] + a;
            else if (OP == 2) c[k] = c[k] * a;
            else if (OP == 3) { double t; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(t) : "v"(f[k])); c[k] = t; }            // (only the conversion is counted: the move is a rename)
            else if (OP == 4) { float t; asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(t) : "v"(c[k])); f[k] = t; }
            else if (OP == 5) { asm volatile("v_add_u32 %0, %1, %2" : "=v"(u[k]) : "v"(u[k]), "v"(i)); }
            else if (OP == 6) { double t; asm volatile("v_rcp_f64 %0, %1" : "=v"(t) : "v"(c[k])); c[k] = t; }
            else if (OP == 7) { double t; asm volatile("v_rsq_f64 %0, %1" : "=v"(t) : "v"(c[k])); c[k] = t; }
            else if (OP == 8) { asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[k]) : "v"(u[k]), "v"(i)); }
            else if (OP == 9) { float t; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(f[k]), "v"(f[(k + 1) & 7]), "v"(f[k])); f[k] = t; }
        }
    }
    double s = 0;
    for (int k = 0; k < 8; k++) s += c[k] + f[k] + u[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
int run(const char *name, double *out, const float *in, double ref_ms, double *ms_out)
{
    const int grid = 256 * 8, iters = 20000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_op<OP>, dim3(grid), dim3(256), 0, 0, out, in, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    // wavefront instructions per SIMD: 8 workgroups x 4 wavefronts per CU over 4 SIMDs = 8 wavefronts per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / (8.0 * iters * 8.0);
    printf("%-16s %.3f ms  %.1f cycles per wavefront instruction and SIMD at 2.4 GHz%s", name, ms, cyc, ref_ms > 0 ? "" : "\n");
    if (ref_ms > 0) printf("  (%.2f x v_fma_f64)\n", ms / ref_ms);
    *ms_out = ms;
    return 0;
}

int main()
{
    double *out; float *in;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * 8 * 256)); CHECK(hipMalloc(&in, 1024)); CHECK(hipMemset(in, 0x3f, 1024));
    double ref = 0, ms;
    if (run<0>("v_fma_f64", out, in, 0, &ref)) return 1;
    run<1>("v_add_f64", out, in, ref, &ms); run<2>("v_mul_f64", out, in, ref, &ms); run<3>("v_cvt_f64_f32", out, in, ref, &ms); run<4>("v_cvt_f32_f64", out, in, ref, &ms);
    run<5>("v_add_u32", out, in, ref, &ms); run<8>("v_cndmask_b32", out, in, ref, &ms); run<9>("v_fma_f32", out, in, ref, &ms); run<6>("v_rcp_f64", out, in, ref, &ms); run<7>("v_rsq_f64", out, in, ref, &ms);
    return 0;
}
