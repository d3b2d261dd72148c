// Microbenchmark (diagnostic, not part of the product): LDS operation rates on gfx950 for the access patterns of the
// solve kernel -- scattered f64 / f32 atomic adds, scattered 8-byte reads.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int LDSN = 4096;   // doubles
template <int MODE, int WG> __global__ __launch_bounds__(WG) void k(double *out, int iters, int spread)
{
    __shared__ double buf[LDSN];
    for (int i = threadIdx.x; i < LDSN; i += WG) buf[i] = 0.0;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
    double acc = 0.0;
    float *fb = reinterpret_cast<float *>(buf);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            s = s * 1664525u + 1013904223u;
            const int a = (s >> 8) & (spread - 1);
            if (MODE == 0) atomicAdd(&buf[a], 1.0);                     // ds_add_f64
            else if (MODE == 1) atomicAdd(&fb[a], 1.0f);                // ds_add_f32
            else if (MODE == 2) acc += buf[a];                           // ds_read_b64
            else if (MODE == 3) acc += fb[a];                            // ds_read_b32
            else if (MODE == 4) atomicAdd(&buf[(threadIdx.x + u * 64 + it) & (spread - 1)], 1.0);   // conflict-free f64 atomics
            else if (MODE == 5) { acc += (double)a; }                         // index generation only
            else if (MODE == 6) buf[a] = acc;                            // ds_write_b64 scattered
            else if (MODE == 7) atomicAdd(reinterpret_cast<unsigned long long *>(buf) + a, (unsigned long long)s);   // ds_add_u64
            else if (MODE == 8) atomicAdd(reinterpret_cast<unsigned *>(buf) + a, s);                                  // ds_add_u32
            else if (MODE == 9) { atomicAdd(&buf[a], 1.0); atomicAdd(&buf[a ^ 1], 1.0); }                             // pairs of neighbours
        }
    }
    __syncthreads();
    if (MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) out[blockIdx.x * WG + threadIdx.x] = acc + buf[threadIdx.x];
    else out[blockIdx.x * WG + threadIdx.x] = buf[threadIdx.x];
}
template <int MODE, int WG> int run(const char *name, int grid, int spread)
{
    double *out; CHECK(hipMalloc(&out, sizeof(double) * grid * WG));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int iters = 2000;
    k<MODE, WG><<<grid, WG>>>(out, 10, spread);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<MODE, WG><<<grid, WG>>>(out, iters, spread);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr_per_cu = (double)iters * 16 * (WG / 64) * (grid / 256.0);
    printf("%-28s WG %4d grid %5d spread %5d: %8.3f ms  -> %7.1f clk per wave-instruction per CU (2.4 GHz)\n", name, WG, grid, spread, ms,
           ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
    CHECK(hipFree(out));
    return 0;
}
int main()
{
    for (int spread : {4096, 256, 32}) {
        run<0, 256>("ds_add_f64 scattered", 256, spread);
        run<1, 256>("ds_add_f32 scattered", 256, spread);
        run<2, 256>("ds_read_b64 scattered", 256, spread);
        run<3, 256>("ds_read_b32 scattered", 256, spread);
        run<6, 256>("ds_write_b64 scattered", 256, spread);
        run<7, 256>("ds_add_u64 scattered", 256, spread);
        run<8, 256>("ds_add_u32 scattered", 256, spread);
        run<9, 256>("2x ds_add_f64 neighbours", 256, spread);
    }
    run<4, 256>("ds_add_f64 conflict-free", 256, 4096);
    run<5, 256>("index generation only", 256, 4096);
    run<0, 256>("ds_add_f64 scattered 2WG/CU", 512, 4096);
    run<0, 512>("ds_add_f64 scattered WG512", 256, 4096);
    run<0, 64>("ds_add_f64 scattered WG64", 256, 4096);
    run<1, 64>("ds_add_f32 scattered WG64", 256, 4096);
    run<2, 64>("ds_read_b64 scattered WG64", 256, 4096);
    return 0;
}
