// Diagnostic: how many workgroups of a given shape (threads, dynamic LDS, vector registers) does the chip hold at once?  Every workgroup stamps its
// start (100 MHz wall clock) and spins for 1 ms; workgroups that start within 100 us of the first are resident together.
//   hipcc --offload-arch=gfx950 -O2 -o occupancy_probe tools/microbench/occupancy_probe.hip && ./occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
extern __shared__ unsigned char smem[];
template <int VGPRS>
__global__ void probe(long long *start, long long spin_ticks)
{
    if (VGPRS == 168) asm volatile("" ::: "v167");
    if (VGPRS == 128) asm volatile("" ::: "v127");
    if (VGPRS == 96) asm volatile("" ::: "v95");
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) { start[blockIdx.x] = t0; smem[0] = 1; }
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(32);
}
template <int VGPRS>
static void run(int threads, int lds, int grid)
{
    long long *d; hipMalloc(&d, sizeof(long long) * grid);
    hipFuncSetAttribute((const void *)probe<VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(probe<VGPRS>, dim3(grid), dim3(threads), lds, 0, d, 100000ll); hipDeviceSynchronize(); }
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), d, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    const long long t0 = *std::min_element(h.begin(), h.end());
    int first = 0, second = 0;
    for (long long v : h) { if (v - t0 < 10000) first++; else if (v - t0 < 110000) second++; }
    printf("threads %4d lds %6d vgprs %3d grid %5d: resident together %5d (per CU %.2f), next round %d\n", threads, lds, VGPRS, grid, first, first / 256.0, second);
    hipFree(d);
}
int main()
{
    run<168>(192, 36144, 4096); run<168>(192, 35104, 4096); run<168>(192, 16384, 4096); run<168>(256, 36144, 4096); run<168>(128, 36144, 4096); run<168>(64, 36144, 4096);
    run<128>(192, 36144, 4096); run<128>(256, 36144, 4096); run<96>(192, 30000, 4096); run<96>(256, 30000, 4096); run<168>(192, 1024, 4096); run<168>(384, 72000, 4096);
    return 0;
}
