// Diagnostic: accuracy of v_rsq_f64 / v_rcp_f64 with 0, 1, 2 Newton steps on gfx950 (decides how many steps the
// factorisation's reciprocal square roots need).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y0 = __builtin_amdgcn_rsq(v);
    const double h = 0.5 * v;
    double e = fma(-h * y0, y0, 0.5);
    double y1 = fma(y0, e, y0);
    e = fma(-h * y1, y1, 0.5);
    double y2 = fma(y1, e, y1);
    out[i] = y0; out[n + i] = y1; out[2 * n + i] = y2;
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = fma(fma(-v, r0, 1.0), r0, r0);
    double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    out[3 * n + i] = r0; out[4 * n + i] = r1; out[5 * n + i] = r2;
}
int main()
{
    const int n = 1 << 20;
    double *hx = new double[n], *ho = new double[6 * n];
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); hx[i] = ldexp(1.0 + u, (int)(s % 200) - 100); }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(ho, dout, 6 * n * 8, hipMemcpyDeviceToHost);
    const char *names[6] = {"rsq", "rsq + 1 Newton", "rsq + 2 Newton", "rcp", "rcp + 1 Newton", "rcp + 2 Newton"};
    for (int c = 0; c < 6; c++) {
        long double mx = 0;
        for (int i = 0; i < n; i++) {
            long double ref = c < 3 ? 1.0L / sqrtl((long double)hx[i]) : 1.0L / (long double)hx[i];
            long double r = fabsl(((long double)ho[c * n + i] - ref) / ref);
            if (r > mx) mx = r;
        }
        printf("%-16s max relative error %.3Le  (%.2Lf ulp of double)\n", names[c], mx, mx / 1.1102230246251565e-16L);
    }
    return 0;
}
