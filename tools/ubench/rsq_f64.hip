// micro-benchmark: accuracy of v_rsq_f64 / v_rcp_f64 / v_sqrt_f64 and of one and two Newton steps on them (the pivot chain of the
// panel Cholesky carries `rsq + Newton` four times per 4-column block), and the dependent-issue latency of the chain's operations.
// hipcc --offload-arch=gfx950 -O3 -o rsq_f64 rsq_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k_acc(const double* x, int n, double* out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r0 = __builtin_amdgcn_rsq(v);
    double r1 = r0 * (1.5 - 0.5 * v * r0 * r0);
    double r2 = r1 * (1.5 - 0.5 * v * r1 * r1);
    // Newton in the fma form: e = 1 - v r^2 (one fma after t = v r), r' = r + r (e / 2)
    double t = v * r0, e = fma(-t, r0, 1.0), h = 0.5 * r0, rf = fma(h, e, r0);
    double c0 = __builtin_amdgcn_rcp(v);
    double c1 = c0 * (2.0 - v * c0);
    out[(size_t)i * 6 + 0] = r0; out[(size_t)i * 6 + 1] = r1; out[(size_t)i * 6 + 2] = r2; out[(size_t)i * 6 + 3] = rf; out[(size_t)i * 6 + 4] = c0; out[(size_t)i * 6 + 5] = c1;
}
template <int MODE>
__global__ void k_lat(double* out, unsigned long long* cyc, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-6, b = 0.999999;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) a = fma(a, b, 1e-9);
            else if (MODE == 1) a = a * b;
            else if (MODE == 2) a = __builtin_amdgcn_rsq(a) + 0.5;
            else if (MODE == 3) a = __builtin_amdgcn_rcp(a) + 0.5;
            else if (MODE == 4) { float f = __builtin_amdgcn_rsqf((float)a); a = (double)f + 0.5; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; const double u = (s >> 11) * (1.0 / 9007199254740992.0); x[i] = std::exp((u - 0.5) * 60.0); }
    double *dx, *dout; unsigned long long* dc;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, (size_t)n * 6 * 8); hipMalloc(&dc, 64);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_acc, dim3(n / 256), dim3(256), 0, 0, dx, n, dout);
    std::vector<double> o((size_t)n * 6);
    hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
    double m[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < n; ++i) {
        const long double rs = 1.0L / sqrtl((long double)x[i]), rc = 1.0L / (long double)x[i];
        for (int k = 0; k < 4; ++k) m[k] = fmax(m[k], (double)fabsl((o[(size_t)i * 6 + k] - rs) / rs));
        for (int k = 4; k < 6; ++k) m[k] = fmax(m[k], (double)fabsl((o[(size_t)i * 6 + k] - rc) / rc));
    }
    printf("max relative error over %d values in [e^-30, e^30] (2^-52 = %.3g):\n  v_rsq_f64 %.3g | + 1 Newton %.3g | + 2 Newton %.3g | + 1 Newton (fma form) %.3g\n  v_rcp_f64 %.3g | + 1 Newton %.3g\n",
           n, ldexp(1.0, -52), m[0], m[1], m[2], m[3], m[4], m[5]);
    const int iters = 4096;
    hipLaunchKernelGGL(k_lat<0>, dim3(1), dim3(64), 0, 0, dout, dc, iters);
    hipLaunchKernelGGL(k_lat<1>, dim3(1), dim3(64), 0, 0, dout, dc, iters);
    hipLaunchKernelGGL(k_lat<2>, dim3(1), dim3(64), 0, 0, dout, dc, iters);
    hipLaunchKernelGGL(k_lat<3>, dim3(1), dim3(64), 0, 0, dout, dc, iters);
    hipLaunchKernelGGL(k_lat<4>, dim3(1), dim3(64), 0, 0, dout, dc, iters);
    unsigned long long c[8];
    hipMemcpy(c, dc, 64, hipMemcpyDeviceToHost);
    const double per = 1.0 / (iters * 16.0);
    printf("dependent chain, one wavefront, s_memtime ticks (100 MHz: x 24 = shader cycles at 2.4 GHz) per operation:\n  v_fma_f64 %.2f | v_mul_f64 %.2f | v_rsq_f64 + add %.2f | v_rcp_f64 + add %.2f | cvt + v_rsq_f32 + cvt + add %.2f\n",
           c[0] * per, c[1] * per, c[2] * per, c[3] * per, c[4] * per);
    return 0;
}
