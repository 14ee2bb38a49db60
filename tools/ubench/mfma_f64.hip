// micro-benchmark: cycles per v_mfma_f64_16x16x4_f64 for one wavefront (independent accumulators, one dependent chain), and
// per-instruction issue cost of the f64 VALU operations the panel kernels' pivot chain is made of.  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma_indep(double* out, unsigned long long* cyc, int iters)
{
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = { 0.0, 0.0, 0.0, 0.0 };
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mfma_dep(double* out, unsigned long long* cyc, int iters)
{
    d4 acc = { 0.0, 0.0, 0.0, 0.0 };
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc[0] + acc[3];
    if (threadIdx.x == 0) cyc[1] = t1 - t0;
}
// result of one MFMA feeds the B operand of the next (the LP -> update pattern)
__global__ void k_mfma_opdep(double* out, unsigned long long* cyc, int iters)
{
    d4 acc = { 0.0, 0.0, 0.0, 0.0 };
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const d4 z = { 0.0, 0.0, 0.0, 0.0 };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, z, 0, 0, 0); b = r[0] * 1e-3 + 1.0; }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = b + acc[0];
    if (threadIdx.x == 0) cyc[2] = t1 - t0;
}
__global__ void k_fma_dep(double* out, unsigned long long* cyc, int iters)
{
    double x = 1.0 + threadIdx.x * 1e-6, y = 0.999;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) x = fma(x, y, 1e-9);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[3] = t1 - t0;
}
__global__ void k_rsq_dep(double* out, unsigned long long* cyc, int iters)
{
    double x = 2.0 + threadIdx.x * 1e-6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) x = __builtin_amdgcn_rsq(x) + 1.5;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[4] = t1 - t0;
}
__global__ void k_readlane(double* out, unsigned long long* cyc, int iters)
{
    double x = 2.0 + threadIdx.x * 1e-6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int lo = __builtin_amdgcn_readlane(__double2loint(x), i + 3), hi = __builtin_amdgcn_readlane(__double2hiint(x), i + 3); x = x * 0.5 + __hiloint2double(hi, lo) * 0.25; }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[5] = t1 - t0;
}
int main()
{
    double* out; unsigned long long* cyc; unsigned long long h[8] = { 0 };
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64);
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        k_mfma_indep<<<1, 64>>>(out, cyc, iters); k_mfma_dep<<<1, 64>>>(out, cyc, iters); k_mfma_opdep<<<1, 64>>>(out, cyc, iters);
        k_fma_dep<<<1, 64>>>(out, cyc, iters); k_rsq_dep<<<1, 64>>>(out, cyc, iters); k_readlane<<<1, 64>>>(out, cyc, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double n = iters * 8.0;
    printf("cycles per op (one wavefront): mfma_f64_16x16x4 independent %.1f  same accumulator %.1f  result->operand (mfma + v_accvgpr_read + fma) %.1f | dependent v_fma_f64 %.1f  v_rsq_f64+add %.1f  readlane pair + 2 fma %.1f\n",
           h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n);
    return 0;
}
