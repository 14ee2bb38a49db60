// D2H of 38.4 MB (the poses of C3) into page-locked memory: the runtime's copy against store kernels.
// hipcc --offload-arch=gfx950 -O3 -o d2h d2h.hip && ./d2h
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
template <bool NT, int U>
__global__ __launch_bounds__(256) void store_kernel(const double2* __restrict__ src, double2* __restrict__ dst, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride * U) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n2) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n2) {
            if (NT) { __builtin_nontemporal_store(v[u].x, &dst[i + u * stride].x); __builtin_nontemporal_store(v[u].y, &dst[i + u * stride].y); }
            else dst[i + u * stride] = v[u];
        }
    }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 400000ull * 96, n2 = bytes / 16;
    void *d, *h; hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipHostMalloc(&h, bytes, hipHostMallocDefault); memset(h, 0, bytes);
    hipStream_t st; hipStreamCreate(&st);
    auto timeit = [&](const char* name, auto&& fn) {
        std::vector<double> t;
        for (int r = 0; r < 12; ++r) { hipStreamSynchronize(st); const double t0 = now(); fn(); hipStreamSynchronize(st); t.push_back(now() - t0); }
        std::sort(t.begin(), t.end());
        printf("%-34s median %8.1f us  min %8.1f us  (%.1f GB/s)\n", name, t[t.size() / 2], t[0], bytes / t[t.size() / 2] / 1e3);
    };
    timeit("hipMemcpyAsync", [&] { hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st); });
    for (int grid : {16, 64, 256, 1024, 4096}) {
        char nm[64];
        snprintf(nm, 64, "kernel nt   U1 grid %d", grid); timeit(nm, [&] { hipLaunchKernelGGL((store_kernel<true, 1>), dim3(grid), dim3(256), 0, st, (const double2*)d, (double2*)h, n2); });
        snprintf(nm, 64, "kernel plain U1 grid %d", grid); timeit(nm, [&] { hipLaunchKernelGGL((store_kernel<false, 1>), dim3(grid), dim3(256), 0, st, (const double2*)d, (double2*)h, n2); });
        snprintf(nm, 64, "kernel plain U4 grid %d", grid); timeit(nm, [&] { hipLaunchKernelGGL((store_kernel<false, 4>), dim3(grid), dim3(256), 0, st, (const double2*)d, (double2*)h, n2); });
    }
    // two halves: copy engine + kernel side by side
    hipStream_t s2; hipStreamCreate(&s2);
    timeit("memcpy half + kernel half", [&] { hipMemcpyAsync(h, d, bytes / 2, hipMemcpyDeviceToHost, s2);
        hipLaunchKernelGGL((store_kernel<false, 1>), dim3(256), dim3(256), 0, st, (const double2*)d + n2 / 2, (double2*)h + n2 / 2, n2 / 2); hipStreamSynchronize(s2); });
    return 0;
}
