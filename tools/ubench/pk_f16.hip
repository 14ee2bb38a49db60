// tools/ubench/pk_f16.hip -- are the packed f16 min / max instructions of gfx950 (two- and THREE-operand: v_pk_minimum3_f16 /
// v_pk_maximum3_f16 are new on this chip) usable for the integer arc minima / maxima of FAST, and at what rate do they issue?
//   (1) semantics: bytes 0..255 held as the f16 DENORMALS n x 2^-24 (bit pattern 0x00nn: what v_perm_b32 delivers today) and as the
//       normals 1024 + n (0x6400 | n): is min / max of the bit patterns the integer min / max, bit for bit?
//   (2) issue rate at 8 wavefronts per SIMD next to v_pk_min_i16 (half rate) and v_add_u32 (full rate).
//     hipcc -O2 --offload-arch=gfx950 tools/ubench/pk_f16.hip -o tools/ubench/pk_f16 && tools/ubench/pk_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { OP_PK_MIN_I16, OP_PK_MIN_F16, OP_PK_MAX_F16, OP_PK_MIN3_F16, OP_PK_MAX3_F16, OP_MIN3_I16, OP_PK_MIN_U16, OP_ADD_U32, OP_OR, OP_COUNT };
static const char* NAME[OP_COUNT] = { "v_pk_min_i16", "v_pk_min_f16", "v_pk_max_f16", "v_pk_minimum3_f16", "v_pk_maximum3_f16", "v_min3_i16", "v_pk_min_u16", "v_add_u32", "v_or_b32" };
template <int OP> __device__ __forceinline__ void one(unsigned& x, unsigned y, unsigned z)
{
    if constexpr (OP == OP_PK_MIN_I16) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_PK_MIN_F16) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_PK_MAX_F16) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_PK_MIN3_F16) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    else if constexpr (OP == OP_PK_MAX3_F16) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    else if constexpr (OP == OP_MIN3_I16) asm volatile("v_min3_i16 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    else if constexpr (OP == OP_PK_MIN_U16) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(y));
}
template <int OP> __global__ void k_issue(unsigned* out, int iters)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = ((tid * 7u + i) & 255u) | ((((tid >> 3) + 13u * i) & 255u) << 16);      // denormal halves: no NaN / Inf patterns in flight
    const unsigned y = (tid & 255u) | (((tid >> 8) & 255u) << 16), z = ((tid * 3u) & 255u) | (((tid * 5u) & 255u) << 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) one<OP>(r[i], y, z);
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= r[i];
    out[tid] = acc;
}
// semantics: every (a, b, c) of 0..255 in both halves, denormal and biased encodings
__global__ void k_check(unsigned* bad)
{
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;       // 2^24 triples
    const unsigned a = t & 255u, b = (t >> 8) & 255u, c = (t >> 16) & 255u;
    const unsigned imin2 = a < b ? a : b, imax2 = a > b ? a : b, imin3 = imin2 < c ? imin2 : c, imax3 = imax2 > c ? imax2 : c;
    for (int enc = 0; enc < 2; ++enc) {
        const unsigned bias = enc ? 0x64006400u : 0u;
        // different values in the two halves: (a, b, c) low, (c, a, b) high
        const unsigned A = (a | (c << 16)) | bias, B = (b | (a << 16)) | bias, Cc = (c | (b << 16)) | bias;
        unsigned r;
        const unsigned hmin2 = c < a ? c : a, hmax2 = c > a ? c : a, hmin3 = hmin2 < b ? hmin2 : b, hmax3 = hmax2 > b ? hmax2 : b;
        asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(A), "v"(B)); if (r != ((imin2 | (hmin2 << 16)) | bias)) atomicAdd(&bad[enc * 4 + 0], 1u);
        asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(A), "v"(B)); if (r != ((imax2 | (hmax2 << 16)) | bias)) atomicAdd(&bad[enc * 4 + 1], 1u);
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(A), "v"(B), "v"(Cc)); if (r != ((imin3 | (hmin3 << 16)) | bias)) atomicAdd(&bad[enc * 4 + 2], 1u);
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(A), "v"(B), "v"(Cc)); if (r != ((imax3 | (hmax3 << 16)) | bias)) atomicAdd(&bad[enc * 4 + 3], 1u);
    }
}
typedef void (*kfn)(unsigned*, int);
static kfn K[OP_COUNT] = { k_issue<OP_PK_MIN_I16>, k_issue<OP_PK_MIN_F16>, k_issue<OP_PK_MAX_F16>, k_issue<OP_PK_MIN3_F16>, k_issue<OP_PK_MAX3_F16>, k_issue<OP_MIN3_I16>, k_issue<OP_PK_MIN_U16>, k_issue<OP_ADD_U32>, k_issue<OP_OR> };
int main()
{
    CK(hipSetDevice(0));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount, nsimd = 4 * ncu, iters = 4096;
    unsigned* d_bad; CK(hipMalloc(&d_bad, 8 * sizeof(unsigned))); CK(hipMemset(d_bad, 0, 8 * sizeof(unsigned)));
    hipLaunchKernelGGL(k_check, dim3((1u << 24) / 256), dim3(256), 0, 0, d_bad);
    unsigned h_bad[8]; CK(hipMemcpy(h_bad, d_bad, sizeof h_bad, hipMemcpyDeviceToHost));
    const char* on[4] = { "v_pk_min_f16", "v_pk_max_f16", "v_pk_minimum3_f16", "v_pk_maximum3_f16" };
    for (int enc = 0; enc < 2; ++enc) for (int o = 0; o < 4; ++o)
        printf("semantics %-18s %-28s mismatches of 16 777 216 triples: %u\n", on[o], enc ? "normals 1024 + n (0x6400|n)" : "denormals n x 2^-24 (0x00nn)", h_bad[enc * 4 + o]);
    unsigned* d_out; CK(hipMalloc(&d_out, sizeof(unsigned) * 64 * nsimd * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# %s, %d CUs; %d x 16 independent instructions per wavefront, 8 wavefronts per SIMD\n", pr.gcnArchName, ncu, iters);
    for (int op = 0; op < OP_COUNT; ++op) {
        const int threads = 1024, blocks = ncu * 2, nwaves = blocks * threads / 64;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(K[op], dim3(blocks), dim3(threads), 0, 0, d_out, iters); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
        const double ips = (double)iters * 16.0 * nwaves / (ms * 1e-3);
        printf("  %-20s %8.1f G wave-instructions/s chip-wide   %5.2f cycles @2.4 GHz per issue per SIMD\n", NAME[op], ips / 1e9, nsimd * 2.4e9 / ips);
    }
    return 0;
}
