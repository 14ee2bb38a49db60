// tools/ubench/valu_issue.hip -- vector-instruction ISSUE rate of the integer / packed-16 / f64 operations the hot kernels are
// made of, at 1 / 2 / 4 / 8 wavefronts per SIMD on every CU of the chip.  One measured constant per instruction class replaces the
// three assumed ones bench.py carried in round 2 (one wave-instruction per 4 cycles per SIMD in two places, per 2 cycles in a third).
//
//     hipcc -O2 --offload-arch=gfx950 tools/ubench/valu_issue.hip -o tools/ubench/valu_issue && tools/ubench/valu_issue
//
// Every kernel runs ITERS x 16 INDEPENDENT instructions of one opcode (sixteen destination registers, inline asm so the
// compiler can neither fuse nor drop them).  Reported per (opcode, waves per SIMD):
//     G inst/s chip   wave-instructions per second over all 1024 SIMDs, from the host-timed launch (HIP events): THE figure of record
//     cyc@2.4/SIMD    the same as nominal 2.4 GHz cycles between two issues on one SIMD = SIMDs x 2.4e9 / (inst/s)
//     ticks/inst/SIMD s_memtime ticks of a wavefront / its instructions / wavefronts per SIMD (in-kernel view; the tick is not the
//                     shader cycle on every launch here -- it disagrees with the host clock by 1.0-2.0x -- so it is not used)
//     tick ratio      s_memtime ticks / s_memrealtime ticks x 100 MHz
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { OP_PK_MIN_I16, OP_PK_MAX_I16, OP_BCNT, OP_XOR, OP_DOT4_U8, OP_PERM, OP_ADD_U32, OP_FMA_F64, OP_MUL_F64, OP_ADD_F64, OP_FMA_F32, OP_MAD_U32_U24, OP_LSHL_OR, OP_COUNT };
static const char* OP_NAME[OP_COUNT] = { "v_pk_min_i16", "v_pk_max_i16", "v_bcnt_u32_b32", "v_xor_b32", "v_dot4_u32_u8", "v_perm_b32", "v_add_u32", "v_fma_f64", "v_mul_f64",
                                         "v_add_f64", "v_fma_f32", "v_mad_u32_u24", "v_lshl_or_b32" };

template <int OP>
__device__ __forceinline__ void one(unsigned& x, unsigned y, unsigned z)
{
    if constexpr (OP == OP_PK_MIN_I16) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_PK_MAX_I16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_DOT4_U8) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
    else if constexpr (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    else if constexpr (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    else if constexpr (OP == OP_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    else if constexpr (OP == OP_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
}
template <int OP>
__device__ __forceinline__ void one64(double& x, double y, double z)
{
    if constexpr (OP == OP_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
    else if constexpr (OP == OP_MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (OP == OP_ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
}

template <int OP>
__global__ void k_issue(unsigned* out, unsigned long long* ticks, int iters)
{
    constexpr bool F64 = OP == OP_FMA_F64 || OP == OP_MUL_F64 || OP == OP_ADD_F64;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long t0, t1, r0, r1;
    unsigned acc = 0;
    if constexpr (F64) {
        double r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = 1.0 + 1e-9 * (tid + i);
        const double y = 0.999999 + 1e-12 * tid, z = 1e-7;
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) one64<OP>(r[i], y, z);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        double s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += r[i];
        acc = (unsigned)__double2loint(s);
    } else {
        unsigned r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = tid * 2654435761u + i * 40503u;
        const unsigned y = tid * 97u + 0x01020304u, z = 0x07060504u ^ (tid & 3);
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) one<OP>(r[i], y, z);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= r[i];
    }
    out[tid] = acc;
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = tid >> 6;
        ticks[2 * w] = t1 - t0; ticks[2 * w + 1] = r1 - r0;
    }
}

typedef void (*kfn)(unsigned*, unsigned long long*, int);
static kfn KERNELS[OP_COUNT] = { k_issue<OP_PK_MIN_I16>, k_issue<OP_PK_MAX_I16>, k_issue<OP_BCNT>, k_issue<OP_XOR>, k_issue<OP_DOT4_U8>, k_issue<OP_PERM>, k_issue<OP_ADD_U32>,
                                 k_issue<OP_FMA_F64>, k_issue<OP_MUL_F64>, k_issue<OP_ADD_F64>, k_issue<OP_FMA_F32>, k_issue<OP_MAD_U32_U24>, k_issue<OP_LSHL_OR> };

int main(int argc, char** argv)
{
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
    const int ncu = pr.multiProcessorCount, nsimd = 4 * ncu;
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    printf("# %s, %d CUs, %d SIMDs; %d x 16 independent instructions per wavefront\n", pr.gcnArchName, ncu, nsimd, iters);
    printf("# %-16s %6s %14s %13s %16s %11s %10s\n", "opcode", "waves", "G inst/s chip", "cyc@2.4/SIMD", "ticks/inst/SIMD", "tick ratio", "launch us");
    const int maxwaves = nsimd * 8;
    unsigned* d_out; unsigned long long* d_ticks;
    CK(hipMalloc(&d_out, sizeof(unsigned) * 64 * maxwaves)); CK(hipMalloc(&d_ticks, sizeof(unsigned long long) * 2 * maxwaves));
    std::vector<unsigned long long> h(2 * maxwaves);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int op = 0; op < OP_COUNT; ++op)
        for (int w : { 1, 2, 4, 8 }) {
            // w wavefronts per SIMD: workgroups of 256 w threads (one per CU) up to w = 4, two workgroups of 1024 per CU at w = 8
            const int threads = 256 * std::min(w, 4), blocks = ncu * (w > 4 ? w / 4 : 1), nwaves = blocks * threads / 64;
            for (int rep = 0; rep < 3; ++rep) {                 // the last repetition is reported (clocks settled)
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(KERNELS[op], dim3(blocks), dim3(threads), 0, 0, d_out, d_ticks, iters);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d_ticks, sizeof(unsigned long long) * 2 * nwaves, hipMemcpyDeviceToHost));
            std::vector<double> cyc(nwaves), clk(nwaves);
            for (int i = 0; i < nwaves; ++i) { cyc[i] = (double)h[2 * i]; clk[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.0; }
            std::nth_element(cyc.begin(), cyc.begin() + nwaves / 2, cyc.end());
            std::nth_element(clk.begin(), clk.begin() + nwaves / 2, clk.end());
            const double inst = (double)iters * 16.0;
            const double ips = inst * nwaves / (ms * 1e-3);
            printf("  %-16s %6d %14.1f %13.2f %16.3f %11.3f %10.1f\n", OP_NAME[op], w, ips / 1e9, nsimd * 2.4e9 / ips, cyc[nwaves / 2] / inst / w, clk[nwaves / 2], ms * 1e3);
        }
    printf("# The guide's figure is 2 cycles per wave64 instruction per SIMD-32 once >= 2 wavefronts share the SIMD.  Read the 8-wave rows: the full-rate class\n"
           "# (v_fma_f32, v_add_u32, v_xor_b32) gets there; packed 16-bit min/max, v_bcnt, v_dot4, v_perm, three-operand integer ops and all f64 ops are half rate.\n"
           "# bench.py: VALU_ISSUE_PER_S = the v_fma_f32 8-wave row.\n");
    return 0;
}
