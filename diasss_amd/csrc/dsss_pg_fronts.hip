// diasss_amd/csrc/dsss_pg_fronts.hip -- the multifrontal TOP of the reduced pose-graph system: dense fronts on the f64 matrix cores.
// Compared with the oracle at 1e-6 on the poses, not bit for bit: multiply-adds may fuse here.
#include "dsss_pg_kernels.h"
#include "dsss_pg_dev.h"
#include <utility>
#pragma clang fp contract(fast)
// ---- multifrontal top of the tree (dsss_pg_sym.h).  A front is a dense ld x ld lower-triangular image
//          [ F11            ]   s6 own scalar columns          assembled from the original entries + the update matrices of its
//          [ F21   F22      ]   n6 - s6 boundary rows          children (extend-add), factorised in 96-column panel steps:
//     pg_front_asm_kernel      zero + original entries + children, parent rows owned by workgroups, children in fixed order
//     pg_front_diag4_kernel    L11 = chol(A11) in 4-column pivot blocks, their inverses Linv, y = L11^-1 b      one workgroup per panel
//     pg_front_trsm2_kernel    L21 = A21 L11^-T, b2 -= L21 y                              one wavefront per 16 rows
//     pg_front_syrk_kernel     A22 -= L21 L21^T                                           64 x 64 tiles of the trailing part
//     pg_front_bwd2_kernel     x1 = L11^-T (y1 - L21^T x2)
// all dense products on v_mfma_f64_16x16x4_f64.  What is left in F22 after the last panel is the front's update matrix.

// one workgroup per (front, block row R): zero the row up to its diagonal block, add the original entries of the row, then
// the child rows that map onto it, children in their fixed order (the list of contributing (child, row) pairs comes from the
// host: no searching on the device).  A thread owns a column of the child row and moves its six scalars at once.
__global__ __launch_bounds__(256) void pg_front_asm_kernel(const int* __restrict__ it_front, const int* __restrict__ it_row, const pg_front* __restrict__ FD,
                                                           const pg_child* __restrict__ CH, const int* __restrict__ rel, const int* __restrict__ xr_ptr,
                                                           const int* __restrict__ xr_child, const int* __restrict__ xr_row, const int* __restrict__ fa_rowptr,
                                                           const int* __restrict__ fa_src, const int* __restrict__ fa_col, const int* __restrict__ fa_tr,
                                                           const double* __restrict__ aval, const double* __restrict__ x, double* __restrict__ F, double* __restrict__ R)
{
    const pg_front fd = FD[it_front[blockIdx.x]];
    const int Rb = it_row[blockIdx.x], ld = fd.ld;
    double* __restrict__ A = F + fd.off + (size_t)(6 * Rb) * ld; double* __restrict__ r = R + fd.roff + 6 * Rb;
    const int ncol = 6 * (Rb + 1);
    for (int cc = threadIdx.x; cc < ncol; cc += 256) {
#pragma unroll
        for (int a = 0; a < 6; ++a) A[(size_t)a * ld + cc] = 0.0;
    }
    if (threadIdx.x < 6) { const int i = 6 * Rb + threadIdx.x; r[threadIdx.x] = i < fd.s6 ? x[(size_t)fd.c0 * 6 + i] : 0.0; }
    __syncthreads();
    {   // original entries; entries with the same destination block are consecutive and are summed by one thread group in list order
        const int lo = fa_rowptr[fd.rowptr + Rb], hi = fa_rowptr[fd.rowptr + Rb + 1];
        const int grp = threadIdx.x / 36, el = threadIdx.x % 36;
        if (grp < 7)
            for (int e = lo + grp; e < hi; e += 7) {
                if (e > lo && fa_col[e - 1] == fa_col[e]) continue;      // not the head of its run
                const int a = el / 6, b = el % 6;
                double v = 0;
                for (int e2 = e; e2 < hi && fa_col[e2] == fa_col[e]; ++e2) v += aval[(size_t)fa_src[e2] * 36 + (fa_tr[e2] ? b * 6 + a : a * 6 + b)];
                A[(size_t)a * ld + fa_col[e] * 6 + b] += v;
            }
    }
    __syncthreads();
    // the record of the NEXT (child, row) pair is fetched while the current one is added: list entry -> child descriptor -> its row is a chain
    // of dependent round trips (in-kernel stamps: 2.7 us per pair, four round trips), and the pairs of a parent row must stay in order
    const int q_lo = xr_ptr[fd.rowptr + Rb], q_hi = xr_ptr[fd.rowptr + Rb + 1];
    pg_child cd_n = {}; int i_n = 0;
    if (q_lo < q_hi) { cd_n = CH[xr_child[q_lo]]; i_n = xr_row[q_lo]; }
    for (int q = q_lo; q < q_hi; ++q) {
        const pg_child cd = cd_n;
        const int i = i_n, wcols = 6 * (i + 1);
        if (q + 1 < q_hi) { cd_n = CH[xr_child[q + 1]]; i_n = xr_row[q + 1]; }
        const int* __restrict__ rl = rel + cd.relptr;
        const double* __restrict__ src = cd.U + (size_t)(6 * i) * cd.cld;
        for (int cc = threadIdx.x; cc < wcols; cc += 256) {
            const int j2 = cc / 6, dcol = 6 * rl[j2] + (cc - 6 * j2);
            double u[6], d[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) { u[a] = src[(size_t)a * cd.cld + cc]; d[a] = A[(size_t)a * ld + dcol]; }
#pragma unroll
            for (int a = 0; a < 6; ++a) A[(size_t)a * ld + dcol] = d[a] + u[a];
        }
        if (threadIdx.x < 6) r[threadIdx.x] += cd.g[6 * i + threadIdx.x];
        __syncthreads();
    }
}

typedef double pg_d4 __attribute__((ext_vector_type(4)));
__device__ inline double pg_readlane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// A22 -= L21 L21^T on the trailing part of the front (rows and columns beyond the panel): one workgroup per 64 x 64 tile of
// the lower triangle (exact tile list from the host), one wavefront per 16 rows of the tile.  The 64 rows of L21 that form the
// tile's COLUMNS are staged once in LDS (coalesced 16-byte loads, conflict-free row stride) and serve all four wavefronts
// as MFMA B operands; every wavefront keeps its own 16 x 96 slab of L21 in 24 A-operand registers.  K = the panel's 96
// columns: 24 v_mfma_f64_16x16x4_f64 per 16 x 16 block.
#define PG_SYRK_LD 98
__global__ __launch_bounds__(256) void pg_front_syrk_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD,
                                                            const int* __restrict__ tile_item, const int* __restrict__ tile_ij, double* __restrict__ F)
{
    __shared__ double sB[64 * PG_SYRK_LD];
    const int item = tile_item[blockIdx.x], ij = tile_ij[blockIdx.x], ti = ij >> 16, tj = ij & 0xffff;
    const pg_front fd = FD[it_front[item]];
    const int step = it_step[item], col0 = 96 * step;
    const int n = min(96, fd.s6 - col0), ld = fd.ld;
    const int row0 = col0 + n, nrows = fd.n6 - row0;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    double* __restrict__ A = F + fd.off;
    {   // rows 64 tj .. 64 tj + 63 of L21 -> LDS, 48 x 16 bytes per row (ld and col0 are multiples of 16 scalars, rows 16-byte aligned)
        const double2* __restrict__ src = reinterpret_cast<const double2*>(A + (size_t)(row0 + 64 * tj) * ld + col0);
        const int ld2 = ld >> 1, n2 = n >> 1, rows_here = min(64, nrows - 64 * tj);
        double2 v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int e = threadIdx.x + 256 * u, rr = e / 48, c2 = e - 48 * rr;
            v[u] = (rr < rows_here && c2 < n2) ? src[(size_t)rr * ld2 + c2] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int e = threadIdx.x + 256 * u, rr = e / 48, c2 = e - 48 * rr;
            *reinterpret_cast<double2*>(&sB[rr * PG_SYRK_LD + 2 * c2]) = v[u];
        }
    }
    const int i0 = 64 * ti + 16 * wave;
    const int ir = i0 + (l & 15);
    double a[24];
    pg_d4 acc[4];
    {
        const double* __restrict__ Ai = A + (size_t)(row0 + min(ir, nrows - 1)) * ld + col0;
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) { const int k = 4 * ks + (l >> 4); a[ks] = (ir < nrows && k < n) ? Ai[k] : 0.0; }      // (the sign further down: negated inside the conditional, every one of the 24 loads waited for its own round trip -- s_waitcnt vmcnt(0) after each)
        // the four 16 x 16 blocks of C this wavefront updates come in with the operands: one round trip to memory, not five
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j0 = 64 * tj + 16 * c, jr = j0 + (l & 15);
            const double* __restrict__ Cp = A + (size_t)(row0 + i0 + (l >> 4)) * ld + row0 + j0 + (l & 15);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[c][v] = (i0 < nrows && j0 <= i0 + 15 && i0 + (l >> 4) + 4 * v < nrows && jr < nrows) ? Cp[(size_t)(4 * v) * ld] : 0.0;
        }
    }
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) a[ks] = -a[ks];
    __syncthreads();
    if (i0 >= nrows) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j0 = 64 * tj + 16 * c;
        if (j0 >= nrows || j0 > i0 + 15) break;              // beyond the front, or entirely above the diagonal (uniform per wavefront)
        const int jr = j0 + (l & 15);
        const double* __restrict__ sb = sB + (16 * c + (l & 15)) * PG_SYRK_LD + (l >> 4);
        double* __restrict__ Cp = A + (size_t)(row0 + i0 + (l >> 4)) * ld + row0 + j0 + (l & 15);
        pg_d4 r = acc[c];
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], sb[4 * ks], r, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) if (i0 + (l >> 4) + 4 * v < nrows && jr < nrows) Cp[(size_t)(4 * v) * ld] = r[v];
    }
}

// ---- panel kernels without an explicit inverse: the 96 x 96 panel lives in registers as 16 x 16
// MFMA accumulator tiles and is factorised RIGHT-LOOKING IN BLOCKS OF FOUR COLUMNS, every rank-4 update being one
// v_mfma_f64_16x16x4_f64 per tile (K = 4 is exactly one instruction).  Tile (T, I), T <= I, holds the symmetric matrix
// TRANSPOSED: D[i][j] = M[16 I + j][16 T + i], i.e. lane (j = l & 15, q = l >> 4), register v holds M[16 I + j][16 T + q + 4 v].
// In that layout the four pivot columns p_k = 16 t + 4 b + k of tile row I are register b of lanes (j, k): exactly the MFMA
// operand layout (A: [i][k] on lane (i, k); B: [k][j] on lane (j, k)), so no data moves between the pivot solve and the update:
//     M44 (4 x 4 pivot block)  -> 10 v_readlane pairs -> Cholesky + inverse Linv, the same scalars on every lane
//     LP_I = Linv x P_I^T                                 one MFMA per tile (P = register b of tile (t, I)), result in operand layout
//     tile (T', I') -= LP_T' LP_I'^T                      one MFMA per tile, A = -LP_T', B = LP_I'
// No explicit L11^-1: the kernels below the panel (row solve, back-substitution) repeat the same 4-column
// steps with the stored Linv blocks.  The right-hand side rides along as row 96 of the matrix (tile row 6): the Cholesky factor
// of the augmented matrix carries y = L11^-1 b in that row.
__device__ inline double pg_rsqrt(double x)
{
    double r = __builtin_amdgcn_rsq(x);            // v_rsq_f64 seed, two Newton steps
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
// Cholesky of the 4 x 4 block m (lower, row-major 10 values m00 m10 m11 m20 m21 m22 m30 m31 m32 m33) and the inverse of its
// factor: li = [i00 i10 i11 i20 i21 i22 i30 i31 i32 i33]
__device__ inline int pg_chol4_inv(const double* m, double* li)
{
    int bad = 0;
    double d0 = m[0]; if (!(d0 > 0) || !isfinite(d0)) { bad = 1; d0 = 1.0; }
    const double r0 = pg_rsqrt(d0);
    const double l10 = m[1] * r0, l20 = m[3] * r0, l30 = m[6] * r0;
    double d1 = m[2] - l10 * l10; if (!(d1 > 0) || !isfinite(d1)) { bad = 1; d1 = 1.0; }
    const double r1 = pg_rsqrt(d1);
    const double l21 = (m[4] - l20 * l10) * r1, l31 = (m[7] - l30 * l10) * r1;
    double d2 = m[5] - l20 * l20 - l21 * l21; if (!(d2 > 0) || !isfinite(d2)) { bad = 1; d2 = 1.0; }
    const double r2 = pg_rsqrt(d2);
    const double l32 = (m[8] - l30 * l20 - l31 * l21) * r2;
    double d3 = m[9] - l30 * l30 - l31 * l31 - l32 * l32; if (!(d3 > 0) || !isfinite(d3)) { bad = 1; d3 = 1.0; }
    const double r3 = pg_rsqrt(d3);
    li[0] = r0; li[2] = r1; li[5] = r2; li[9] = r3;
    li[1] = -(l10 * r0) * r1;
    li[3] = -(l20 * r0 + l21 * li[1]) * r2; li[4] = -(l21 * r1) * r2;
    li[6] = -(l30 * r0 + l31 * li[1] + l32 * li[3]) * r3; li[7] = -(l31 * r1 + l32 * li[4]) * r3; li[8] = -(l32 * r2) * r3;
    return bad;
}
// workgroup barrier that orders LDS traffic only (__syncthreads() also waits for the global stores in flight)
#define PG_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// ---- The panel factorisation on FOUR wavefronts (tile column I = tile row I of the matrix belongs to wavefront I mod 4: at most nine tiles
// and a quarter of the updates each), as a PIPELINE.  The plain form (rounds 1 - 4, removed in round 5: per 4-column block the pivot
// wavefront's Cholesky + inverse -> barrier -> LP of every tile column -> barrier -> updates) costs two workgroup barriers
// and two LDS round trips on the one chain that matters: pivot block -> Cholesky + inverse -> LP of the pivot tile -> update of the
// pivot tile -> next pivot block (1 950 cycles per block, 870 of them the Cholesky).  Here the owner of the pivot tile runs that
// chain through its own registers (the tile register that comes out of the matrix core IS both operands of the pivot tile's update)
// and the other work trails behind it, one barrier per block:
//     region r (between barriers r - 1 and r)
//         every wavefront   U(r - 2): updates of block r - 2 on its tiles, operands from the LDS ring (3 buffers)
//                           L(r - 1): LP of block r - 1 for its tile columns, Linv(r - 1) from LDS (2 buffers) -> ring
//         pivot wavefront   C(r): Linv(r), LP of the pivot tile, update of the pivot tile -- registers only; Linv and LP -> LDS
// When the pivot tile changes, its new owner catches up on the one update it trails by (again from its own registers).
// U and L of a region do not feed C, so the compiler is free to fill the Cholesky's dependency bubbles with their matrix-core
// work.  The tile count NT = ceil(n / 16) is a template parameter: straight-line code, no runtime guards inside the pipeline.
// Every tile receives the same updates in the same order as in the two-barrier form: the result was bit-identical to it.
struct pg_d4_lds { double lp[3][7][64]; double li[2][16]; int bad; };
// The trailing work of region R on wavefront W, as a compile-time list of matrix-core operations: kind 1 = update U(R - 2) of tile
// (T2, I), kind 2 = LP of block R - 1 for tile column I (with the catch-up update when I becomes the pivot tile), 0 = end of list.
struct pg_lag_desc { int kind, I, T2; };
constexpr pg_lag_desc pg_lag_get(int W, int NT, int R, int want)
{
    const int I0 = W, I1 = W + 4, K = 4 * NT;
    const bool act0 = I0 < NT, act1 = I1 < 7 && (I1 < NT || I1 == 6);
    int idx = 0;
    if (R >= 2) {
        const int k = R - 2, t = k / 4, b = k % 4;
        const bool piv = (t & 3) == W;                                  // this wavefront ran the critical part of block k
        for (int c = 0; c < 2; ++c) {
            const int I = c ? I1 : I0;
            if (!(c ? act1 : act0) || I < t) continue;
            for (int T2 = t; T2 < 6; ++T2) {
                if (T2 > I || T2 >= NT) continue;
                if (piv && T2 == t && I == t) continue;                 // done in C(k)
                if (b == 3 && T2 == t + 1 && I == t + 1) continue;      // done in the catch-up of region k + 1
                if (idx == want) return { 1, I, T2 };
                ++idx;
            }
        }
    }
    if (R >= 1 && R <= K) {
        const int k = R - 1, t = k / 4;
        const bool piv = (t & 3) == W;
        for (int c = 0; c < 2; ++c) {
            const int I = c ? I1 : I0;
            if (!(c ? act1 : act0) || I < t || (piv && I == t)) continue;
            if (idx == want) return { 2, I, 0 };
            ++idx;
        }
    }
    return { 0, 0, 0 };
}
template <int W, int NT, int R, int IDX, typename TS0, typename TS1>
__device__ __forceinline__ void pg_d4_lag_one(pg_d4_lds& sh, TS0& S0, TS1& S1, const double* aop, double B0, double B1, double lopL, int l)
{
    constexpr pg_lag_desc d = pg_lag_get(W, NT, R, IDX);
    constexpr int I0 = W;
    if constexpr (d.kind == 1) {
        if constexpr (d.I == I0) S0[d.T2] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[d.T2], B0, S0[d.T2], 0, 0, 0);
        else S1[d.T2] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[d.T2], B1, S1[d.T2], 0, 0, 0);
    } else if constexpr (d.kind == 2) {
        constexpr int k = R - 1, t = k / 4, b = k % 4, rb = k % 3;
        const pg_d4 zero4 = { 0.0, 0.0, 0.0, 0.0 };
        if constexpr (d.I == I0) {
            const pg_d4 r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(lopL, S0[t][b], zero4, 0, 0, 0);
            const double v = r4[0];
            S0[t][b] = v;
            sh.lp[rb][d.I][l] = v;
            if constexpr (b == 3 && d.I == t + 1 && d.I < NT) S0[d.I] = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, S0[d.I], 0, 0, 0);      // next pivot tile: its update of block k now
        } else {
            const pg_d4 r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(lopL, S1[t][b], zero4, 0, 0, 0);
            const double v = r4[0];
            S1[t][b] = v;
            sh.lp[rb][d.I][l] = v;
            if constexpr (b == 3 && d.I == t + 1 && d.I < NT) S1[d.I] = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, S1[d.I], 0, 0, 0);
        }
    }
}
template <int W, int NT, int R, int LO, int HI, typename TS0, typename TS1>
__device__ __forceinline__ void pg_d4_lag(pg_d4_lds& sh, TS0& S0, TS1& S1, const double* aop, double B0, double B1, double lopL, int l)
{
    if constexpr (LO < HI) {
        if constexpr (pg_lag_get(W, NT, R, LO).kind != 0) {
            pg_d4_lag_one<W, NT, R, LO>(sh, S0, S1, aop, B0, B1, lopL, l);
            pg_d4_lag<W, NT, R, LO + 1, HI>(sh, S0, S1, aop, B0, B1, lopL, l);
        }
    }
}
// one region (see above); `bad` accumulates the pivot failures
template <int W, int NT, int R, typename TS0, typename TS1>
__device__ __forceinline__ void pg_d4_region(pg_d4_lds& sh, TS0& S0, TS1& S1, double* __restrict__ tout, int l, int& bad)
{
    constexpr int I0 = W, I1 = W + 4, K = 4 * NT;
    constexpr bool act0 = I0 < NT, act1 = I1 < 7 && (I1 < NT || I1 == 6);
    const int j = l & 15, q = l >> 4;
    // operands of the trailing work, read from LDS at the top of the region
    double aop[6] = { 0, 0, 0, 0, 0, 0 }, lopL = 0.0, B0 = 0.0, B1 = 0.0;
    if constexpr (R >= 2) {
        constexpr int k = R - 2, t = k / 4, b = k % 4, rb = k % 3;
#pragma unroll
        for (int T2 = 0; T2 < 6; ++T2) aop[T2] = (T2 >= t && T2 < NT) ? -sh.lp[rb][T2][l] : 0.0;
        if (j < 4 * b + 4) aop[t] = 0.0;                                // pivot tile row: only the rows below the pivot block are updated
        if constexpr (act0 && I0 >= t) B0 = sh.lp[rb][I0][l];
        if constexpr (act1 && I1 >= t) B1 = sh.lp[rb][I1][l];
    }
    if constexpr (R >= 1 && R <= K) lopL = j < 4 ? sh.li[(R - 1) & 1][j * 4 + q] : 0.0;      // A operand of Linv x P^T: lane (i, m) = Linv[i][m]
    constexpr bool piv_now = R < K && ((R / 4) & 3) == W;
    constexpr bool steady = piv_now && (R % 4) != 0;                    // same pivot tile as in the region before: C(R) does not wait for the trailing work
    if constexpr (!steady) pg_d4_lag<W, NT, R, 0, 64>(sh, S0, S1, aop, B0, B1, lopL, l);
    // ---- C(R): the pivot chain, registers only.  In a steady region the trailing products sit BETWEEN its dependent vector
    // instructions (a wavefront issues in order: a product placed there costs an issue slot, its 64 cycles run beside the chain)
    if constexpr (piv_now) {
        constexpr int t = R / 4, b = R % 4, rb = R % 3;
        double m[10], li[10];
        double dv;
        if constexpr (t < 4) dv = S0[t][b]; else dv = S1[t][b];
        {
            int e = 0;
#pragma unroll
            for (int rr2 = 0; rr2 < 4; ++rr2)
#pragma unroll
                for (int c2 = 0; c2 <= rr2; ++c2) m[e++] = pg_readlane(dv, (4 * b + rr2) + 16 * c2);
        }
// slot i: the i-th trailing product, tied to the chain value `cv` just computed by an empty asm (the operands "pass through" it), so
// that neither the optimiser nor the scheduler can lift the product above this point of the chain
#define PG_SLOT(i, cv) do { if constexpr (steady) { asm volatile("" : "+v"(B0), "+v"(B1), "+v"(lopL) : "v"(cv)); pg_d4_lag<W, NT, R, (i), (i) + 1>(sh, S0, S1, aop, B0, B1, lopL, l); } } while (0)
#define PG_RSQ(x, rv, s0) do { rv = __builtin_amdgcn_rsq(x); PG_SLOT(s0, rv); rv = rv * (1.5 - 0.5 * x * rv * rv); PG_SLOT((s0) + 1, rv); rv = rv * (1.5 - 0.5 * x * rv * rv); } while (0)
        {   // pg_chol4_inv with the slots
            double d0 = m[0]; if (!(d0 > 0) || !isfinite(d0)) { bad = 1; d0 = 1.0; }
            double r0, r1, r2, r3;
            PG_RSQ(d0, r0, 0);
            const double l10 = m[1] * r0, l20 = m[3] * r0, l30 = m[6] * r0;
            PG_SLOT(2, l30);
            double d1 = m[2] - l10 * l10; if (!(d1 > 0) || !isfinite(d1)) { bad = 1; d1 = 1.0; }
            PG_RSQ(d1, r1, 3);
            const double l21 = (m[4] - l20 * l10) * r1, l31 = (m[7] - l30 * l10) * r1;
            PG_SLOT(5, l31);
            double d2 = m[5] - l20 * l20 - l21 * l21; if (!(d2 > 0) || !isfinite(d2)) { bad = 1; d2 = 1.0; }
            PG_RSQ(d2, r2, 6);
            const double l32 = (m[8] - l30 * l20 - l31 * l21) * r2;
            PG_SLOT(8, l32);
            double d3 = m[9] - l30 * l30 - l31 * l31 - l32 * l32; if (!(d3 > 0) || !isfinite(d3)) { bad = 1; d3 = 1.0; }
            PG_RSQ(d3, r3, 9);
            li[0] = r0; li[2] = r1; li[5] = r2; li[9] = r3;
            li[1] = -(l10 * r0) * r1;
            PG_SLOT(11, li[1]);
            li[3] = -(l20 * r0 + l21 * li[1]) * r2; li[4] = -(l21 * r1) * r2;
            li[6] = -(l30 * r0 + l31 * li[1] + l32 * li[3]) * r3; li[7] = -(l31 * r1 + l32 * li[4]) * r3; li[8] = -(l32 * r2) * r3;
        }
        if constexpr (steady) pg_d4_lag<W, NT, R, 12, 64>(sh, S0, S1, aop, B0, B1, lopL, l);
#undef PG_RSQ
#undef PG_SLOT
        if (l == 0) {                        // Linv row-major 4 x 4 into LDS (the zeros above its diagonal are there): the other wavefronts
            int e = 0;                       // read it in the next region, this one reads its own operand back right away
#pragma unroll
            for (int rr2 = 0; rr2 < 4; ++rr2)
#pragma unroll
                for (int c2 = 0; c2 <= rr2; ++c2) sh.li[R & 1][rr2 * 4 + c2] = li[e++];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // same wavefront: LDS operations complete in order
        const double lop = j < 4 ? sh.li[R & 1][j * 4 + q] : 0.0;      // A operand of Linv x P^T: lane (i, m) = Linv[i][m]
        const int rj = j - 4 * b;
        const pg_d4 zero4 = { 0.0, 0.0, 0.0, 0.0 };
        if constexpr (t < 4) {
            const pg_d4 r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(lop, S0[t][b], zero4, 0, 0, 0);
            double v = r4[0];
            if (rj < 0 || (rj < 4 && q > rj)) v = 0.0;               // rows above the block; zeros of L44
            S0[t][b] = v;
            sh.lp[rb][t][l] = v;
            const double a = j < 4 * b + 4 ? 0.0 : -v;               // only the rows below the pivot block are updated
            S0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, v, S0[t], 0, 0, 0);
        } else {
            const pg_d4 r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(lop, S1[t][b], zero4, 0, 0, 0);
            double v = r4[0];
            if (rj < 0 || (rj < 4 && q > rj)) v = 0.0;
            S1[t][b] = v;
            sh.lp[rb][t][l] = v;
            const double a = j < 4 * b + 4 ? 0.0 : -v;
            S1[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, v, S1[t], 0, 0, 0);
        }
    }
    if constexpr (piv_now) {                 // Linv for the kernels below the panel, off the chain
        if (l < 16) tout[(4 * (R / 4) + (R % 4)) * 16 + l] = sh.li[R & 1][l];
    }
    if constexpr (R < K + 1) PG_LDS_BARRIER();
}
template <int W, int NT, typename TS0, typename TS1, int... Rs>
__device__ __forceinline__ void pg_d4_regions(pg_d4_lds& sh, TS0& S0, TS1& S1, double* __restrict__ tout, int l, int& bad, std::integer_sequence<int, Rs...>)
{
    (pg_d4_region<W, NT, Rs>(sh, S0, S1, tout, l, bad), ...);
}
template <int W, int NT>
__device__ __forceinline__ void pg_diag4_body(pg_d4_lds& sh, double* __restrict__ A, double* __restrict__ rr, double* __restrict__ tout, int n, int ld, int l)
{
    constexpr int I0 = W, I1 = W + 4;
    constexpr bool has1 = I1 < 7;
    constexpr int K = 4 * NT;
    const int j = l & 15, q = l >> 4;
    pg_d4 S0[I0 + 1], S1[6];                       // tiles (T, I0), T <= I0 and (T, I1), T <= min(I1, 5)
#pragma unroll
    for (int T = 0; T <= I0; ++T)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = 16 * I0 + j, col = 16 * T + q + 4 * v;
            double val;
            if (row < n && col < n) val = col <= row ? A[(size_t)row * ld + col] : A[(size_t)col * ld + row];
            else val = row == col ? 1.0 : 0.0;
            S0[T][v] = val;
        }
    if (has1) {
#pragma unroll
        for (int T = 0; T <= (I1 < 6 ? I1 : 5); ++T)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = 16 * I1 + j, col = 16 * T + q + 4 * v;
                double val;
                if (I1 == 6) val = (j == 0 && col < n) ? rr[col] : 0.0;
                else if (row < n && col < n) val = col <= row ? A[(size_t)row * ld + col] : A[(size_t)col * ld + row];
                else val = row == col ? 1.0 : 0.0;
                S1[T][v] = val;
            }
    }
    int bad = 0;
    pg_d4_regions<W, NT>(sh, S0, S1, tout, l, bad, std::make_integer_sequence<int, K + 2>{});
    if (bad) sh.bad = 1;
#pragma unroll
    for (int T = 0; T <= I0; ++T)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = 16 * I0 + j, col = 16 * T + q + 4 * v;
            if (row < n && col <= row) A[(size_t)row * ld + col] = S0[T][v];
        }
    if (has1) {
#pragma unroll
        for (int T = 0; T <= (I1 < 6 ? I1 : 5); ++T)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = 16 * I1 + j, col = 16 * T + q + 4 * v;
                if (I1 == 6) { if (j == 0 && col < n) rr[col] = S1[T][v]; }
                else if (row < n && col <= row) A[(size_t)row * ld + col] = S1[T][v];
            }
    }
}
template <int NT>
__device__ __forceinline__ void pg_diag4_waves(pg_d4_lds& sh, double* __restrict__ A, double* __restrict__ rr, double* __restrict__ tout, int n, int ld, int w, int l)
{
    if (w == 0) pg_diag4_body<0, NT>(sh, A, rr, tout, n, ld, l);
    else if (w == 1) pg_diag4_body<1, NT>(sh, A, rr, tout, n, ld, l);
    else if (w == 2) pg_diag4_body<2, NT>(sh, A, rr, tout, n, ld, l);
    else pg_diag4_body<3, NT>(sh, A, rr, tout, n, ld, l);
}
__global__ __launch_bounds__(256) void pg_front_diag4_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD,
                                                             double* __restrict__ F, double* __restrict__ R, int* __restrict__ fail, double* __restrict__ Tinv)
{
    __shared__ pg_d4_lds sh;
    const pg_front fd = FD[it_front[blockIdx.x]];
    const int step = it_step[blockIdx.x], col0 = 96 * step;
    const int n = min(96, fd.s6 - col0), p = fd.pan0 + step, ld = fd.ld;
    double* __restrict__ A = F + fd.off + (size_t)col0 * ld + col0;
    double* __restrict__ rr = R + fd.roff + col0;
    double* __restrict__ tout = Tinv + (size_t)p * PG_NB4 * 16;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (threadIdx.x == 0) sh.bad = 0;
    if (threadIdx.x < 32) sh.li[threadIdx.x >> 4][threadIdx.x & 15] = 0.0;     // the zeros above the diagonal of Linv stay
    __syncthreads();
    switch ((n + 15) / 16) {                       // uniform over the workgroup
    case 1: pg_diag4_waves<1>(sh, A, rr, tout, n, ld, w, l); break;
    case 2: pg_diag4_waves<2>(sh, A, rr, tout, n, ld, w, l); break;
    case 3: pg_diag4_waves<3>(sh, A, rr, tout, n, ld, w, l); break;
    case 4: pg_diag4_waves<4>(sh, A, rr, tout, n, ld, w, l); break;
    case 5: pg_diag4_waves<5>(sh, A, rr, tout, n, ld, w, l); break;
    default: pg_diag4_waves<6>(sh, A, rr, tout, n, ld, w, l); break;
    }
    __syncthreads();
    if (sh.bad && threadIdx.x == 0) *fail = 1;
}

// L21 = A21 L11^-T for a slab of 16 rows per wavefront, by the same 4-column steps: LP = P Linv^T (three shuffles), then the
// later columns of the slab lose LP L11[later rows][pivot columns]^T (one MFMA per 16 columns, A operand straight from L11).
// Forward substitution rides along: b2 -= L21 y.
#define PG_T2_LD 97
__global__ __launch_bounds__(256) void pg_front_trsm2_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD,
                                                             double* __restrict__ F, double* __restrict__ R, const double* __restrict__ Tinv)
{
    __shared__ double sL[(PG_PW * 6) * PG_T2_LD];      // L11 of the panel (lower triangle), the A operands of every update
    __shared__ double sT[PG_NB4 * 16];                 // the 4 x 4 inverse blocks
    __shared__ double sY[PG_PW * 6];
    const pg_front fd = FD[it_front[blockIdx.x]];
    const int step = it_step[blockIdx.x], col0 = 96 * step;
    const int n = min(96, fd.s6 - col0), p = fd.pan0 + step, ld = fd.ld;
    const int row0 = col0 + n, nrows = fd.n6 - row0;
    if ((int)blockIdx.y * 64 >= nrows) return;     // workgroup-uniform
    const int l = threadIdx.x & 63, c = l & 15, q = l >> 4;
    const int rowbase = ((int)blockIdx.y * 4 + (int)(threadIdx.x >> 6)) * 16;
    const bool rok = rowbase + c < nrows;
    double* __restrict__ Arow = F + fd.off + (size_t)(row0 + min(max(rowbase + c, 0), nrows - 1)) * ld + col0;
    pg_d4 S[6];                                    // the slab's own rows are requested first: their latency hides behind the staging of L11
#pragma unroll
    for (int T = 0; T < 6; ++T)
#pragma unroll
        for (int v = 0; v < 4; ++v) { const int col = 16 * T + q + 4 * v; S[T][v] = (rok && col < n) ? Arow[col] : 0.0; }
    {
        const double* __restrict__ L11 = F + fd.off + (size_t)col0 * ld + col0;
        double v[36];
#pragma unroll
        for (int e = 0; e < 36; ++e) { const int id = e * 256 + threadIdx.x, r = id / 96, cc = id - 96 * r; v[e] = (r < n && cc <= r) ? L11[(size_t)r * ld + cc] : 0.0; }
#pragma unroll
        for (int e = 0; e < 36; ++e) { const int id = e * 256 + threadIdx.x, r = id / 96, cc = id - 96 * r; sL[r * PG_T2_LD + cc] = v[e]; }
        for (int e = threadIdx.x; e < PG_NB4 * 16; e += 256) sT[e] = Tinv[(size_t)p * PG_NB4 * 16 + e];
        if (threadIdx.x < 96) sY[threadIdx.x] = (int)threadIdx.x < n ? R[fd.roff + col0 + threadIdx.x] : 0.0;
    }
    __syncthreads();
    if (rowbase >= nrows) return;                  // wavefront-uniform
    // The operands of a 4-column step do not depend on the step before: they are read from LDS one step ahead, so that the matrix
    // core never waits for an LDS round trip between two dependent products (it did, 130 cycles per product).
    double lop_n, a_n[6];
    auto fetch = [&](int t, int b) {
        lop_n = c < 4 ? sT[(4 * t + b) * 16 + c * 4 + q] : 0.0;     // A operand of LP = Linv x P^T: lane (i, m) = Linv[i][m], i < 4
#pragma unroll
        for (int T2 = 0; T2 < 6; ++T2) {
            // A operand of the updates: -L11[16 T2 + i][16 t + 4 b + k] on lane (i = c, k = q), rows beyond the pivot block only
            const int ri = 16 * T2 + c, ck = 16 * t + 4 * b + q;
            a_n[T2] = (T2 > t || (T2 == t && c > 4 * b + 3)) ? -sL[ri * PG_T2_LD + ck] : 0.0;
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        if (16 * t >= n) break;                     // uniform: nothing beyond the panel's columns
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double lop = lop_n;
            double a[6];
#pragma unroll
            for (int T2 = 0; T2 < 6; ++T2) a[T2] = a_n[T2];
            if (b < 3) fetch(t, b + 1); else if (t < 5) fetch(t + 1, 0);
            // LP = P Linv^T through the matrix core (see pg_front_diag2_kernel)
            const pg_d4 zero4 = { 0.0, 0.0, 0.0, 0.0 };
            const pg_d4 r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(lop, S[t][b], zero4, 0, 0, 0);
            const double LP = r4[0];
            S[t][b] = LP;
#pragma unroll
            for (int T2 = t; T2 < 6; ++T2) S[T2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T2], LP, S[T2], 0, 0, 0);
        }
    }
    double dot = 0;
#pragma unroll
    for (int T = 0; T < 6; ++T)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int col = 16 * T + q + 4 * v;
            if (col < n) { if (rok) Arow[col] = S[T][v]; dot += S[T][v] * sY[col]; }
        }
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (l < 16 && rok) R[fd.roff + row0 + rowbase + c] -= dot;
}

// ---- Row solve and trailing update FUSED per 64 x 64 tile (levels with at most PG_RSU_MAX_TILES tiles: every level but the few at the
// bottom of the front tree).  pg_front_trsm2_kernel followed by pg_front_syrk_kernel costs two dependent launches per level, each with
// its ~10 us floor (arrival of the data the previous kernel wrote, prologue, strided stores), for a few microseconds of products.
// Here the workgroup of tile (ti, tj) solves BOTH row chunks it needs itself -- wavefronts 0..3 chunk ti (kept in registers: the tile
// registers ARE the A operands of the update), wavefronts 4..7 chunk tj (into LDS, the B operands) -- by the very steps of
// pg_front_trsm2_kernel, then updates the tile by the steps of pg_front_syrk_kernel: the same products in the same order, bit for bit.
// A chunk is solved once per tile that needs it (redundant flops on idle matrix cores); the diagonal tile (ti, ti) of a chunk stores
// its L21 rows and folds them into the right-hand side.  L21 goes to a SECOND front arena (FL): the tiles of a level run concurrently
// and read A21 in place, which an in-place store would pull from under them (the race that stopped round 2's version of this).
// TS = 64: the tile list of the level as it is (512 threads).  TS = 32 (round 4): every 64 x 64 tile of the list is cut into its (up to) four
// 32 x 32 quarters, one workgroup of 256 threads each (blockIdx & 3 = quarter; the quarters above the diagonal or beyond the front leave at
// once).  On the levels near the root a launch holds a handful of tiles on a 256-CU chip, and what a workgroup costs there is what ONE
// compute unit can pull and multiply: in-kernel variants with parts switched off put the fused kernel at 6.4 us (empty launch, with the
// event scope) + 9.2 (operands: L11, Tinv, two 64-row chunks, the tile -- 170 KB through one CU) + 5.7 (the two solves, two wavefronts per
// SIMD on the matrix core) + 3.6 (update).  A quarter moves 94 KB, solves two 32-row chunks on four SIMDs and updates a quarter of the
// tile.  The 16-row slabs and the 16 x 16 blocks see the same products in the same order: bit-identical to the 64 x 64 form.
template <int TS>
__global__ __launch_bounds__(TS * 8) void pg_front_rsu_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD,
                                                              const int* __restrict__ tile_item, const int* __restrict__ tile_ij,
                                                              double* __restrict__ F, double* __restrict__ FL, double* __restrict__ R, const double* __restrict__ Tinv)
{
    constexpr int NW = TS / 16, NTH = TS * 8;          // wavefronts per chunk, threads
    __shared__ double sL[(PG_PW * 6) * PG_T2_LD];      // L11 of the panel (lower triangle), the A operands of every solve step
    __shared__ double sT[PG_NB4 * 16];                 // the 4 x 4 inverse blocks
    __shared__ double sY[PG_PW * 6];
    __shared__ double sB[TS * PG_SYRK_LD];             // solved chunk tj: the B operands of the update
    const int tix = TS == 64 ? (int)blockIdx.x : (int)(blockIdx.x >> 2);
    const int item = tile_item[tix], ij = tile_ij[tix];
    int ti = ij >> 16, tj = ij & 0xffff;
    const pg_front fd = FD[it_front[item]];
    const int step = it_step[item], col0 = 96 * step;
    const int n = min(96, fd.s6 - col0), p = fd.pan0 + step, ld = fd.ld;
    const int row0 = col0 + n, nrows = fd.n6 - row0;
    if (TS == 32) {
        const int sub = blockIdx.x & 3;
        ti = 2 * ti + (sub >> 1); tj = 2 * tj + (sub & 1);
        if (tj > ti || 32 * ti >= nrows) return;       // workgroup-uniform: a quarter above the diagonal, or one that lies beyond the front
    }
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63, c = l & 15, q = l >> 4;
    const bool grp_i = wave < NW;                      // first NW wavefronts: chunk ti; the others: chunk tj
    const bool diag = ti == tj;
    const int w4 = wave & (NW - 1);
    const int rowbase = ((grp_i ? ti : tj) * NW + w4) * 16;
    const bool rok = rowbase + c < nrows;
    const bool solve = grp_i || !diag;                 // on a diagonal tile the second group has nothing to solve
    double* __restrict__ A = F + fd.off;
    const double* __restrict__ Arow = A + (size_t)(row0 + min(max(rowbase + c, 0), nrows - 1)) * ld + col0;
    pg_d4 S[6];                                        // the slab's own rows are requested first: their latency hides behind the staging of L11
#pragma unroll
    for (int T = 0; T < 6; ++T)
#pragma unroll
        for (int v = 0; v < 4; ++v) { const int col = 16 * T + q + 4 * v; S[T][v] = (solve && rok && col < n) ? Arow[col] : 0.0; }
    // the tile of C this wavefront updates comes in with the operands too (first group only)
    const int i0 = TS * ti + 16 * w4;
    pg_d4 acc[NW];
#pragma unroll
    for (int cb = 0; cb < NW; ++cb) {
        const int j0 = TS * tj + 16 * cb, jr = j0 + (l & 15);
        const double* __restrict__ Cp = A + (size_t)(row0 + i0 + (l >> 4)) * ld + row0 + j0 + (l & 15);
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[cb][v] = (grp_i && i0 < nrows && j0 <= i0 + 15 && i0 + (l >> 4) + 4 * v < nrows && jr < nrows) ? Cp[(size_t)(4 * v) * ld] : 0.0;
    }
    {
        const double* __restrict__ L11 = A + (size_t)col0 * ld + col0;
        constexpr int NE = 96 * 96 / NTH;
        double v[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) { const int id = e * NTH + threadIdx.x, r = id / 96, cc = id - 96 * r; v[e] = (r < n && cc <= r) ? L11[(size_t)r * ld + cc] : 0.0; }
#pragma unroll
        for (int e = 0; e < NE; ++e) { const int id = e * NTH + threadIdx.x, r = id / 96, cc = id - 96 * r; sL[r * PG_T2_LD + cc] = v[e]; }
        for (int e = threadIdx.x; e < PG_NB4 * 16; e += NTH) sT[e] = Tinv[(size_t)p * PG_NB4 * 16 + e];
        if (threadIdx.x < 96) sY[threadIdx.x] = (int)threadIdx.x < n ? R[fd.roff + col0 + threadIdx.x] : 0.0;
    }
    __syncthreads();
    if (solve && rowbase < nrows) {                    // wavefront-uniform: the 4-column steps of pg_front_trsm2_kernel
        double lop_n, a_n[6];
        auto fetch = [&](int t, int b) {
            lop_n = c < 4 ? sT[(4 * t + b) * 16 + c * 4 + q] : 0.0;
#pragma unroll
            for (int T2 = 0; T2 < 6; ++T2) {
                const int ri = 16 * T2 + c, ck = 16 * t + 4 * b + q;
                a_n[T2] = (T2 > t || (T2 == t && c > 4 * b + 3)) ? -sL[ri * PG_T2_LD + ck] : 0.0;
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (16 * t >= n) break;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double lop = lop_n;
                double a[6];
#pragma unroll
                for (int T2 = 0; T2 < 6; ++T2) a[T2] = a_n[T2];
                if (b < 3) fetch(t, b + 1); else if (t < 5) fetch(t + 1, 0);
                const pg_d4 zero4 = { 0.0, 0.0, 0.0, 0.0 };
                const pg_d4 r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(lop, S[t][b], zero4, 0, 0, 0);
                const double LP = r4[0];
                S[t][b] = LP;
#pragma unroll
                for (int T2 = t; T2 < 6; ++T2) S[T2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[T2], LP, S[T2], 0, 0, 0);
            }
        }
    }
    // chunk tj -> LDS in row-major order (the B operands); on a diagonal tile the first group's rows are that chunk
    if (diag ? grp_i : !grp_i) {
#pragma unroll
        for (int T = 0; T < 6; ++T)
#pragma unroll
            for (int v = 0; v < 4; ++v) sB[(16 * w4 + c) * PG_SYRK_LD + 16 * T + q + 4 * v] = S[T][v];
    }
    if (diag && grp_i && rowbase < nrows) {            // the chunk's L21 rows for the back-substitution, and their share of the forward solve
        double* __restrict__ Lrow = FL + fd.off + (size_t)(row0 + min(rowbase + c, nrows - 1)) * ld + col0;
        double dot = 0;
#pragma unroll
        for (int T = 0; T < 6; ++T)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int col = 16 * T + q + 4 * v;
                if (col < n) { if (rok) Lrow[col] = S[T][v]; dot += S[T][v] * sY[col]; }
            }
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        if (l < 16 && rok) R[fd.roff + row0 + rowbase + c] -= dot;
    }
    __syncthreads();
    if (!grp_i || i0 >= nrows) return;
    // A22 -= L21 L21^T on the tile: the steps of pg_front_syrk_kernel; A operand k = 4 ks + (l >> 4) of row (l & 15) is register (ks & 3) of
    // tile register ks >> 2 of this very lane
#pragma unroll
    for (int cb = 0; cb < NW; ++cb) {
        const int j0 = TS * tj + 16 * cb;
        if (j0 >= nrows || j0 > i0 + 15) break;
        const int jr = j0 + (l & 15);
        const double* __restrict__ sb = sB + (16 * cb + (l & 15)) * PG_SYRK_LD + (l >> 4);
        double* __restrict__ Cp = A + (size_t)(row0 + i0 + (l >> 4)) * ld + row0 + j0 + (l & 15);
        pg_d4 r = acc[cb];
#pragma unroll
        for (int ks = 0; ks < 24; ++ks) r = __builtin_amdgcn_mfma_f64_16x16x4f64(-S[ks >> 2][ks & 3], sb[4 * ks], r, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) if (i0 + (l >> 4) + 4 * v < nrows && jr < nrows) Cp[(size_t)(4 * v) * ld] = r[v];
    }
}

// x1 = L11^-T (y1 - L21^T x2) for one panel: one workgroup of 1024 threads.  x2 (the rows below the panel) is gathered into LDS,
// ten row slots accumulate the 96 column sums (folded in slot order), L11 goes global -> registers -> LDS behind them, and
// wavefront 0 runs the block back-substitution: x_blk = Linv^T z_blk, z[earlier columns] -= L11[blk rows][columns]^T x_blk.
// Tall fronts (the 4 M-pose graph of BASELINE config 5 has fronts of 7 000 rows): L21^T x2 of a panel is a 5 MB stream, and one
// workgroup pulls it through one compute unit at 30 - 50 GB/s -- 100 us and more per panel on the levels where the root front is alone.
// For panels with more than PG_BWD_SPLIT rows below them the product is split over workgroups of PG_BWD_RC rows each (this kernel:
// partial column sums, folded in slot order), and pg_front_bwd2_kernel adds the partial sums in chunk order instead of streaming L21.
__global__ __launch_bounds__(1024) void pg_front_bwd_part_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD,
                                                                 const int* __restrict__ f_rows, const double* __restrict__ FL, const double* __restrict__ x,
                                                                 double* __restrict__ part, int maxchunks)
{
    __shared__ double sx[PG_BWD_RC];
    __shared__ double s_acc[10 * (PG_PW * 6)];
    const pg_front fd = FD[it_front[blockIdx.y]];
    const int step = it_step[blockIdx.y], col0 = 96 * step;
    const int n = min(96, fd.s6 - col0), ld = fd.ld;
    const int row0 = col0 + n, nrows = fd.n6 - row0;
    if (nrows <= PG_BWD_SPLIT) return;
    const int r0 = blockIdx.x * PG_BWD_RC;
    if (r0 >= nrows) return;
    const int nr = min(PG_BWD_RC, nrows - r0);
    for (int i = threadIdx.x; i < nr; i += 1024) { const int g = row0 + r0 + i; sx[i] = x[(size_t)f_rows[fd.rowptr + g / 6] * 6 + g % 6]; }
    __syncthreads();
    const int slot = threadIdx.x / 96, cc = threadIdx.x - slot * 96;
    if (slot < 10) {
        double acc0 = 0, acc1 = 0;
        if (cc < n) {
            const double* __restrict__ Ab = FL + fd.off + (size_t)(row0 + r0) * ld + col0 + cc;
            int i = slot;
            for (; i + 150 < nr; i += 160) {                   // sixteen loads in flight
                double a16[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) a16[u] = Ab[(size_t)(i + 10 * u) * ld];
#pragma unroll
                for (int u = 0; u < 16; ++u) { if (u & 1) acc1 += a16[u] * sx[i + 10 * u]; else acc0 += a16[u] * sx[i + 10 * u]; }
            }
            for (; i < nr; i += 10) acc0 += Ab[(size_t)i * ld] * sx[i];
        }
        s_acc[slot * (PG_PW * 6) + cc] = acc0 + acc1;
    }
    __syncthreads();
    if (threadIdx.x < 96) {
        double v = 0;
        for (int g = 0; g < 10; ++g) v += s_acc[g * (PG_PW * 6) + threadIdx.x];
        part[((size_t)blockIdx.y * maxchunks + blockIdx.x) * 96 + threadIdx.x] = v;
    }
}
__global__ __launch_bounds__(1024) void pg_front_bwd2_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD,
                                                             const int* __restrict__ f_rows, const double* __restrict__ F, const double* __restrict__ FL, const double* __restrict__ R,
                                                             double* __restrict__ x, const double* __restrict__ Tinv, const double* __restrict__ part, int maxchunks)
{
    extern __shared__ double s_bw[];               // L11 [96 x 97] | Tinv [24][16] | slot sums [10][96] | x2 [nrows]
    double* sL = s_bw;
    double* sT = s_bw + (PG_PW * 6) * PG_BWD2_LD;  // the panel's 4 x 4 inverse blocks: a global load per block would sit on the serial chain of wave 0
    double* s_acc = sT + PG_NB4 * 16;
    double* sx = s_acc + 10 * (PG_PW * 6);
    const pg_front fd = FD[it_front[blockIdx.x]];
    const int step = it_step[blockIdx.x], col0 = 96 * step;
    const int n = min(96, fd.s6 - col0), p = fd.pan0 + step, ld = fd.ld;
    const int row0 = col0 + n, nrows = fd.n6 - row0;
    double lreg[9];
    { const double* __restrict__ L11 = F + fd.off + (size_t)col0 * ld + col0;
#pragma unroll
      for (int e = 0; e < 9; ++e) { const int id = e * 1024 + threadIdx.x, r = id / 96, cc = id - 96 * r; lreg[e] = (r < n && cc <= r) ? L11[(size_t)r * ld + cc] : 0.0; } }
    if (threadIdx.x < PG_NB4 * 16) sT[threadIdx.x] = Tinv[(size_t)p * PG_NB4 * 16 + threadIdx.x];
    const bool split = part != nullptr && nrows > PG_BWD_SPLIT;      // the product came in as partial sums (pg_front_bwd_part_kernel)
    const bool big = nrows > PG_BWD2_SX;           // only the largest interface fronts: x2 does not fit the LDS, read it through the row map
    if (!big && !split) for (int i = threadIdx.x; i < nrows; i += 1024) { const int g = row0 + i; sx[i] = x[(size_t)f_rows[fd.rowptr + g / 6] * 6 + g % 6]; }
    __syncthreads();
    {
        const int slot = threadIdx.x / 96, cc = threadIdx.x - slot * 96;
        if (slot < 10) {
            double acc0 = 0, acc1 = 0;
            if (split) {
                if (slot == 0 && cc < n) {
                    const int nch = (nrows + PG_BWD_RC - 1) / PG_BWD_RC;
                    const double* __restrict__ pp = part + (size_t)blockIdx.x * maxchunks * 96 + cc;
                    for (int ch = 0; ch < nch; ++ch) acc0 += pp[(size_t)ch * 96];
                }
            } else if (cc < n && !big) {
                const double* __restrict__ Ab = FL + fd.off + (size_t)row0 * ld + col0 + cc;      // L21: in place, or in the second arena where the level ran the fused row solve + update
                int i = slot;
                for (; i + 150 < nrows; i += 160) {                // sixteen loads in flight
                    double a16[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) a16[u] = Ab[(size_t)(i + 10 * u) * ld];
#pragma unroll
                    for (int u = 0; u < 16; ++u) { if (u & 1) acc1 += a16[u] * sx[i + 10 * u]; else acc0 += a16[u] * sx[i + 10 * u]; }
                }
                for (; i + 30 < nrows; i += 40) {
                    double a4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) a4[u] = Ab[(size_t)(i + 10 * u) * ld];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { if (u & 1) acc1 += a4[u] * sx[i + 10 * u]; else acc0 += a4[u] * sx[i + 10 * u]; }
                }
                for (; i < nrows; i += 10) acc0 += Ab[(size_t)i * ld] * sx[i];
            } else if (cc < n) {
                const double* __restrict__ Ab = FL + fd.off + (size_t)row0 * ld + col0 + cc;      // L21: in place, or in the second arena where the level ran the fused row solve + update
                for (int i = slot; i < nrows; i += 10) { const int g = row0 + i; acc0 += Ab[(size_t)i * ld] * x[(size_t)f_rows[fd.rowptr + g / 6] * 6 + g % 6]; }
            }
            s_acc[slot * (PG_PW * 6) + cc] = acc0 + acc1;
        }
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) { const int id = e * 1024 + threadIdx.x, r = id / 96, cc = id - 96 * r; sL[r * PG_BWD2_LD + cc] = lreg[e]; }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    // wavefront 0: lane c owns columns c and c + 64 of z
    const int lane = threadIdx.x;
    double z0, z1;
    {
        double v = lane < n ? R[fd.roff + col0 + lane] : 0.0;
        if (lane < n) for (int g = 0; g < 10; ++g) v -= s_acc[g * (PG_PW * 6) + lane];
        z0 = v;
        const int c1 = lane + 64;
        double w = c1 < n ? R[fd.roff + col0 + c1] : 0.0;
        if (c1 < n) for (int g = 0; g < 10; ++g) w -= s_acc[g * (PG_PW * 6) + c1];
        z1 = w;
    }
    const double* __restrict__ tin = sT;
#pragma unroll
    for (int blk = PG_NB4 - 1; blk >= 0; --blk) {
        if (4 * blk >= n) continue;                 // uniform (identity padding)
        // z of the four pivot columns -> every lane
        double zb[4], xb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int pc = 4 * blk + k; zb[k] = pc < 64 ? pg_readlane(z0, pc) : pg_readlane(z1, pc - 64); }
        const double* __restrict__ li = tin + blk * 16;        // Linv row-major; x = Linv^T z
#pragma unroll
        for (int k = 0; k < 4; ++k) { double s = 0; for (int m2 = k; m2 < 4; ++m2) s += li[m2 * 4 + k] * zb[m2]; xb[k] = s; }
        // earlier columns lose L11[pivot rows][column] x
        {
            double s0 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s0 += sL[(4 * blk + k) * PG_BWD2_LD + lane] * xb[k];
            if (lane < 4 * blk) z0 -= s0;
        }
        if (4 * blk > 64) {
            double s1 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s1 += sL[(4 * blk + k) * PG_BWD2_LD + min(lane + 64, 95)] * xb[k];
            if (lane + 64 < 4 * blk) z1 -= s1;
        }
        // the pivot columns take their solution
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int pc = 4 * blk + k; if (pc < 64) { if (lane == pc) z0 = xb[k]; } else if (lane == pc - 64) z1 = xb[k]; }
    }
    if (lane < n) x[(size_t)fd.c0 * 6 + col0 + lane] = z0;
    if (lane + 64 < n) x[(size_t)fd.c0 * 6 + col0 + lane + 64] = z1;
}

// the instances the driver launches (the template is declared in dsss_pg_kernels.h)
template __global__ void pg_front_rsu_kernel<32>(const int* __restrict__, const int* __restrict__, const pg_front* __restrict__, const int* __restrict__, const int* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, const double* __restrict__);
template __global__ void pg_front_rsu_kernel<64>(const int* __restrict__, const int* __restrict__, const pg_front* __restrict__, const int* __restrict__, const int* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, const double* __restrict__);
