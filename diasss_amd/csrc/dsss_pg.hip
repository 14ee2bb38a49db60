// diasss_amd/csrc/dsss_pg.hip -- pose-graph assembly and batch Levenberg-Marquardt solve on the device.
// Replaces the GTSAM NonlinearFactorGraph + iSAM2 of Optimizer::TrajOptimizationAll
// (/root/reference/src/core/optimizer.cpp:101-279): prior on X0 (:164-170), odometry BetweenFactor chain over every
// ping of every frame (:173-200, sigmas :24-28), at most one loop-closure BetweenFactor per target ping
// (:203-258, "last pair wins, first kp in it", score > 0, Diagonal::Variances), initial values DR o noise (:150-160).
// LM schedule = GTSAM LevenbergMarquardtParams() defaults, same loop as oracle/orc_posegraph.c.
//
// Linear algebra per LM trial (all f64 on the device, host only steers):
//   1. per-factor residuals + Jacobians, per-pose 6x6 Hessian blocks (block tridiagonal chain + LC blocks);
//   2. Schur complement of every chain segment between two LC-touched poses ("separators") onto its end points
//      -- segments are independent, one thread each, 6x6 block Thomas recursion;
//   3. the reduced system over the separators (chain couplings + LC blocks) is factorised by a sparse block
//      Cholesky: geometric nested-dissection ordering and symbolic analysis on the host (once per solve),
//      level-scheduled left-looking numeric factorisation and triangular solves in kernels;
//   4. back-substitution through the segments.
#include "dsss_internal.h"
#include "dsss_pose.h"
#include <algorithm>
#include <numeric>
#include <random>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <mutex>

// ------------------------------------------------------------------ small dense helpers (6x6 row-major)
__device__ inline int chol6(double* A)
{
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0) || !isfinite(d)) return -1;
        d = sqrt(d); A[j * 6 + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s / d;
        }
    }
    return 0;
}
// same factorisation with one reciprocal square root per pivot instead of a square root and five divisions (the
// dependent-latency chain of the panel kernels); ri[j] = 1 / L[j][j]
__device__ inline int chol6_fast(double* A, double* ri)
{
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0) || !isfinite(d)) { bad = 1; d = 1.0; }
        const double r = rsqrt(d);
        A[j * 6 + j] = d * r; ri[j] = r;
#pragma unroll
        for (int i = 0; i < 6; ++i) if (i > j) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k < j) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s * r;
        }
    }
    return bad;
}
// b (6 x nrhs, row-major) <- (L L^T)^-1 b
__device__ inline void chol6_solve(const double* L, double* b, int nrhs)
{
    for (int c = 0; c < nrhs; ++c) {
        for (int i = 0; i < 6; ++i) { double s = b[i * nrhs + c]; for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * b[k * nrhs + c]; b[i * nrhs + c] = s / L[i * 6 + i]; }
        for (int i = 5; i >= 0; --i) { double s = b[i * nrhs + c]; for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * b[k * nrhs + c]; b[i * nrhs + c] = s / L[i * 6 + i]; }
    }
}

struct pg_weights { double prior[6], odo[6]; };

// ------------------------------------------------------------------ factors
// factor k < n: k == 0 prior on X0 (measurement DR0), else Between(X_{k-1}, X_k); factor n + e: LC edge e.
// r = whitened residual, Ji = whitened Jacobian wrt the first pose (-W Ad(h^-1)); the Jacobian wrt the second
// pose is W itself (BetweenFactor with GTSAM_SLOW_BUT_CORRECT_BETWEENFACTOR off, PriorFactor H = I).
__device__ inline void factor_eval(int k, int n, const pose_t* X, const pose_t* meas, const pg_weights& W,
                                   const int* ea, const int* eb, const pose_t* emeas, const double* ew,
                                   double* r, double* Ji)
{
    double xi[6];
    if (k == 0) {
        pose_t d;
        pose_between(&meas[0], &X[0], &d);
        pose_log(&d, xi);
        for (int a = 0; a < 6; ++a) r[a] = xi[a] * W.prior[a];
        if (Ji) for (int a = 0; a < 36; ++a) Ji[a] = 0.0;
        return;
    }
    int i, j; const pose_t* m; const double* w;
    if (k < n) { i = k - 1; j = k; m = &meas[k]; w = W.odo; }
    else { const int e = k - n; i = ea[e]; j = eb[e]; m = &emeas[e]; w = ew + (size_t)e * 6; }
    pose_t h, er;
    pose_between(&X[i], &X[j], &h);
    pose_between(m, &h, &er);
    pose_log(&er, xi);
    for (int a = 0; a < 6; ++a) r[a] = xi[a] * w[a];
    if (Ji) {
        pose_t hi; double Ad[36];
        pose_inverse(&h, &hi);
        pose_adjoint(&hi, Ad);
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) Ji[a * 6 + b] = -Ad[a * 6 + b] * w[a];
    }
}

// deterministic block sum: wave shuffle tree then the 4 wave sums in order
__device__ inline double block_sum256(double v, double* s_w)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}

__global__ __launch_bounds__(256) void pg_linearize_kernel(int n, int ne, const pose_t* __restrict__ X, const pose_t* __restrict__ meas,
                                                           pg_weights W, const int* __restrict__ ea, const int* __restrict__ eb,
                                                           const pose_t* __restrict__ emeas, const double* __restrict__ ew,
                                                           double* __restrict__ r, double* __restrict__ Ji, double* __restrict__ partial)
{
    __shared__ double s_w[4];
    const int k = blockIdx.x * 256 + threadIdx.x;
    double e2 = 0;
    if (k < n + ne) {
        double rr[6], J[36];
        factor_eval(k, n, X, meas, W, ea, eb, emeas, ew, rr, Ji ? J : nullptr);
        for (int a = 0; a < 6; ++a) { e2 += rr[a] * rr[a]; if (r) r[(size_t)k * 6 + a] = rr[a]; }
        if (Ji) for (int a = 0; a < 36; ++a) Ji[(size_t)k * 36 + a] = J[a];
    }
    const double s = block_sum256(e2, s_w);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void pg_final_sum_kernel(const double* __restrict__ partial, int n, double scale, double* __restrict__ out)
{
    __shared__ double s_w[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const double s = block_sum256(acc, s_w);
    if (threadIdx.x == 0) *out = s * scale;
}

// per-pose Hessian blocks: D (diagonal), C = H(i, i+1), g = J^T r.  LC contributions are summed over the
// pose's incidence list in a fixed order (no atomics).
__global__ __launch_bounds__(256) void pg_assemble_kernel(int n, pg_weights W, const double* __restrict__ r, const double* __restrict__ Ji,
                                                          const int* __restrict__ adj_ptr, const int* __restrict__ adj_edge,
                                                          const double* __restrict__ ew, const double* __restrict__ lambda_ptr,
                                                          double* __restrict__ D, double* __restrict__ C, double* __restrict__ g)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double Dd[36], Cc[36], gg[6];
    for (int a = 0; a < 36; ++a) { Dd[a] = 0; Cc[a] = 0; }
    for (int a = 0; a < 6; ++a) gg[a] = 0;
    // factor i with this pose as the second variable (Jacobian W)
    const double* w2 = i == 0 ? W.prior : W.odo;
    for (int a = 0; a < 6; ++a) { Dd[a * 6 + a] += w2[a] * w2[a]; gg[a] += w2[a] * r[(size_t)i * 6 + a]; }
    if (i + 1 < n) {   // factor i+1 with this pose as the first variable
        const double* J = Ji + (size_t)(i + 1) * 36; const double* rr = r + (size_t)(i + 1) * 6;
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) {
                double s = 0;
                for (int q = 0; q < 6; ++q) s += J[q * 6 + a] * J[q * 6 + b];
                Dd[a * 6 + b] += s;
                Cc[a * 6 + b] = J[b * 6 + a] * W.odo[b];            // Ji^T W
            }
            double s = 0;
            for (int q = 0; q < 6; ++q) s += J[q * 6 + a] * rr[q];
            gg[a] += s;
        }
    }
    for (int p = adj_ptr[i]; p < adj_ptr[i + 1]; ++p) {
        const int code = adj_edge[p], e = code >> 1, second = code & 1;
        const double* rr = r + (size_t)(n + e) * 6;
        if (second) {
            const double* w = ew + (size_t)e * 6;
            for (int a = 0; a < 6; ++a) { Dd[a * 6 + a] += w[a] * w[a]; gg[a] += w[a] * rr[a]; }
        } else {
            const double* J = Ji + (size_t)(n + e) * 36;
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) { double s = 0; for (int q = 0; q < 6; ++q) s += J[q * 6 + a] * J[q * 6 + b]; Dd[a * 6 + b] += s; }
                double s = 0;
                for (int q = 0; q < 6; ++q) s += J[q * 6 + a] * rr[q];
                gg[a] += s;
            }
        }
    }
    const double lambda = *lambda_ptr;
    for (int a = 0; a < 6; ++a) Dd[a * 6 + a] += lambda;
    for (int a = 0; a < 36; ++a) { D[(size_t)i * 36 + a] = Dd[a]; C[(size_t)i * 36 + a] = Cc[a]; }
    for (int a = 0; a < 6; ++a) g[(size_t)i * 6 + a] = gg[a];
}

// Schur complement of the interior of segment s (poses L+1 .. R-1) onto its end points L, R.
// Stores the Cholesky factor of every eliminated pivot (Dl), the fill block E_i = H(L, i) and the updated
// gradient for the back-substitution; outputs the end-point corrections.
__global__ __launch_bounds__(64) void pg_segment_kernel(int nseg, const int* __restrict__ sep_pose, const double* __restrict__ D,
                                                        const double* __restrict__ C, const double* __restrict__ g,
                                                        double* __restrict__ E, double* __restrict__ Dl, double* __restrict__ gi,
                                                        double* __restrict__ segDL, double* __restrict__ segDR, double* __restrict__ segGL,
                                                        double* __restrict__ segGR, double* __restrict__ segS, int* __restrict__ fail)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= nseg) return;
    const int L = sep_pose[s], R = sep_pose[s + 1];
    double DL[36], GL[6], Dn[36], Gn[6], Ei[36];
    for (int a = 0; a < 36; ++a) DL[a] = 0;
    for (int a = 0; a < 6; ++a) GL[a] = 0;
    if (R == L + 1) {
        for (int a = 0; a < 36; ++a) { segDL[(size_t)s * 36 + a] = 0; segDR[(size_t)s * 36 + a] = 0; segS[(size_t)s * 36 + a] = C[(size_t)L * 36 + a]; }
        for (int a = 0; a < 6; ++a) { segGL[(size_t)s * 6 + a] = 0; segGR[(size_t)s * 6 + a] = 0; }
        return;
    }
    for (int a = 0; a < 36; ++a) { Ei[a] = C[(size_t)L * 36 + a]; Dn[a] = D[(size_t)(L + 1) * 36 + a]; }
    for (int a = 0; a < 6; ++a) Gn[a] = g[(size_t)(L + 1) * 6 + a];
    for (int i = L + 1; i < R; ++i) {
        double Li[36], XE[36], XC[36], Xg[6], Ci[36];
        for (int a = 0; a < 36; ++a) { Li[a] = Dn[a]; Ci[a] = C[(size_t)i * 36 + a]; E[(size_t)i * 36 + a] = Ei[a]; }
        for (int a = 0; a < 6; ++a) { Xg[a] = Gn[a]; gi[(size_t)i * 6 + a] = Gn[a]; }
        if (chol6(Li)) { *fail = 1; return; }
        for (int a = 0; a < 36; ++a) Dl[(size_t)i * 36 + a] = Li[a];
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) XE[a * 6 + b] = Ei[b * 6 + a];      // E_i^T
        for (int a = 0; a < 36; ++a) XC[a] = Ci[a];
        chol6_solve(Li, XE, 6); chol6_solve(Li, XC, 6); chol6_solve(Li, Xg, 1);
        // next pivot block and its gradient (the right separator's share when i + 1 == R)
        double Dnext[36], Gnext[6], En[36];
        const bool last = (i + 1 == R);
        for (int a = 0; a < 36; ++a) Dnext[a] = last ? 0.0 : D[(size_t)(i + 1) * 36 + a];
        for (int a = 0; a < 6; ++a) Gnext[a] = last ? 0.0 : g[(size_t)(i + 1) * 6 + a];
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) {
                double sLL = 0, sLn = 0, snn = 0;
                for (int q = 0; q < 6; ++q) {
                    sLL += Ei[a * 6 + q] * XE[q * 6 + b];
                    sLn += Ei[a * 6 + q] * XC[q * 6 + b];
                    snn += Ci[q * 6 + a] * XC[q * 6 + b];
                }
                DL[a * 6 + b] -= sLL;
                En[a * 6 + b] = -sLn;
                Dnext[a * 6 + b] -= snn;
            }
            double tL = 0, tn = 0;
            for (int q = 0; q < 6; ++q) { tL += Ei[a * 6 + q] * Xg[q]; tn += Ci[q * 6 + a] * Xg[q]; }
            GL[a] -= tL;
            Gnext[a] -= tn;
        }
        for (int a = 0; a < 36; ++a) { Ei[a] = En[a]; Dn[a] = Dnext[a]; }
        for (int a = 0; a < 6; ++a) Gn[a] = Gnext[a];
    }
    for (int a = 0; a < 36; ++a) { segDL[(size_t)s * 36 + a] = DL[a]; segDR[(size_t)s * 36 + a] = Dn[a]; segS[(size_t)s * 36 + a] = Ei[a]; }
    for (int a = 0; a < 6; ++a) { segGL[(size_t)s * 6 + a] = GL[a]; segGR[(size_t)s * 6 + a] = Gn[a]; }
}

// reduced system: diagonal blocks, chain couplings and right-hand side (one thread per separator, chain order)
__global__ __launch_bounds__(256) void pg_scatter_base_kernel(int ns, const int* __restrict__ sep_pose, const int* __restrict__ perm,
                                                              const double* __restrict__ D, const double* __restrict__ g,
                                                              const double* __restrict__ segDL, const double* __restrict__ segDR,
                                                              const double* __restrict__ segGL, const double* __restrict__ segGR,
                                                              const double* __restrict__ segS, const int* __restrict__ diag_pos,
                                                              const int* __restrict__ ch_pos, double* __restrict__ Lvals, double* __restrict__ rhs)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns) return;
    const int p = sep_pose[k];
    double* dst = Lvals + (size_t)diag_pos[k] * 36;
    for (int a = 0; a < 36; ++a) {
        double v = D[(size_t)p * 36 + a];
        if (k > 0) v += segDR[(size_t)(k - 1) * 36 + a];
        if (k + 1 < ns) v += segDL[(size_t)k * 36 + a];
        dst[a] = v;
    }
    double* rr = rhs + (size_t)perm[k] * 6;
    for (int a = 0; a < 6; ++a) {
        double v = g[(size_t)p * 6 + a];
        if (k > 0) v += segGR[(size_t)(k - 1) * 6 + a];
        if (k + 1 < ns) v += segGL[(size_t)k * 6 + a];
        rr[a] = -v;
    }
    if (k + 1 < ns) {      // S(k, k+1): stored as the (larger index, smaller index) block
        const int code = ch_pos[k], pos = code >> 1, tr = code & 1;
        double* c = Lvals + (size_t)pos * 36;
        const double* S = segS + (size_t)k * 36;
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) c[a * 6 + b] = tr ? S[b * 6 + a] : S[a * 6 + b];
    }
}
// LC off-diagonal blocks H(a, b) = Ji^T W (added after the chain couplings; (a,b) is unique per edge)
__global__ __launch_bounds__(256) void pg_scatter_lc_kernel(int n, int ne, const double* __restrict__ Ji, const double* __restrict__ ew,
                                                            const int* __restrict__ lc_pos, double* __restrict__ Lvals)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= ne) return;
    const int code = lc_pos[e], pos = code >> 1, tr = code & 1;
    const double* J = Ji + (size_t)(n + e) * 36; const double* w = ew + (size_t)e * 6;
    double* c = Lvals + (size_t)pos * 36;
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) {
        const double h_ab = J[b * 6 + a] * w[b];               // (Ji^T W)(a, b)
        if (tr) c[b * 6 + a] += h_ab; else c[a * 6 + b] += h_ab;
    }
}

// The factorisation kernels below (down to pg_sep_delta_kernel) are compared with the oracle at 1e-6 on the poses, not
// bit for bit, so they may fuse multiply-adds; everything else in the library stays at -ffp-contract=off.
#pragma clang fp contract(fast)
// ---- sparse block Cholesky, left-looking, one workgroup per column of the current elimination-tree level.
// Column j holds blocks L(i, j), i in rowidx[colptr[j] .. colptr[j+1]) ascending, first the diagonal.
// rowlist(j) = columns k < j with L(j, k) != 0 and the position of that block.
// upd_map (built once per solve by pg_build_map_kernel): for update t of column j and target block q the position of
// L(i_q, k_t) or -1; layout [mapptr[j] + t * m_j + q], so the factor kernel has no dependent index search.
__global__ __launch_bounds__(256) void pg_build_map_kernel(int nupd, const int* __restrict__ rlrow, const int* __restrict__ rlptr,
                                                           const int* __restrict__ rlcol, const int* __restrict__ rlpos,
                                                           const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                           const long long* __restrict__ mapptr, int* __restrict__ upd_map)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nupd) return;
    const int j = rlrow[e], k = rlcol[e];
    const int c0 = colptr[j], m = colptr[j + 1] - c0;
    int* mp = upd_map + mapptr[j] + (long long)(e - rlptr[j]) * m;
    int q = 0;
    for (int p = rlpos[e]; p < colptr[k + 1]; ++p) {           // rows of column k from row j downwards: a subset of column j's rows
        const int i = rowidx[p];
        while (q < m && rowidx[c0 + q] < i) ++q;
        if (q < m && rowidx[c0 + q] == i) mp[q] = p;
    }
}

#define PG_TCH 128
// acc[s] += sum_t L(i,k_t)[r][:] . L(j,k_t)[s][:] for the tn updates staged in LDS.  Updates that do not touch this
// row (map entry -1) are masked instead of skipped, and four updates are in flight at once, so the index load and the
// six operand loads of different updates overlap instead of forming one dependent chain per update.
__device__ inline void pg_acc_rows(const int* __restrict__ mp, int m, int tn, const double* __restrict__ Lvals, int r,
                                   const double* __restrict__ s_Ljk, double* acc)
{
    int t = 0;
    for (; t + 4 <= tn; t += 4) {
        int pos[4]; double a[4][6];
#pragma unroll
        for (int u = 0; u < 4; ++u) pos[u] = mp[(size_t)(t + u) * m];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double* Lik = Lvals + (size_t)(pos[u] < 0 ? 0 : pos[u]) * 36 + r * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) a[u][c] = Lik[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (pos[u] < 0) continue;
            const double* B = s_Ljk + (t + u) * 36;
#pragma unroll
            for (int s = 0; s < 6; ++s)
                acc[s] += a[u][0] * B[s * 6] + a[u][1] * B[s * 6 + 1] + a[u][2] * B[s * 6 + 2] + a[u][3] * B[s * 6 + 3] + a[u][4] * B[s * 6 + 4] + a[u][5] * B[s * 6 + 5];
        }
    }
    for (; t < tn; ++t) {
        const int pos = mp[(size_t)t * m];
        if (pos < 0) continue;
        const double* Lik = Lvals + (size_t)pos * 36 + r * 6;
        const double* B = s_Ljk + t * 36;
        const double a0 = Lik[0], a1 = Lik[1], a2 = Lik[2], a3 = Lik[3], a4 = Lik[4], a5 = Lik[5];
#pragma unroll
        for (int s = 0; s < 6; ++s)
            acc[s] += a0 * B[s * 6] + a1 * B[s * 6 + 1] + a2 * B[s * 6 + 2] + a3 * B[s * 6 + 3] + a4 * B[s * 6 + 4] + a5 * B[s * 6 + 5];
    }
}
// accumulate sum_k L(i,k) L(j,k)^T for a slice of the update list: grid (columns of the level, 256-row chunks of the
// column, slices of the update list).  With one slice the result is subtracted from A(i,j) in place; with several
// slices each writes its partial sum to `part` and the panel kernels subtract them in slice order (deterministic).
__global__ __launch_bounds__(256) void pg_factor_acc_kernel(const int* __restrict__ lvcols, const int* __restrict__ colptr,
                                                            const int* __restrict__ rlptr, const int* __restrict__ rlpos,
                                                            const long long* __restrict__ mapptr, const int* __restrict__ upd_map,
                                                            double* __restrict__ Lvals, double* __restrict__ part, int col_stride,
                                                            const int* __restrict__ rlcol, double* __restrict__ x, const int* __restrict__ tbeg, const int* __restrict__ tend,
                                                            const double* __restrict__ fold_part, int fold_nsl)
{
    __shared__ double s_Ljk[PG_TCH * 36];
    __shared__ double s_yk[PG_TCH * 6];
    const int j = lvcols[blockIdx.x];
    const int c0 = colptr[j], m = colptr[j + 1] - c0;
    if ((int)blockIdx.y * 256 >= 6 * m) return;
    const int t0 = rlptr[j], Tb = tbeg ? tbeg[j] : 0, T = tend[j] - Tb;   // this launch's share of the updates from outside the column's own panel
    const int nsl = gridDim.z, sl = blockIdx.z;
    const int per = (T + nsl - 1) / nsl;
    const int ta = Tb + sl * per, tb = min(Tb + T, ta + per);
    const int* mp = upd_map + mapptr[j];
    const int idx = blockIdx.y * 256 + threadIdx.x;
    const bool act = idx < 6 * m;
    const int q = act ? idx / 6 : 0, r = idx - q * 6;
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    // forward substitution fused in: the right-hand side is one more block row of the column, y_j -= sum_k L(j,k) y_k
    const bool rhs = blockIdx.y == 0 && threadIdx.x >= 250;
    const int rs_ = threadIdx.x - 250;
    double accy = 0;
    for (int tc = ta; tc < tb; tc += PG_TCH) {
        const int tn = min(PG_TCH, tb - tc);
        __syncthreads();
        for (int x0 = threadIdx.x; x0 < tn * 36; x0 += 4 * 256) {      // four dependent (position -> block) loads in flight per thread
            int pos[4]; double val[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int xx = x0 + u * 256; pos[u] = xx < tn * 36 ? rlpos[t0 + tc + xx / 36] : 0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int xx = x0 + u * 256; val[u] = xx < tn * 36 ? Lvals[(size_t)pos[u] * 36 + (xx % 36)] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int xx = x0 + u * 256; if (xx < tn * 36) s_Ljk[xx] = val[u]; }
        }
        if (blockIdx.y == 0) for (int xx = threadIdx.x; xx < tn * 6; xx += 256) s_yk[xx] = x[(size_t)rlcol[t0 + tc + xx / 6] * 6 + (xx % 6)];
        __syncthreads();
        if (rhs)
            for (int t = 0; t < tn; ++t) {
                const double* yk = s_yk + t * 6;
                const double* B = s_Ljk + t * 36 + rs_ * 6;
                accy += B[0] * yk[0] + B[1] * yk[1] + B[2] * yk[2] + B[3] * yk[3] + B[4] * yk[4] + B[5] * yk[5];
            }
        if (act)
            pg_acc_rows(mp + (size_t)tc * m + q, m, tn, Lvals, r, s_Ljk, acc);
    }
    // an in-place launch (one slice) can also fold the partial sums another, sliced launch left for this level: slice
    // order fixed, after this launch's own sum (same arithmetic as pg_fold_kernel running behind it)
    if (rhs) {
        if (nsl == 1) {
            double v = x[(size_t)j * 6 + rs_] - accy;
            for (int s2 = 0; s2 < fold_nsl; ++s2) v -= fold_part[((size_t)blockIdx.x * fold_nsl + s2) * col_stride + (size_t)col_stride - 8 + rs_];
            x[(size_t)j * 6 + rs_] = v;
        } else part[((size_t)blockIdx.x * nsl + sl) * col_stride + (size_t)col_stride - 8 + rs_] = accy;
    }
    if (!act) return;
    if (nsl == 1) {
        double* row = Lvals + (size_t)(c0 + q) * 36 + r * 6;
        double v[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) v[s] = row[s] - acc[s];
        int s2 = 0;
        for (; s2 + 4 <= fold_nsl; s2 += 4) {                   // four slices in flight, subtracted in slice order
            double o[4][6];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double* op = fold_part + ((size_t)blockIdx.x * fold_nsl + s2 + u) * col_stride + (size_t)idx * 6;
#pragma unroll
                for (int s = 0; s < 6; ++s) o[u][s] = op[s];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int s = 0; s < 6; ++s) v[s] -= o[u][s];
        }
        for (; s2 < fold_nsl; ++s2) {
            const double* o = fold_part + ((size_t)blockIdx.x * fold_nsl + s2) * col_stride + (size_t)idx * 6;
#pragma unroll
            for (int s = 0; s < 6; ++s) v[s] -= o[s];
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) row[s] = v[s];
    }
    else { double* o = part + ((size_t)blockIdx.x * nsl + sl) * col_stride + (size_t)idx * 6; for (int s = 0; s < 6; ++s) o[s] = acc[s]; }
}
// fold the per-slice partial sums of pg_factor_acc_kernel into the column blocks and the right-hand side, slice order
// fixed (deterministic); keeps the serial critical paths of the panel kernels free of the nsl-fold reads
__global__ __launch_bounds__(256) void pg_fold_kernel(const int* __restrict__ lvcols, const int* __restrict__ colptr, double* __restrict__ Lvals,
                                                      const double* __restrict__ part, int nsl, int col_stride, double* __restrict__ x)
{
    const int j = lvcols[blockIdx.x];
    const int c0 = colptr[j], m = colptr[j + 1] - c0;
    const int idx = blockIdx.y * 256 + threadIdx.x;
    if (idx < 6 * m) {
        double* row = Lvals + (size_t)c0 * 36 + (size_t)idx * 6;
        double v[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) v[s] = row[s];
        for (int sl = 0; sl < nsl; ++sl) {
            const double* o = part + ((size_t)blockIdx.x * nsl + sl) * col_stride + (size_t)idx * 6;
#pragma unroll
            for (int s = 0; s < 6; ++s) v[s] -= o[s];
        }
#pragma unroll
        for (int s = 0; s < 6; ++s) row[s] = v[s];
    }
    if (blockIdx.y == 0 && threadIdx.x < 6) {
        double v = x[(size_t)j * 6 + threadIdx.x];
        for (int sl = 0; sl < nsl; ++sl) v -= part[((size_t)blockIdx.x * nsl + sl) * col_stride + (size_t)col_stride - 8 + threadIdx.x];
        x[(size_t)j * 6 + threadIdx.x] = v;
    }
}

// ---- supernodal panels for the top of the tree.  A panel is up to PG_PW consecutive columns j0 .. j0+w-1 with nested
// structure (struct(j+1) = struct(j) minus j), i.e. a dense trapezoid: a 6w x 6w diagonal block A11 on top of the rows
// A21 shared by all its columns.  After the external updates (pg_factor_acc_kernel over all columns of the level at once)
//     L11 = chol(A11), y = L11^-1 b, W = L11^-1     pg_panel_diag_kernel   one workgroup per panel, A11 dense in LDS
//     L21 = A21 W^T                                 pg_panel_trsm_kernel   one wavefront per 16 scalar rows of A21
// which replaces w column-levels by one panel-level; both run on the f64 matrix cores.
#define PG_PW 16
#define PG_BWD_LDS (((PG_PW * 6) * (PG_PW * 6) + 64 * (PG_PW * 6)) * (int)sizeof(double))
__device__ inline size_t pan_pos(const int* __restrict__ colptr, int j0, int qi, int c) { return (size_t)colptr[j0 + c] + (size_t)(qi - c); }

// ---- pg_panel_diag_kernel: the 6w x 6w diagonal block of a panel as a dense matrix in LDS, worked in 16 x 16 tiles.
//   per tile step t:  A  (16 lanes) L_tt = chol(A_tt) and V_t = L_tt^-1, rows/columns held across lanes, pivots and
//                        multipliers broadcast with v_readlane
//                     B  L_it = A_it V_t^T for the tiles below            (v_mfma_f64_16x16x4_f64, 4 per tile)
//                     C  A_ij -= L_it L_jt^T for the trailing tiles       (same); wavefront 0 takes tile (t+1, t+1) first
//                        and goes straight on to step A of t+1 while wavefronts 1-3 finish the rest (look-ahead)
//   then W = L11^-1 by recursive doubling over tiles, [A 0; B C]^-1 = [A^-1 0; -C^-1 B A^-1  C^-1], again on the matrix
//   cores, so that the row solve below the panel and the back-substitution are plain products with W.
// The right-hand side rides along: y_t = V_t b_t, b_i -= L_it y_t.
#define PG_LD 98                                   // LDS row stride of the 96 x 96 images (doubles)
#define PG_DIAG_LDS (2 * (PG_PW * 6) * PG_LD * (int)sizeof(double))
typedef double pg_d4 __attribute__((ext_vector_type(4)));
__device__ inline double pg_readlane(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// MFMA operand images of a 16 x 16 tile at (R0, C0) of a row-major LDS matrix.  "a": lane l holds [l & 15][4 ks + (l >> 4)],
// i.e. the tile as the A operand, or its transpose as the B operand; "b": lane l holds [4 ks + (l >> 4)][l & 15].
__device__ inline void pg_ld_a(const double* s, int R0, int C0, int l, double sign, double* f)
{
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = sign * s[(R0 + (l & 15)) * PG_LD + C0 + 4 * ks + (l >> 4)];
}
__device__ inline void pg_ld_b(const double* s, int R0, int C0, int l, double* f)
{
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = s[(R0 + 4 * ks + (l >> 4)) * PG_LD + C0 + (l & 15)];
}
__device__ inline pg_d4 pg_ld_c(const double* s, int R0, int C0, int l)
{
    pg_d4 c;
#pragma unroll
    for (int v = 0; v < 4; ++v) c[v] = s[(R0 + (l >> 4) + 4 * v) * PG_LD + C0 + (l & 15)];
    return c;
}
__device__ inline void pg_st_c(double* s, int R0, int C0, int l, pg_d4 c)
{
#pragma unroll
    for (int v = 0; v < 4; ++v) s[(R0 + (l >> 4) + 4 * v) * PG_LD + C0 + (l & 15)] = c[v];
}
__device__ inline pg_d4 pg_mma4(const double* a, const double* b, pg_d4 c)
{
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], c, 0, 0, 0);
    return c;
}
// step A for tile t, executed by lanes 0..15 of one wavefront: lane i owns row i of A_tt, then column i of V_t
__device__ inline int pg_tile_factor(double* sA, double* sW, int t, int lane)
{
    double d[16], v[16];
    double* row = sA + (16 * t + lane) * PG_LD + 16 * t;
#pragma unroll
    for (int j = 0; j < 16; ++j) d[j] = row[j];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double pj = pg_readlane(d[j], j);
        if (!(pj > 0) || !isfinite(pj)) { bad = 1; pj = 1.0; }
        double r = __builtin_amdgcn_rsq(pj);                 // v_rsq_f64 seed, two Newton steps
        r = r * (1.5 - 0.5 * pj * r * r);
        r = r * (1.5 - 0.5 * pj * r * r);
        d[j] = (lane == j ? pj : d[j]) * r;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k > j) d[k] -= d[j] * pg_readlane(d[j], k);      // A_ik -= L_ij L_kj (used for i >= k)
        // row j of V = L^-1 is complete data-wise now (it needs rows <= j of L): lane c holds V[j][c], zero above the
        // diagonal; two interleaved partial sums, and the work overlaps with the next pivot's reciprocal square root
        {
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) if (k < j) { const double pr = pg_readlane(d[k], j) * v[k]; if (k & 1) s1 += pr; else s0 += pr; }
            v[j] = lane == j ? r : (lane < j ? -(s0 + s1) * r : 0.0);
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) row[j] = j <= lane ? d[j] : 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) sW[(16 * t + i) * PG_LD + 16 * t + lane] = v[i];
    return bad;
}

__global__ __launch_bounds__(256) void pg_panel_diag_kernel(const int* __restrict__ plvpan, const int* __restrict__ pan_first, const int* __restrict__ pan_w,
                                                            const int* __restrict__ colptr, double* __restrict__ Lvals,
                                                            double* __restrict__ x, int* __restrict__ fail, double* __restrict__ Wsw, double* __restrict__ Wrow)
{
    extern __shared__ double s_dyn[];
    __shared__ double sy[PG_PW * 6];
    __shared__ int s_bad;
    double* sA = s_dyn;
    double* sW = s_dyn + (PG_PW * 6) * PG_LD;
    const int p = plvpan[blockIdx.x];
    const int j0 = pan_first[p], w = pan_w[p], n = 6 * w;
    const int nt = (n + 15) >> 4, np = 16 * nt;
    const int bi = threadIdx.x >> 4, bj = threadIdx.x & 15;
    const bool act = bi < w && bj <= bi;
    const int wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (threadIdx.x == 0) s_bad = 0;
    if (act) {
        const double* src = Lvals + pan_pos(colptr, j0, bi, bj) * 36;
#pragma unroll
        for (int e = 0; e < 36; ++e) sA[(6 * bi + e / 6) * PG_LD + 6 * bj + e % 6] = src[e];
    }
    for (int e = threadIdx.x; e < (np - n) * np; e += 256) {        // identity padding up to the tile boundary
        const int i = n + e / np, c2 = e % np;
        sA[i * PG_LD + c2] = i == c2 ? 1.0 : 0.0;
    }
    if ((int)threadIdx.x < np) sy[threadIdx.x] = (int)threadIdx.x < n ? x[(size_t)(j0 + threadIdx.x / 6) * 6 + threadIdx.x % 6] : 0.0;
    __syncthreads();
    if (threadIdx.x < 16) { if (pg_tile_factor(sA, sW, 0, threadIdx.x)) s_bad = 1; }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        // ---- B: tiles below the diagonal tile, one per wavefront in turn; lanes 0..15 of wavefront 0 also finish y_t
        for (int i = t + 1 + wave; i < nt; i += 4) {
            double fa[4], fb[4];
            pg_ld_a(sA, 16 * i, 16 * t, l, 1.0, fa);
            pg_ld_a(sW, 16 * t, 16 * t, l, 1.0, fb);          // B[k][c] = V_t[c][k]
            pg_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
            acc = pg_mma4(fa, fb, acc);
            pg_st_c(sA, 16 * i, 16 * t, l, acc);
        }
        if (threadIdx.x < 16) {
            double bb[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) bb[k] = sy[16 * t + k];
            double yv = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) yv += sW[(16 * t + threadIdx.x) * PG_LD + 16 * t + k] * bb[k];
            sy[16 * t + threadIdx.x] = yv;
        }
        __syncthreads();
        // ---- C: trailing update
        if (t + 1 < nt) {
            if (wave == 0) {
                double fa[4], fb[4];
                pg_ld_a(sA, 16 * (t + 1), 16 * t, l, -1.0, fa);
                pg_ld_a(sA, 16 * (t + 1), 16 * t, l, 1.0, fb);
                pg_d4 acc = pg_ld_c(sA, 16 * (t + 1), 16 * (t + 1), l);
                acc = pg_mma4(fa, fb, acc);
                pg_st_c(sA, 16 * (t + 1), 16 * (t + 1), l, acc);
                if (l < 16) { if (pg_tile_factor(sA, sW, t + 1, l)) s_bad = 1; }
            } else {
                int cnt = 0;
                for (int i = t + 1; i < nt; ++i)
                    for (int j = t + 1; j <= i; ++j) {
                        if (i == t + 1 && j == t + 1) continue;
                        if (cnt++ % 3 != wave - 1) continue;
                        double fa[4], fb[4];
                        pg_ld_a(sA, 16 * i, 16 * t, l, -1.0, fa);
                        pg_ld_a(sA, 16 * j, 16 * t, l, 1.0, fb);
                        pg_d4 acc = pg_ld_c(sA, 16 * i, 16 * j, l);
                        acc = pg_mma4(fa, fb, acc);
                        pg_st_c(sA, 16 * i, 16 * j, l, acc);
                    }
                const int i = 16 * (t + 1) + (int)threadIdx.x - 64;      // b_i -= L_it y_t
                if (i < np) {
                    double v = sy[i];
#pragma unroll
                    for (int k = 0; k < 16; ++k) v -= sA[i * PG_LD + 16 * t + k] * sy[16 * t + k];
                    sy[i] = v;
                }
            }
        }
        __syncthreads();
    }
    if (s_bad && threadIdx.x == 0) *fail = 1;
    if (act) {
        double* dst = Lvals + pan_pos(colptr, j0, bi, bj) * 36;
#pragma unroll
        for (int e = 0; e < 36; ++e) dst[e] = (bi == bj && e % 6 > e / 6) ? 0.0 : sA[(6 * bi + e / 6) * PG_LD + 6 * bj + e % 6];
    }
    if ((int)threadIdx.x < n) x[(size_t)(j0 + threadIdx.x / 6) * 6 + threadIdx.x % 6] = sy[threadIdx.x];
    // ---- W = L11^-1: off-diagonal tiles by recursive doubling (the diagonal tiles V_t are in place)
    for (int h = 1; h < nt; h <<= 1) {
        {   int cnt = 0;                                        // T = B A^-1 into the target tiles
            for (int g = 0; (2 * g + 1) * h < nt; ++g) {
                const int gmid = (2 * g + 1) * h, iend = min(gmid + h, nt);
                for (int i = gmid; i < iend; ++i)
                    for (int j = 2 * g * h; j < gmid; ++j) {
                        if ((cnt++ & 3) != wave) continue;
                        pg_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
                        for (int k = j; k < gmid; ++k) {
                            double fa[4], fb[4];
                            pg_ld_a(sA, 16 * i, 16 * k, l, 1.0, fa);
                            pg_ld_b(sW, 16 * k, 16 * j, l, fb);
                            acc = pg_mma4(fa, fb, acc);
                        }
                        pg_st_c(sW, 16 * i, 16 * j, l, acc);
                    }
            }
        }
        __syncthreads();
        pg_d4 r0 = { 0.0, 0.0, 0.0, 0.0 }, r1 = { 0.0, 0.0, 0.0, 0.0 };
        int ti0 = -1, tj0 = 0, ti1 = -1, tj1 = 0;
        {   int cnt = 0;                                        // W_B = -C^-1 T, kept in registers until every T is consumed
            for (int g = 0; (2 * g + 1) * h < nt; ++g) {
                const int gmid = (2 * g + 1) * h, iend = min(gmid + h, nt);
                for (int i = gmid; i < iend; ++i)
                    for (int j = 2 * g * h; j < gmid; ++j) {
                        if ((cnt++ & 3) != wave) continue;
                        pg_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
                        for (int k = gmid; k <= i; ++k) {
                            double fa[4], fb[4];
                            pg_ld_a(sW, 16 * i, 16 * k, l, -1.0, fa);
                            pg_ld_b(sW, 16 * k, 16 * j, l, fb);
                            acc = pg_mma4(fa, fb, acc);
                        }
                        if (ti0 < 0) { r0 = acc; ti0 = i; tj0 = j; } else { r1 = acc; ti1 = i; tj1 = j; }
                    }
            }
        }
        __syncthreads();
        if (ti0 >= 0) pg_st_c(sW, 16 * ti0, 16 * tj0, l, r0);
        if (ti1 >= 0) pg_st_c(sW, 16 * ti1, 16 * tj1, l, r1);
        __syncthreads();
    }
    // W out: row-major for the back-substitution, and in MFMA B-operand order for the row solve (tile nt = 16 output
    // columns, k-step ks = 4 k's: lane l holds W[16 nt + (l & 15)][4 ks + (l >> 4)]).  Thread (bi, bj) writes its block;
    // everything outside the lower block triangle of the first w block rows was zeroed once by the host and stays zero.
    if (act) {
        double* wr = Wrow + (size_t)p * (PG_PW * 6) * (PG_PW * 6);
        double* ws = Wsw + (size_t)p * (PG_PW * 6) * (PG_PW * 6);
        int offc[6];
#pragma unroll
        for (int c2 = 0; c2 < 6; ++c2) { const int kc = 6 * bj + c2; offc[c2] = (kc >> 2) * 64 + (kc & 3) * 16; }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int jr = 6 * bi + r;
            const int offr = (jr >> 4) * (24 * 64) + (jr & 15);
#pragma unroll
            for (int c2 = 0; c2 < 6; ++c2) {
                const double v = (bi == bj && c2 > r) ? 0.0 : sW[jr * PG_LD + 6 * bj + c2];
                wr[(size_t)jr * (PG_PW * 6) + 6 * bj + c2] = v;
                ws[offr + offc[c2]] = v;
            }
        }
    }
}

// L21 = A21 W^T with W = L11^-1 from pg_panel_diag_kernel: a plain f64 GEMM on the matrix cores.  One wavefront per
// 16 scalar rows: the 16 x 96 slab of A21 sits in 24 A-operand registers per lane (lane l: row l & 15, k = 4 ks + (l >> 4)),
// W^T streams in as pre-swizzled B operands (one coalesced 512-byte load per v_mfma_f64_16x16x4_f64), and output tile nt
// only runs the k-steps its triangular W reaches (4 nt + 4 of 24).
__global__ __launch_bounds__(256) void pg_panel_trsm_kernel(const int* __restrict__ plvpan, const int* __restrict__ pan_first, const int* __restrict__ pan_w,
                                                            const int* __restrict__ colptr, double* __restrict__ Lvals, const double* __restrict__ Wsw)
{
    const int p = plvpan[blockIdx.x];
    const int j0 = pan_first[p], w = pan_w[p], n = 6 * w;
    const int m = colptr[j0 + 1] - colptr[j0];     // block rows of the first column
    const int nrows = 6 * (m - w);
    const int l = threadIdx.x & 63;
    const int rowbase = ((int)blockIdx.y * 4 + (int)(threadIdx.x >> 6)) * 16;
    if (rowbase >= nrows) return;                  // wavefront-uniform
    const double* Wp = Wsw + (size_t)p * (PG_PW * 6) * (PG_PW * 6);
    const int arow = rowbase + (l & 15);
    const bool rok = arow < nrows;
    const int qi = w + arow / 6, r = arow % 6;
    double a[24];
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) {
        const int k = 4 * ks + (l >> 4), cb = k / 6, s = k - 6 * cb;
        a[ks] = (rok && k < n) ? Lvals[((size_t)colptr[j0 + min(cb, w - 1)] + (size_t)(qi - cb)) * 36 + r * 6 + s] : 0.0;
    }
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        if (16 * nt < n) {                         // uniform
            pg_d4 acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int ks = 0; ks < 4 * nt + 4; ++ks)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], Wp[(nt * 24 + ks) * 64 + l], acc, 0, 0, 0);
            const int col = 16 * nt + (l & 15);
            if (col < n) {
                const int cb = col / 6, s = col - 6 * cb;
                const size_t base = (size_t)colptr[j0 + cb];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = rowbase + (l >> 4) + 4 * v;
                    if (row < nrows) Lvals[(base + (size_t)(w + row / 6 - cb)) * 36 + (row % 6) * 6 + s] = acc[v];
                }
            }
        }
    }
}

// x_panel = W^T (y_panel - L21^T x_below), W = L11^-1: one workgroup of 1024 threads per panel
#define PG_BWD_SLOTS 64
__global__ __launch_bounds__(1024) void pg_panel_bwd_kernel(const int* __restrict__ plvpan, const int* __restrict__ pan_first, const int* __restrict__ pan_w,
                                                            const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                            const double* __restrict__ Lvals, double* __restrict__ x, const double* __restrict__ Wrow)
{
    extern __shared__ double s_bw[];               // W [96 x 96] then the slot sums [PG_BWD_SLOTS][96]
    __shared__ double sz[PG_PW * 6];
    double* sW = s_bw;
    double* s_acc = s_bw + (PG_PW * 6) * (PG_PW * 6);
    const int p = plvpan[blockIdx.x];
    const int j0 = pan_first[p], w = pan_w[p], n = 6 * w;
    const int c0 = colptr[j0], m = colptr[j0 + 1] - c0;
    // W goes global -> registers now, -> LDS after the row sums, so its latency hides behind them
    double wreg[9];
    { const double* wr = Wrow + (size_t)p * (PG_PW * 6) * (PG_PW * 6);
#pragma unroll
      for (int e = 0; e < 9; ++e) wreg[e] = wr[e * 1024 + threadIdx.x]; }
    // z[gj] = y[gj] - sum over rows below of L(row, gj) x_row : thread (slot, c) walks block rows slot, slot+64, ... of block
    // column c; the slot sums of every scalar column are then folded in slot order (deterministic)
    {
        const int slot = threadIdx.x >> 4, cc = threadIdx.x & 15;
        double acc[6] = { 0, 0, 0, 0, 0, 0 };
        if (cc < w) {
            const size_t cbase = (size_t)colptr[j0 + cc] - (size_t)cc;
            for (int qi = w + slot; qi < m; qi += PG_BWD_SLOTS) {
                const double* B = Lvals + (cbase + qi) * 36; const double* xi = x + (size_t)rowidx[c0 + qi] * 6;
                const double x0 = xi[0], x1 = xi[1], x2 = xi[2], x3 = xi[3], x4 = xi[4], x5 = xi[5];
#pragma unroll
                for (int s6 = 0; s6 < 6; ++s6) acc[s6] += B[s6] * x0 + B[6 + s6] * x1 + B[12 + s6] * x2 + B[18 + s6] * x3 + B[24 + s6] * x4 + B[30 + s6] * x5;
            }
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) s_acc[slot * (PG_PW * 6) + cc * 6 + s6] = acc[s6];
        }
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) sW[e * 1024 + threadIdx.x] = wreg[e];
    __syncthreads();
    if ((int)threadIdx.x < n) {
        const int gj = threadIdx.x;
        double v = x[(size_t)(j0 + gj / 6) * 6 + gj % 6];
        for (int g = 0; g < PG_BWD_SLOTS; ++g) v -= s_acc[g * (PG_PW * 6) + gj];
        sz[gj] = v;
    }
    __syncthreads();
    // x[i] = sum_{k >= i} W[k][i] z[k], ascending k, two interleaved partial sums
    if ((int)threadIdx.x < n) {
        const int i = threadIdx.x;
        double v0 = 0, v1 = 0;
        int k = i;
        for (; k + 1 < n; k += 2) { v0 += sW[k * (PG_PW * 6) + i] * sz[k]; v1 += sW[(k + 1) * (PG_PW * 6) + i] * sz[k + 1]; }
        if (k < n) v0 += sW[k * (PG_PW * 6) + i] * sz[k];
        x[(size_t)(j0 + i / 6) * 6 + i % 6] = v0 + v1;
    }
}

// ---- bottom of the elimination tree: whole subtrees per workgroup (no grid-wide level barriers).
// A bin is a list of columns in ascending order whose ancestors outside the bin are never its descendants, so the
// workgroup can run them back to back with workgroup barriers only.  Same arithmetic and summation order as the
// level-scheduled kernels.  Only columns with at most 42 blocks (6m <= 256 rows: one pass) are binned.
__global__ __launch_bounds__(256) void pg_factor_subtree_kernel(const int* __restrict__ binptr, const int* __restrict__ bincols,
                                                                const int* __restrict__ colptr, const int* __restrict__ rlptr,
                                                                const int* __restrict__ rlcol, const int* __restrict__ rlpos,
                                                                const long long* __restrict__ mapptr, const int* __restrict__ upd_map,
                                                                double* __restrict__ Lvals, double* __restrict__ x, int* __restrict__ fail)
{
    __shared__ double s_Ljk[PG_TCH * 36];
    __shared__ double s_yk[PG_TCH * 6];
    __shared__ double s_diag[36];
    __shared__ int s_ok;
    for (int ci = binptr[blockIdx.x]; ci < binptr[blockIdx.x + 1]; ++ci) {
        const int j = bincols[ci];
        const int c0 = colptr[j], m = colptr[j + 1] - c0;
        const int t0 = rlptr[j], T = rlptr[j + 1] - t0;
        const int* mp = upd_map + mapptr[j];
        const int idx = threadIdx.x;
        const bool act = idx < 6 * m;
        const int q = act ? idx / 6 : 0, r = idx - q * 6;
        const bool rhs = threadIdx.x >= 250;
        const int rs_ = threadIdx.x - 250;
        double acc[6] = { 0, 0, 0, 0, 0, 0 }, accy = 0;
        for (int tc = 0; tc < T; tc += PG_TCH) {
            const int tn = min(PG_TCH, T - tc);
            __syncthreads();
            for (int x0 = threadIdx.x; x0 < tn * 36; x0 += 4 * 256) {      // four dependent (position -> block) loads in flight per thread
            int pos[4]; double val[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int xx = x0 + u * 256; pos[u] = xx < tn * 36 ? rlpos[t0 + tc + xx / 36] : 0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int xx = x0 + u * 256; val[u] = xx < tn * 36 ? Lvals[(size_t)pos[u] * 36 + (xx % 36)] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int xx = x0 + u * 256; if (xx < tn * 36) s_Ljk[xx] = val[u]; }
        }
            for (int xx = threadIdx.x; xx < tn * 6; xx += 256) s_yk[xx] = x[(size_t)rlcol[t0 + tc + xx / 6] * 6 + (xx % 6)];
            __syncthreads();
            if (rhs)
                for (int t = 0; t < tn; ++t) {
                    const double* yk = s_yk + t * 6; const double* B = s_Ljk + t * 36 + rs_ * 6;
                    accy += B[0] * yk[0] + B[1] * yk[1] + B[2] * yk[2] + B[3] * yk[3] + B[4] * yk[4] + B[5] * yk[5];
                }
            if (act)
                pg_acc_rows(mp + (size_t)tc * m + q, m, tn, Lvals, r, s_Ljk, acc);
        }
        if (rhs && T > 0) x[(size_t)j * 6 + rs_] -= accy;
        if (act && T > 0) for (int s = 0; s < 6; ++s) Lvals[(size_t)(c0 + q) * 36 + r * 6 + s] -= acc[s];
        __syncthreads();
        if (threadIdx.x == 0) {
            double A[36];
            for (int a = 0; a < 36; ++a) A[a] = Lvals[(size_t)c0 * 36 + a];
            const int bad = chol6(A);
            if (bad) *fail = 1;
            s_ok = !bad;
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) { const double v = b <= a ? A[a * 6 + b] : 0.0; s_diag[a * 6 + b] = v; Lvals[(size_t)c0 * 36 + a * 6 + b] = v; }
            if (!bad) {
                double v[6];
                for (int a = 0; a < 6; ++a) { double t = x[(size_t)j * 6 + a]; for (int b = 0; b < a; ++b) t -= A[a * 6 + b] * v[b]; v[a] = t / A[a * 6 + a]; }
                for (int a = 0; a < 6; ++a) x[(size_t)j * 6 + a] = v[a];
            }
        }
        __syncthreads();
        if (!s_ok) return;
        if (act && idx >= 6) {
            double* row = Lvals + (size_t)(c0 + q) * 36 + r * 6;
            double xr[6];
            for (int s = 0; s < 6; ++s) { double v = row[s]; for (int c = 0; c < s; ++c) v -= xr[c] * s_diag[s * 6 + c]; xr[s] = v / s_diag[s * 6 + s]; }
            for (int s = 0; s < 6; ++s) row[s] = xr[s];
        }
        __syncthreads();
        __threadfence_block();
    }
}
// backward substitution through a bin, columns in descending order, one wave per bin
__global__ __launch_bounds__(64) void pg_bwd_subtree_kernel(const int* __restrict__ binptr, const int* __restrict__ bincols,
                                                            const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                            const double* __restrict__ Lvals, double* __restrict__ x)
{
    const int lane = threadIdx.x;
    for (int ci = binptr[blockIdx.x + 1] - 1; ci >= binptr[blockIdx.x]; --ci) {
        const int j = bincols[ci];
        double acc[6] = { 0, 0, 0, 0, 0, 0 };
        for (int p = colptr[j] + 1 + lane; p < colptr[j + 1]; p += 64) {
            const double* B = Lvals + (size_t)p * 36; const double* xi = x + (size_t)rowidx[p] * 6;
            for (int a = 0; a < 6; ++a) { double s = 0; for (int b = 0; b < 6; ++b) s += B[b * 6 + a] * xi[b]; acc[a] += s; }
        }
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) acc[a] += __shfl_xor(acc[a], o, 64);
        if (lane == 0) {
            const double* Ld = Lvals + (size_t)colptr[j] * 36;
            double v[6];
            for (int a = 5; a >= 0; --a) { double s = x[(size_t)j * 6 + a] - acc[a]; for (int b = a + 1; b < 6; ++b) s -= Ld[b * 6 + a] * v[b]; v[a] = s / Ld[a * 6 + a]; }
            for (int a = 0; a < 6; ++a) x[(size_t)j * 6 + a] = v[a];
        }
        __threadfence_block();
        __builtin_amdgcn_s_barrier();
    }
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void pg_sep_delta_kernel(int ns, const int* __restrict__ sep_pose, const int* __restrict__ perm,
                                                           const double* __restrict__ x, double* __restrict__ delta)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns) return;
    for (int a = 0; a < 6; ++a) delta[(size_t)sep_pose[k] * 6 + a] = x[(size_t)perm[k] * 6 + a];
}

// interiors, right to left: delta_i = D_i^-1 (-g_i - E_i^T delta_L - C_i delta_{i+1})
__global__ __launch_bounds__(64) void pg_backsub_kernel(int nseg, const int* __restrict__ sep_pose, const double* __restrict__ C,
                                                        const double* __restrict__ E, const double* __restrict__ Dl, const double* __restrict__ gi,
                                                        double* __restrict__ delta)
{
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= nseg) return;
    const int L = sep_pose[s], R = sep_pose[s + 1];
    double dL[6], dn[6];
    for (int a = 0; a < 6; ++a) { dL[a] = delta[(size_t)L * 6 + a]; dn[a] = delta[(size_t)R * 6 + a]; }
    for (int i = R - 1; i > L; --i) {
        double b[6];
        for (int a = 0; a < 6; ++a) {
            double t = -gi[(size_t)i * 6 + a];
            for (int q = 0; q < 6; ++q) { t -= E[(size_t)i * 36 + q * 6 + a] * dL[q]; t -= C[(size_t)i * 36 + a * 6 + q] * dn[q]; }
            b[a] = t;
        }
        double Li[36];
        for (int a = 0; a < 36; ++a) Li[a] = Dl[(size_t)i * 36 + a];
        chol6_solve(Li, b, 1);
        for (int a = 0; a < 6; ++a) { delta[(size_t)i * 6 + a] = b[a]; dn[a] = b[a]; }
    }
}

// 0.5 * || J delta + r ||^2 over all factors (linear.error(delta))
__global__ __launch_bounds__(256) void pg_linerr_kernel(int n, int ne, pg_weights W, const int* __restrict__ ea, const int* __restrict__ eb,
                                                        const double* __restrict__ ew, const double* __restrict__ r, const double* __restrict__ Ji,
                                                        const double* __restrict__ delta, double* __restrict__ partial)
{
    __shared__ double s_w[4];
    const int k = blockIdx.x * 256 + threadIdx.x;
    double e2 = 0;
    if (k < n + ne) {
        int i = -1, j; const double* w;
        if (k == 0) { j = 0; w = W.prior; }
        else if (k < n) { i = k - 1; j = k; w = W.odo; }
        else { i = ea[k - n]; j = eb[k - n]; w = ew + (size_t)(k - n) * 6; }
        for (int a = 0; a < 6; ++a) {
            double s = r[(size_t)k * 6 + a] + w[a] * delta[(size_t)j * 6 + a];
            if (i >= 0) for (int q = 0; q < 6; ++q) s += Ji[(size_t)k * 36 + a * 6 + q] * delta[(size_t)i * 6 + q];
            e2 += s * s;
        }
    }
    const double s = block_sum256(e2, s_w);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void pg_retract_kernel(int n, const pose_t* __restrict__ X, const double* __restrict__ delta, pose_t* __restrict__ Xn)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    pose_t o;
    pose_retract(&X[i], delta + (size_t)i * 6, &o);
    Xn[i] = o;
}

// ------------------------------------------------------------------ initial values on the device
// std::default_random_engine (minstd_rand0, seed 1) + std::normal_distribution<double> (libstdc++ Marsaglia polar,
// optimizer.cpp:30-31,154-158) without the sequential dependency: polar attempt a always consumes engine outputs
// 4a+1 .. 4a+4 (two generate_canonical calls of two engine calls each), so every attempt is evaluated independently
// after an O(log a) jump-ahead of the LCG; accepted attempts are compacted in order and each yields (y*mult, x*mult).
__device__ inline unsigned long long minstd_pow(unsigned long long e)
{
    unsigned long long r = 1, b = 16807ULL;
    const unsigned long long m = 2147483647ULL;
    while (e) { if (e & 1) r = (r * b) % m; b = (b * b) % m; e >>= 1; }
    return r;
}
#define RNG_PER_THREAD 16
__global__ __launch_bounds__(256) void pg_rng_attempts_kernel(long long nattempts, double* __restrict__ pairs, int* __restrict__ flags)
{
    const long long a0 = ((long long)blockIdx.x * 256 + threadIdx.x) * RNG_PER_THREAD;
    if (a0 >= nattempts) return;
    const unsigned long long m = 2147483647ULL;
    unsigned long long x = minstd_pow((unsigned long long)(4 * a0));        // state after 4*a0 engine calls (seed 1)
    const double R = 2147483646.0;
    for (int k = 0; k < RNG_PER_THREAD && a0 + k < nattempts; ++k) {
        double cn[2];
        for (int q = 0; q < 2; ++q) {
            x = (x * 16807ULL) % m; const double e1 = (double)(x - 1);
            x = (x * 16807ULL) % m; const double e2 = (double)(x - 1);
            double can = (e1 + e2 * R) / (R * R);
            if (can >= 1.0) can = 0.99999999999999988897769753748;   // nextafter(1, 0)
            cn[q] = can;
        }
        const double u = 2.0 * cn[0] - 1.0, v = 2.0 * cn[1] - 1.0, r2 = u * u + v * v;
        const bool ok = !(r2 > 1.0 || r2 == 0.0);
        double mult = 0;
        if (ok) mult = sqrt(-2 * log(r2) / r2);
        pairs[2 * (a0 + k)] = v * mult; pairs[2 * (a0 + k) + 1] = u * mult;
        flags[a0 + k] = ok ? 1 : 0;
    }
}
// exclusive scan of flags in three steps (block sums, scan of block sums by one block, compaction)
__global__ __launch_bounds__(256) void pg_flag_blocksum_kernel(const int* __restrict__ flags, long long n, int* __restrict__ bsum)
{
    __shared__ int s_w[4];
    const long long i0 = (long long)blockIdx.x * 4096;
    int acc = 0;
    for (int k = threadIdx.x; k < 4096; k += 256) if (i0 + k < n) acc += flags[i0 + k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void pg_flag_scan_kernel(int* __restrict__ bsum, int nb, int* __restrict__ total)
{
    if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < nb; ++i) { const int v = bsum[i]; bsum[i] = run; run += v; } *total = run; }
}
__global__ __launch_bounds__(256) void pg_flag_compact_kernel(const int* __restrict__ flags, const double* __restrict__ pairs, long long n,
                                                              const int* __restrict__ bsum, long long need_pairs, double* __restrict__ normals)
{
    __shared__ int s_w[4];
    __shared__ int s_run;
    const long long i0 = (long long)blockIdx.x * 4096;
    if (threadIdx.x == 0) s_run = bsum[blockIdx.x];
    __syncthreads();
    for (int c = 0; c < 4096; c += 256) {
        const long long i = i0 + c + threadIdx.x;
        const int f = (i < n) ? flags[i] : 0;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int inc = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        __syncthreads();
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int base = s_run;
        for (int k = 0; k < w; ++k) base += s_w[k];
        const long long pos = (long long)base + inc - f;
        if (f && pos < need_pairs) { normals[2 * pos] = pairs[2 * i]; normals[2 * pos + 1] = pairs[2 * i + 1]; }
        __syncthreads();
        if (threadIdx.x == 0) s_run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}
// x, y of the separator poses (the coordinates the nested dissection bisects)
__global__ __launch_bounds__(256) void pg_sep_xy_kernel(int ns, const int* __restrict__ sep_pose, const double* __restrict__ dr6, double* __restrict__ xy)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns) return;
    xy[2 * k] = dr6[(size_t)sep_pose[k] * 6 + 3]; xy[2 * k + 1] = dr6[(size_t)sep_pose[k] * 6 + 4];
}
// DR poses, odometry measurements and initial estimate (optimizer.cpp:150-200)
__global__ __launch_bounds__(256) void pg_init_kernel(int n, const double* __restrict__ dr6, const double* __restrict__ normals, int add_noise,
                                                      pose_t* __restrict__ X, pose_t* __restrict__ meas)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double PI = DSSS_PI_REF;
    pose_t cur, prev, m;
    pose_from_rodrigues(dr6 + (size_t)i * 6, &cur);
    if (i == 0) m = cur;
    else { pose_from_rodrigues(dr6 + (size_t)(i - 1) * 6, &prev); pose_between(&prev, &cur, &m); }
    meas[i] = m;
    if (add_noise) {
        const double* z = normals + (size_t)i * 6;
        const double noise_xyz = 0.5, noise_rpy = 0.5 * PI / 180;
        const double w[3] = { z[0] * noise_rpy, z[1] * noise_rpy, z[2] * noise_rpy };
        pose_t N, o;
        so3_exp(w, N.R);
        N.t[0] = z[3] * noise_xyz; N.t[1] = z[4] * noise_xyz; N.t[2] = z[5] * noise_xyz;
        pose_compose(&cur, &N, &o);
        X[i] = o;
    } else X[i] = cur;
}

// trajectory rows "r p y x y z" of SaveTrajactoryAll (optimizer.cpp:1199-1203), computed where the poses live
__global__ __launch_bounds__(256) void pg_rpy_kernel(int n, const pose_t* __restrict__ X, double* __restrict__ rpy6)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const pose_t T = X[i];
    double rpy[3];
    pose_rpy(&T, rpy);
    double* o = rpy6 + (size_t)i * 6;
    o[0] = rpy[0]; o[1] = rpy[1]; o[2] = rpy[2]; o[3] = T.t[0]; o[4] = T.t[1]; o[5] = T.t[2];
}

// ------------------------------------------------------------------ host: ordering + symbolic analysis
namespace {

struct sym_t {
    int ns = 0;
    std::vector<int> perm;                 // chain-order separator -> elimination index
    std::vector<int> colptr, rowidx, rlptr, rlcol, rlpos, rlrow, lvptr, lvcols, diag_pos, ch_pos, lc_pos, binptr, bincols;
    std::vector<int> pan_first, pan_w, pan_lcol0, plvptr, plvpan, tlim, tfar; // panels of the top part, panel levels
    std::vector<long long> mapptr;
};

// host threads of the symbolic phase
inline int sym_threads() { static const int env = getenv("DSSS_SYM_THREADS") ? atoi(getenv("DSSS_SYM_THREADS")) : 0; if (env > 0) return env;
                          const unsigned hc = std::thread::hardware_concurrency(); return (int)std::min(4u, std::max(1u, hc)); }      // more threads do not help (serial parts dominate) and add scheduling jitter
template <class F> void par_ranges(int n, int T, F fn)          // fn(t, lo, hi) over T contiguous ranges of [0, n)
{
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { fn(t, (int)((long long)n * t / T), (int)((long long)n * (t + 1) / T)); });
    fn(0, 0, (int)((long long)n / T));
    for (auto& x : th) x.join();
}

// geometric nested dissection: recursive coordinate bisection with vertex separators taken from the lower half.
// The order of a subtree is [A subtree][B subtree][separator]; A and B never touch, so the first PG_ND_PAR levels run
// their two halves on two host threads, and the same tree of ranges later drives the parallel column-structure pass.
struct nd_tree { int a = -1, b = -1, size = 0; };          // children (indices into the node pool) or -1,-1 for a leaf
#define PG_ND_PAR 4
struct nd_ctx {
    const int* adj_ptr; const int* adj_idx; const double* cx; const double* cy; char* side; int leaf;
    std::vector<nd_tree>* pool; std::mutex* mu;
};
int nd_order(std::vector<int>& nodes, const nd_ctx& C, std::vector<int>& order, int depth)
{
    auto new_node = [&](int a, int b, int size) { std::lock_guard<std::mutex> g(*C.mu); C.pool->push_back({ a, b, size }); return (int)C.pool->size() - 1; };
    const int total = (int)nodes.size();
    if (total <= C.leaf) { std::sort(nodes.begin(), nodes.end()); for (int v : nodes) order.push_back(v); return depth <= PG_ND_PAR ? new_node(-1, -1, total) : -1; }
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int v : nodes) { x0 = std::min(x0, C.cx[v]); x1 = std::max(x1, C.cx[v]); y0 = std::min(y0, C.cy[v]); y1 = std::max(y1, C.cy[v]); }
    const bool byx = (x1 - x0) >= (y1 - y0);
    const size_t half = nodes.size() / 2;
    // split at the median of the (coordinate, index) total order; only the two halves matter, not their inner order
    const double* key = byx ? C.cx : C.cy;
    std::nth_element(nodes.begin(), nodes.begin() + half, nodes.end(), [&](int a, int b) {
        const double ka = key[a], kb = key[b];
        return ka != kb ? ka < kb : a < b; });
    for (size_t i = 0; i < nodes.size(); ++i) C.side[nodes[i]] = i < half ? 1 : 2;
    std::vector<int> A, B, S;
    for (size_t i = 0; i < half; ++i) {
        const int v = nodes[i];
        bool cut = false;
        for (int q = C.adj_ptr[v]; q < C.adj_ptr[v + 1]; ++q) if (C.side[C.adj_idx[q]] == 2) { cut = true; break; }
        (cut ? S : A).push_back(v);
    }
    B.assign(nodes.begin() + half, nodes.end());
    for (int v : nodes) C.side[v] = 0;
    std::sort(S.begin(), S.end());
    if (A.empty() || B.empty()) {          // degenerate cut: fall back to index order
        std::sort(nodes.begin(), nodes.end());
        for (int v : nodes) order.push_back(v);
        return depth <= PG_ND_PAR ? new_node(-1, -1, total) : -1;
    }
    int na = -1, nb = -1;
    if (depth < PG_ND_PAR && total > 2048) {
        std::vector<int> oa;
        std::thread th([&] { na = nd_order(A, C, oa, depth + 1); });
        std::vector<int> ob;
        nb = nd_order(B, C, ob, depth + 1);
        th.join();
        order.insert(order.end(), oa.begin(), oa.end());
        order.insert(order.end(), ob.begin(), ob.end());
    } else {
        nd_order(A, C, order, PG_ND_PAR + 1);
        nd_order(B, C, order, PG_ND_PAR + 1);
    }
    for (int v : S) order.push_back(v);
    return depth <= PG_ND_PAR ? new_node(na, nb, total) : -1;
}

// column structures of the range [lo, lo + size) of the elimination order described by tree node `t`, children merged
// into parents (elimination tree built on the fly).  A column whose parent lies outside the range hands the
// (parent, column) pair up to its caller.  No per-column allocations: the row lists of one call go into that call's pool
// (cref = pool, offset, length) and the children of a column are a linked list (kid_head / kid_next).
struct cref { int pool, off, n; };
struct cs_ctx {
    const int* adj_ptr; const int* adj_idx; const int* order; const int* perm; const std::vector<nd_tree>* pool;
    std::vector<std::vector<int>>* pools; cref* cols; int* kid_head; int* kid_next; int* parent;
};
void col_structs(const cs_ctx& C, int t, int lo, int size, std::vector<std::pair<int, int>>& up, int depth)
{
    const nd_tree nd = t >= 0 ? (*C.pool)[t] : nd_tree();
    int seq_lo = lo;
    const int hi = lo + size;
    auto add_kid = [&](int par, int j) { C.kid_next[j] = C.kid_head[par]; C.kid_head[par] = j; };
    if (t >= 0 && nd.a >= 0 && nd.b >= 0) {
        const int sa = (*C.pool)[nd.a].size, sb = (*C.pool)[nd.b].size;
        std::vector<std::pair<int, int>> ua, ub;
        std::thread th([&] { col_structs(C, nd.a, lo, sa, ua, depth + 1); });
        col_structs(C, nd.b, lo + sa, sb, ub, depth + 1);
        th.join();
        for (auto* u : { &ua, &ub })
            for (auto& e : *u) { if (e.first < hi) add_kid(e.first, e.second); else up.push_back(e); }
        seq_lo = lo + sa + sb;
    }
    const int my_pool = t >= 0 ? t : (int)C.pools->size() - 1;
    std::vector<int>& P = (*C.pools)[my_pool];
    P.reserve((size_t)(hi - seq_lo) * 24);
    std::vector<int> c;
    for (int j = seq_lo; j < hi; ++j) {
        c.clear();
        const int v = C.order[j];
        for (int q = C.adj_ptr[v]; q < C.adj_ptr[v + 1]; ++q) { const int pu = C.perm[C.adj_idx[q]]; if (pu > j) c.push_back(pu); }
        for (int k = C.kid_head[j]; k >= 0; k = C.kid_next[k]) {
            const cref ck = C.cols[k]; const int* d = (*C.pools)[ck.pool].data() + ck.off;
            for (int q = 1; q < ck.n; ++q) if (d[q] != j) c.push_back(d[q]);
        }
        std::sort(c.begin(), c.end()); c.erase(std::unique(c.begin(), c.end()), c.end());
        C.cols[j] = { my_pool, (int)P.size(), (int)c.size() + 1 };
        P.push_back(j); P.insert(P.end(), c.begin(), c.end());
        if (!c.empty()) {
            C.parent[j] = c[0];
            if (c[0] < hi) add_kid(c[0], j); else up.push_back({ c[0], j });
        }
    }
}

// edges: pairs of chain-order separator indices (chain couplings first, then LC edges)
void symbolic(int ns, const std::vector<std::pair<int, int>>& edges, int nchain, const std::vector<double>& cx,
              const std::vector<double>& cy, bool use_nd, sym_t& S)
{
    S.ns = ns;
    const bool tv = getenv("DSSS_PG_VERBOSE") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto q0 = tnow();
    // adjacency in CSR form, rows sorted and deduplicated
    std::vector<int> adj_ptr(ns + 1, 0), adj_idx;
    {
        for (auto& e : edges) if (e.first != e.second) { adj_ptr[e.first + 1]++; adj_ptr[e.second + 1]++; }
        for (int i = 0; i < ns; ++i) adj_ptr[i + 1] += adj_ptr[i];
        std::vector<int> raw(adj_ptr[ns]), fill(adj_ptr.begin(), adj_ptr.end() - 1);
        for (auto& e : edges) if (e.first != e.second) { raw[fill[e.first]++] = e.second; raw[fill[e.second]++] = e.first; }
        adj_idx.reserve(raw.size());
        std::vector<int> nptr(ns + 1, 0);
        for (int i = 0; i < ns; ++i) {
            int* b0 = raw.data() + adj_ptr[i]; int* e0 = raw.data() + adj_ptr[i + 1];
            std::sort(b0, e0); e0 = std::unique(b0, e0);
            adj_idx.insert(adj_idx.end(), b0, e0);
            nptr[i + 1] = (int)adj_idx.size();
        }
        adj_ptr.swap(nptr);
    }
    std::vector<int> order; order.reserve(ns);
    std::vector<nd_tree> pool; std::mutex mu;
    int root = -1;
    if (use_nd) {
        std::vector<int> nodes(ns); std::iota(nodes.begin(), nodes.end(), 0);
        std::vector<char> side(ns, 0);
        nd_ctx C{ adj_ptr.data(), adj_idx.data(), cx.data(), cy.data(), side.data(), 24, &pool, &mu };
        root = nd_order(nodes, C, order, 0);
    } else { order.resize(ns); std::iota(order.begin(), order.end(), 0); }
    const auto q1 = tnow();
    S.perm.assign(ns, 0);
    for (int i = 0; i < ns; ++i) S.perm[order[i]] = i;
    std::vector<cref> cols(ns);
    std::vector<int> parent(ns, -1), kid_head(ns, -1), kid_next(ns, -1);
    std::vector<std::vector<int>> pools(pool.size() + 1);
    {
        cs_ctx C{ adj_ptr.data(), adj_idx.data(), order.data(), S.perm.data(), &pool, &pools, cols.data(), kid_head.data(), kid_next.data(), parent.data() };
        std::vector<std::pair<int, int>> up;
        col_structs(C, root, 0, ns, up, 0);
    }
    const auto q2 = tnow();
    S.colptr.assign(ns + 1, 0);
    for (int j = 0; j < ns; ++j) S.colptr[j + 1] = S.colptr[j] + cols[j].n;
    auto csz = [&](int j) { return S.colptr[j + 1] - S.colptr[j]; };
    S.rowidx.resize(S.colptr[ns]);
    // row lists (transpose of the strictly lower structure), ascending k.  Threads own ascending ranges of SOURCE columns
    // with private histograms, so the entries of every target list still come out in ascending k.
    const int T = sym_threads();
    std::vector<std::vector<int>> hist(T, std::vector<int>(ns, 0));
    par_ranges(ns, T, [&](int t, int lo, int hi) {
        std::vector<int>& h = hist[t];
        for (int k = lo; k < hi; ++k) {
            const int* d = pools[cols[k].pool].data() + cols[k].off;
            std::copy(d, d + cols[k].n, S.rowidx.begin() + S.colptr[k]);
            for (int q = 1; q < cols[k].n; ++q) h[d[q]]++;
        }
    });
    S.rlptr.assign(ns + 1, 0);
    for (int j = 0; j < ns; ++j) { int tot = 0; for (int t = 0; t < T; ++t) { const int c = hist[t][j]; hist[t][j] = tot; tot += c; } S.rlptr[j + 1] = S.rlptr[j] + tot; }
    S.rlcol.resize(S.rlptr[ns]); S.rlpos.resize(S.rlptr[ns]); S.rlrow.resize(S.rlptr[ns]);
    par_ranges(ns, T, [&](int t, int lo, int hi) {
        std::vector<int>& fill = hist[t];                       // offset of this thread's first entry inside every target list
        for (int k = lo; k < hi; ++k)
            for (int q = 1; q < csz(k); ++q) {
                const int j = S.rowidx[S.colptr[k] + q], at = S.rlptr[j] + fill[j]++;
                S.rlcol[at] = k; S.rlpos[at] = S.colptr[k] + (int)q; S.rlrow[at] = j;
            }
    });
    S.mapptr.assign(ns + 1, 0);
    for (int j = 0; j < ns; ++j) S.mapptr[j + 1] = S.mapptr[j] + (long long)(S.rlptr[j + 1] - S.rlptr[j]) * (long long)csz(j);
    const auto q3 = tnow();
    // bottom subtrees -> bins (one workgroup each); the remaining "top" columns are level-scheduled
    std::vector<double> sub_cost(ns, 0);
    std::vector<char> sub_ok(ns, 0);
    static const double BIN_COST = getenv("DSSS_PG_BIN_COST") ? atof(getenv("DSSS_PG_BIN_COST")) : 1000;   // ~ update-list iterations + 20 per column; measured optimum at C3 (800-1500): about 0.2 ms of one workgroup
    for (int j = 0; j < ns; ++j) {
        const int mj = csz(j), Tj = S.rlptr[j + 1] - S.rlptr[j];
        double cst = Tj + 20.0; bool ok = mj <= 42;
        for (int k = kid_head[j]; k >= 0; k = kid_next[k]) { cst += sub_cost[k]; ok = ok && sub_ok[k]; }
        sub_cost[j] = cst; sub_ok[j] = ok && cst <= BIN_COST;
    }
    std::vector<int> root_of(ns, -1);                       // subtree root of every binned column
    for (int j = ns - 1; j >= 0; --j) {
        if (!sub_ok[j]) continue;
        const int par = parent[j];
        root_of[j] = (par >= 0 && sub_ok[par]) ? root_of[par] : j;
    }
    {   // greedy packing of whole subtrees into bins, subtrees taken in ascending root order
        std::vector<int> roots;
        for (int j = 0; j < ns; ++j) if (sub_ok[j] && root_of[j] == j) roots.push_back(j);
        std::vector<int> bin_of_root(ns, -1);
        int nbins = 0; double fill = BIN_COST + 1;
        for (int r : roots) { if (fill + sub_cost[r] > BIN_COST) { ++nbins; fill = 0; } fill += sub_cost[r]; bin_of_root[r] = nbins - 1; }
        S.binptr.assign(nbins + 1, 0);
        for (int j = 0; j < ns; ++j) if (sub_ok[j]) S.binptr[bin_of_root[root_of[j]] + 1]++;
        for (int b = 0; b < nbins; ++b) S.binptr[b + 1] += S.binptr[b];
        S.bincols.resize(S.binptr[nbins]);
        std::vector<int> fillp(S.binptr.begin(), S.binptr.end() - 1);
        for (int j = 0; j < ns; ++j) if (sub_ok[j]) S.bincols[fillp[bin_of_root[root_of[j]]]++] = j;     // ascending within a bin
    }
    // top part: supernodal panels (consecutive columns with nested structure) and their levels
    std::vector<int> col_pan(ns, -1);
    for (int j = 0; j < ns; ++j) {
        if (sub_ok[j] || col_pan[j] >= 0) continue;
        int w = 1;
        while (w < PG_PW && j + w < ns && !sub_ok[j + w] && parent[j + w - 1] == j + w && csz(j + w) + 1 == csz(j + w - 1)) ++w;
        const int p = (int)S.pan_first.size();
        S.pan_first.push_back(j); S.pan_w.push_back(w);
        for (int c = 0; c < w; ++c) col_pan[j + c] = p;
    }
    const int npan = (int)S.pan_first.size();
    std::vector<int> plevel(npan, 0);
    int maxl = -1;
    for (int p = 0; p < npan; ++p) {
        int lv = 0;
        for (int c = 0; c < S.pan_w[p]; ++c)
            for (int k = kid_head[S.pan_first[p] + c]; k >= 0; k = kid_next[k]) if (!sub_ok[k] && col_pan[k] != p) lv = std::max(lv, plevel[col_pan[k]] + 1);
        plevel[p] = lv; maxl = std::max(maxl, lv);
    }
    S.plvptr.assign(maxl + 2, 0);
    for (int p = 0; p < npan; ++p) S.plvptr[plevel[p] + 1]++;
    for (int l = 0; l <= maxl; ++l) S.plvptr[l + 1] += S.plvptr[l];
    S.plvpan.resize(npan); S.pan_lcol0.assign(npan, 0);
    { std::vector<int> fill(S.plvptr.begin(), S.plvptr.end() - 1); for (int p = 0; p < npan; ++p) S.plvpan[fill[plevel[p]]++] = p; }
    // columns of each panel level, panel by panel (kernel A works on columns); lvptr/lvcols now describe panel levels
    S.lvptr.assign(maxl + 2, 0); S.lvcols.clear();
    for (int l = 0; l <= maxl; ++l) {
        for (int q = S.plvptr[l]; q < S.plvptr[l + 1]; ++q) {
            const int p = S.plvpan[q];
            S.pan_lcol0[p] = (int)S.lvcols.size() - S.lvptr[l];
            for (int c = 0; c < S.pan_w[p]; ++c) S.lvcols.push_back(S.pan_first[p] + c);
        }
        S.lvptr[l + 1] = (int)S.lvcols.size();
    }
    // external update count per column: rowlist entries (ascending k) that lie before the column's panel
    S.tlim.assign(ns, 0);
    for (int j = 0; j < ns; ++j) {
        const int lim = col_pan[j] >= 0 ? S.pan_first[col_pan[j]] : j;
        S.tlim[j] = (int)(std::lower_bound(S.rlcol.begin() + S.rlptr[j], S.rlcol.begin() + S.rlptr[j + 1], lim) - (S.rlcol.begin() + S.rlptr[j]));
    }
    // look-ahead split of the external updates of every top column (panel level l): "far" sources were finished two
    // or more levels ago (or in the subtree bins) and are accumulated on a second stream while level l-1 is still
    // being factorised; "near" sources are the panels of level l-1.  The list is reordered [far | near | own panel].
    S.tfar.assign(ns, 0);
    par_ranges(ns, T, [&](int, int lo, int hi) {
        std::vector<std::pair<int, int>> nearv;
        for (int j = lo; j < hi; ++j) {
            if (col_pan[j] < 0) continue;
            const int lj = plevel[col_pan[j]], b0 = S.rlptr[j], Tn = S.tlim[j];
            nearv.clear();
            int wpos = b0;
            for (int t = b0; t < b0 + Tn; ++t) {
                const int k = S.rlcol[t];
                const bool isnear = col_pan[k] >= 0 && plevel[col_pan[k]] == lj - 1;
                if (isnear) nearv.push_back({ k, S.rlpos[t] });
                else { S.rlcol[wpos] = k; S.rlpos[wpos] = S.rlpos[t]; ++wpos; }
            }
            S.tfar[j] = wpos - b0;
            for (auto& e : nearv) { S.rlcol[wpos] = e.first; S.rlpos[wpos] = e.second; ++wpos; }
        }
    });
    const auto q4 = tnow();
    // where the assembled blocks go
    auto find = [&](int row, int col) { const auto b = S.rowidx.begin() + S.colptr[col], e = S.rowidx.begin() + S.colptr[col + 1];
                                        return (int)(std::lower_bound(b, e, row) - S.rowidx.begin()); };
    S.diag_pos.resize(ns);
    S.ch_pos.assign(std::max(ns - 1, 0), 0);
    par_ranges(ns, T, [&](int, int lo, int hi) {
        for (int k = lo; k < hi; ++k) {
            S.diag_pos[k] = S.colptr[S.perm[k]];
            if (k + 1 < ns) {
                const int pa = S.perm[k], pb = S.perm[k + 1];          // block S(k, k+1): rows k, cols k+1
                S.ch_pos[k] = pa > pb ? (find(pa, pb) << 1) : ((find(pb, pa) << 1) | 1);
            }
        }
    });
    if (tv) fprintf(stderr, "[dsss pg symbolic] adjacency+ND %.1f ms, column structures %.1f ms, rowlists+map ptrs %.1f ms, bins+panels %.1f ms\n", tms(q0, q1), tms(q1, q2), tms(q2, q3), tms(q3, q4));
    S.lc_pos.resize(edges.size() - nchain);
    par_ranges((int)(edges.size() - nchain), T, [&](int, int lo, int hi) {
        for (int e2 = lo; e2 < hi; ++e2) {
            const size_t e = (size_t)nchain + e2;
            const int pa = S.perm[edges[e].first], pb = S.perm[edges[e].second];   // block H(a, b)
            S.lc_pos[e2] = pa > pb ? (find(pa, pb) << 1) : ((find(pb, pa) << 1) | 1);
        }
    });
    if (tv) fprintf(stderr, "[dsss pg symbolic] positions %.1f ms\n", tms(q4, tnow()));
}

struct pg_dev {
    // device memory of one solve comes from the context's arena: a few large chunks that stay allocated between solves,
    // so a solve costs no hipMalloc / hipFree once the arena has grown to its working size
    dsss_ctx* ctx = nullptr;
    static constexpr size_t CHUNK = (size_t)256 << 20;
    template <typename T> int alloc(dsss_ctx* c, T** p, size_t n) {
        if (!ctx) { ctx = c; c->pg_chunk_cur = 0; c->pg_chunk_off = 0; }
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        for (;;) {
            if (c->pg_chunk_cur < c->pg_chunks.size()) {
                auto& ch = c->pg_chunks[c->pg_chunk_cur];
                if (c->pg_chunk_off + bytes <= ch.second) { *p = (T*)((char*)ch.first + c->pg_chunk_off); c->pg_chunk_off += bytes; return DSSS_OK; }
                ++c->pg_chunk_cur; c->pg_chunk_off = 0;
                continue;
            }
            void* q = nullptr; const size_t sz = std::max(bytes, CHUNK);
            HIPCHK(c, hipMalloc(&q, sz));
            c->pg_chunks.push_back({ q, sz });
        }
    }
    template <typename T> int upload(dsss_ctx* c, T** p, const std::vector<T>& v) { int rc = alloc(c, p, v.size()); if (rc) return rc; if (!v.empty()) HIPCHK(c, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return DSSS_OK; }
    std::vector<hipEvent_t> events;
    hipEvent_t event() { hipEvent_t e = nullptr; hipEventCreateWithFlags(&e, hipEventDisableTiming); events.push_back(e); return e; }
    void release() { if (ctx) { hipStreamSynchronize(ctx->stream); ctx->pg_chunk_cur = 0; ctx->pg_chunk_off = 0; } for (hipEvent_t e : events) if (e) hipEventDestroy(e); events.clear(); }
};

} // namespace

void dsss_pg_free(dsss_ctx* c) { for (auto& ch : c->pg_chunks) hipFree(ch.first); c->pg_chunks.clear(); c->pg_chunk_cur = 0; c->pg_chunk_off = 0; }

// batch LM over `total` poses with `ne` LC edges (host).  The DR rows (total x 6) are either one host array (dr6) or,
// with dr6 == NULL, the rows of frames 0 .. nframes-1 of the context: read on the host from the frames' pinned copies
// (only the separator poses are looked at) and gathered on the device straight from the frames' device copies.
static int pg_solve_impl(dsss_ctx* c, const double* dr6, int total, const dsss_lc_edge* edges, int ne, double* poses12, double* stats4, double* rpy6 = nullptr,
                         int nframes = 0)
{
    std::vector<int> foff;
    if (!dr6) { foff.assign(nframes + 1, 0); for (int f = 0; f < nframes; ++f) foff[f + 1] = foff[f] + c->frames[f].N; }
    const int n = total;
    if (n < 2) DSSS_FAIL(c, DSSS_E_ARG, "pose graph needs at least 2 poses");
    const auto T0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    const double PI = DSSS_PI_REF;
    pg_weights W;
    { const double wgt1 = 0.001, wgt2 = 10;                                   // optimizer.cpp:24,28
      const double so[6] = { wgt1 * PI / 180, wgt1 * PI / 180, 0.1 * wgt1 * wgt2 * PI / 180, wgt1 * wgt2, wgt1 * wgt2, wgt1 };
      for (int k = 0; k < 6; ++k) { W.prior[k] = 1.0 / 0.000001; W.odo[k] = 1.0 / so[k]; } }
    // DR poses, measurements and initial values are produced on the device (pg_init_kernel) further down
    std::vector<int> ea(ne), eb(ne); std::vector<pose_t> emeas(ne); std::vector<double> ew((size_t)ne * 6);
    for (int e = 0; e < ne; ++e) {
        ea[e] = edges[e].a; eb[e] = edges[e].b;
        if (ea[e] < 0 || ea[e] >= n || eb[e] < 0 || eb[e] >= n || ea[e] == eb[e]) DSSS_FAIL(c, DSSS_E_ARG, "LC edge %d out of range", e);
        for (int k = 0; k < 9; ++k) emeas[e].R[k] = edges[e].rel[k];
        for (int k = 0; k < 3; ++k) emeas[e].t[k] = edges[e].rel[9 + k];
        for (int k = 0; k < 6; ++k) {
            if (!(edges[e].var[k] > 0) || !std::isfinite(edges[e].var[k])) DSSS_FAIL(c, DSSS_E_NUMERIC, "LC edge %d: variance %d is not finite and positive", e, k);
            ew[(size_t)e * 6 + k] = 1.0 / std::sqrt(edges[e].var[k]);
        }
        for (int k = 0; k < 12; ++k) if (!std::isfinite(edges[e].rel[k])) DSSS_FAIL(c, DSSS_E_NUMERIC, "LC edge %d: relative pose is not finite", e);
    }
    // incidence lists (edge order), separators, segments
    std::vector<int> adj_ptr(n + 1, 0), adj_edge(2 * (size_t)ne);
    for (int e = 0; e < ne; ++e) { adj_ptr[ea[e] + 1]++; adj_ptr[eb[e] + 1]++; }
    for (int i = 0; i < n; ++i) adj_ptr[i + 1] += adj_ptr[i];
    { std::vector<int> fill(adj_ptr.begin(), adj_ptr.end() - 1);
      for (int e = 0; e < ne; ++e) { adj_edge[fill[ea[e]]++] = e << 1; adj_edge[fill[eb[e]]++] = (e << 1) | 1; } }
    std::vector<char> is_sep(n, 0);
    is_sep[0] = 1; is_sep[n - 1] = 1;
    for (int e = 0; e < ne; ++e) { is_sep[ea[e]] = 1; is_sep[eb[e]] = 1; }
    // bound the sequential depth of the per-segment block-Thomas recursion: every PG_CHUNK-th pose is promoted to a
    // separator as well (exact: it only moves that pose from the chain elimination into the sparse factorisation,
    // where a pose with two chain neighbours costs O(1) fill)
    { const char* ev = getenv("DSSS_PG_CHUNK"); const int chunk = ev ? std::max(2, atoi(ev)) : 16;
      for (int i = 0; i < n; i += chunk) is_sep[i] = 1; }
    std::vector<int> sep_pose, sidx(n, -1);
    for (int i = 0; i < n; ++i) if (is_sep[i]) { sidx[i] = (int)sep_pose.size(); sep_pose.push_back(i); }
    const int ns = (int)sep_pose.size(), nseg = ns - 1;
    std::vector<std::pair<int, int>> redges;
    for (int k = 0; k + 1 < ns; ++k) redges.push_back({ k, k + 1 });
    for (int e = 0; e < ne; ++e) redges.push_back({ sidx[ea[e]], sidx[eb[e]] });
    // device state
    pg_dev dv;
    int rc = DSSS_OK;
#define TRY(x) do { rc = (x); if (rc) { dv.release(); return rc; } } while (0)
    // DR rows on the device first: the separator coordinates for the ordering come back from there (host reads of the
    // frames' pinned copies are slow), and the copies overlap with the rest of the host preparation
    double* d_dr6; double* d_sxy; int* d_sep;
    TRY(dv.alloc(c, &d_dr6, (size_t)n * 6)); TRY(dv.alloc(c, &d_sxy, (size_t)ns * 2)); TRY(dv.upload(c, &d_sep, sep_pose));
    std::vector<double> sxy((size_t)ns * 2), cx(ns), cy(ns);
    {
        hipError_t e = hipSuccess;
        if (dr6) e = hipMemcpyAsync(d_dr6, dr6, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice, c->stream);
        else for (int f = 0; f < nframes && e == hipSuccess; ++f)
            e = hipMemcpyAsync(d_dr6 + (size_t)foff[f] * 6, c->frames[f].pose6, (size_t)c->frames[f].N * 6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) { hipLaunchKernelGGL(pg_sep_xy_kernel, dim3((ns + 255) / 256), dim3(256), 0, c->stream, ns, d_sep, d_dr6, d_sxy); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipMemcpyAsync(sxy.data(), d_sxy, sxy.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // the library's stream does not synchronise with the null stream
        if (e != hipSuccess) { dv.release(); HIPCHK(c, e); }
        for (int k = 0; k < ns; ++k) { cx[k] = sxy[2 * (size_t)k]; cy[k] = sxy[2 * (size_t)k + 1]; }
    }
    const double t_prep = ms_since(T0);
    const auto T1 = std::chrono::steady_clock::now();
    sym_t S;
    symbolic(ns, redges, nseg, cx, cy, true, S);
    const double t_sym = ms_since(T1);
    const auto T2 = std::chrono::steady_clock::now();
    const int nlev = (int)S.lvptr.size() - 1;
    const size_t nnzL = S.rowidx.size();
    std::vector<int> lv_chunks(nlev, 1), lv_far(nlev, 0), lv_near(nlev, 0), lv_slices(nlev, 1);
    size_t part2_doubles = 1;
    for (int l = 0; l < nlev; ++l) {
        int maxF = 0, maxN = 0;
        for (int q = S.lvptr[l]; q < S.lvptr[l + 1]; ++q) {
            const int j = S.lvcols[q];
            lv_chunks[l] = std::max(lv_chunks[l], (6 * (S.colptr[j + 1] - S.colptr[j]) + 255) / 256);
            maxF = std::max(maxF, S.tfar[j]); maxN = std::max(maxN, S.tlim[j] - S.tfar[j]);
        }
        lv_far[l] = maxF > 0; lv_near[l] = maxN > 0;
        const int ncl = S.lvptr[l + 1] - S.lvptr[l];
        // few wide columns near the root: slice their (far) update lists over more workgroups
        if (ncl <= 256 && maxF > 48) lv_slices[l] = std::min(ncl <= 64 ? 16 : 4, (maxF + 31) / 32);
        if (lv_slices[l] > 1) part2_doubles = std::max(part2_doubles, (size_t)ncl * lv_slices[l] * ((size_t)lv_chunks[l] * 256 * 6 + 8));
    }
    const bool verbose = getenv("DSSS_PG_VERBOSE") != nullptr;
    if (verbose) {
        int maxcol = 0; for (int j = 0; j < ns; ++j) maxcol = std::max(maxcol, S.colptr[j + 1] - S.colptr[j]);
        fprintf(stderr, "[dsss pg] poses %d  LC edges %d  separators %d  nnz(L) blocks %zu  etree levels %d  max column %d blocks  update map %lld  bins %d (%d cols)  panels %d\n", n, ne, ns, nnzL, nlev, maxcol, S.mapptr[ns], (int)S.binptr.size() - 1, (int)S.bincols.size(), (int)S.pan_first.size());
    }

    pose_t *d_X, *d_Xn, *d_meas, *d_emeas; int *d_ea, *d_eb, *d_adj_ptr, *d_adj_edge, *d_perm;
    double *d_ew, *d_r, *d_Ji, *d_D, *d_C, *d_g, *d_delta, *d_E, *d_Dl, *d_gi, *d_sDL, *d_sDR, *d_sGL, *d_sGR, *d_sS, *d_L, *d_x, *d_part, *d_scal, *d_part2;
    int *d_colptr, *d_rowidx, *d_rlptr, *d_rlcol, *d_rlpos, *d_rlrow, *d_binptr, *d_bincols, *d_pan_first, *d_pan_w, *d_pan_lcol0, *d_plvpan, *d_tlim, *d_lvcols, *d_diag, *d_ch, *d_lc, *d_fail, *d_map; long long* d_mapptr;
    const int nf = n + ne, nblk = (nf + 255) / 256;
    TRY(dv.alloc(c, &d_X, n)); TRY(dv.alloc(c, &d_Xn, n)); TRY(dv.alloc(c, &d_meas, n)); TRY(dv.upload(c, &d_emeas, emeas));
    TRY(dv.upload(c, &d_ea, ea)); TRY(dv.upload(c, &d_eb, eb)); TRY(dv.upload(c, &d_ew, ew));
    TRY(dv.upload(c, &d_adj_ptr, adj_ptr)); TRY(dv.upload(c, &d_adj_edge, adj_edge)); TRY(dv.upload(c, &d_perm, S.perm));
    TRY(dv.alloc(c, &d_r, (size_t)nf * 6)); TRY(dv.alloc(c, &d_Ji, (size_t)nf * 36));
    TRY(dv.alloc(c, &d_D, (size_t)n * 36)); TRY(dv.alloc(c, &d_C, (size_t)n * 36)); TRY(dv.alloc(c, &d_g, (size_t)n * 6)); TRY(dv.alloc(c, &d_delta, (size_t)n * 6));
    TRY(dv.alloc(c, &d_E, (size_t)n * 36)); TRY(dv.alloc(c, &d_Dl, (size_t)n * 36)); TRY(dv.alloc(c, &d_gi, (size_t)n * 6));
    TRY(dv.alloc(c, &d_sDL, (size_t)nseg * 36)); TRY(dv.alloc(c, &d_sDR, (size_t)nseg * 36)); TRY(dv.alloc(c, &d_sGL, (size_t)nseg * 6));
    TRY(dv.alloc(c, &d_sGR, (size_t)nseg * 6)); TRY(dv.alloc(c, &d_sS, (size_t)nseg * 36));
    TRY(dv.alloc(c, &d_L, nnzL * 36)); TRY(dv.alloc(c, &d_x, (size_t)ns * 6)); TRY(dv.alloc(c, &d_part, (size_t)nblk)); TRY(dv.alloc(c, &d_scal, 8)); TRY(dv.alloc(c, &d_fail, 1));
    TRY(dv.upload(c, &d_colptr, S.colptr)); TRY(dv.upload(c, &d_rowidx, S.rowidx)); TRY(dv.upload(c, &d_rlptr, S.rlptr)); TRY(dv.upload(c, &d_rlcol, S.rlcol));
    TRY(dv.upload(c, &d_rlpos, S.rlpos)); TRY(dv.upload(c, &d_lvcols, S.lvcols)); TRY(dv.upload(c, &d_diag, S.diag_pos)); TRY(dv.upload(c, &d_ch, S.ch_pos)); TRY(dv.upload(c, &d_lc, S.lc_pos));
    TRY(dv.upload(c, &d_rlrow, S.rlrow)); TRY(dv.upload(c, &d_mapptr, S.mapptr)); TRY(dv.upload(c, &d_binptr, S.binptr)); TRY(dv.upload(c, &d_bincols, S.bincols));
    const int nbins = (int)S.binptr.size() - 1;
    TRY(dv.upload(c, &d_pan_first, S.pan_first)); TRY(dv.upload(c, &d_pan_w, S.pan_w)); TRY(dv.upload(c, &d_pan_lcol0, S.pan_lcol0)); TRY(dv.upload(c, &d_plvpan, S.plvpan)); TRY(dv.upload(c, &d_tlim, S.tlim));
    std::vector<int> plv_n(nlev, 6), plv_rowchunks(nlev, 0);
    std::vector<double> fl_acc(nlev, 0), fl_diag(nlev, 0), fl_trsm(nlev, 0), fl_bwd(nlev, 0);
    for (int l = 0; l < nlev; ++l) {
        for (int q = S.lvptr[l]; q < S.lvptr[l + 1]; ++q) {
            const int j = S.lvcols[q];
            for (int t = 0; t < S.tlim[j]; ++t) { const int k = S.rlcol[S.rlptr[j] + t]; fl_acc[l] += 432.0 * (S.colptr[k + 1] - S.rlpos[S.rlptr[j] + t]); }
        }
        for (int q = S.plvptr[l]; q < S.plvptr[l + 1]; ++q) {
            const int p = S.plvpan[q], j0 = S.pan_first[p], w = S.pan_w[p]; const double nn = 6.0 * w, rows = 6.0 * (S.colptr[j0 + 1] - S.colptr[j0] - w);
            fl_diag[l] += nn * nn * nn / 3.0 + nn * nn; fl_trsm[l] += rows * nn * nn; fl_bwd[l] += 2.0 * rows * nn + nn * nn;
        }
    }
    for (int l = 0; l < nlev; ++l)
        for (int q = S.plvptr[l]; q < S.plvptr[l + 1]; ++q) {
            const int p = S.plvpan[q], j0 = S.pan_first[p], w = S.pan_w[p];
            plv_n[l] = std::max(plv_n[l], 6 * w);
            plv_rowchunks[l] = std::max(plv_rowchunks[l], (6 * (S.colptr[j0 + 1] - S.colptr[j0] - w) + 63) / 64);
        }
    const long long mapsz = S.mapptr[ns];
    if (mapsz > (1LL << 31)) { dv.release(); DSSS_FAIL(c, DSSS_E_CAPACITY, "update map of %lld entries", mapsz); }
    TRY(dv.alloc(c, &d_map, (size_t)mapsz)); TRY(dv.alloc(c, &d_part2, 2 * part2_doubles));
    int* d_tfar; TRY(dv.upload(c, &d_tfar, S.tfar));
    double *d_Wsw, *d_Wrow;      // per panel: W = L11^-1 (96 x 96, zero padded) in MFMA operand order and row-major
    { const size_t npan = S.pan_first.size(); const size_t wn = npan * (PG_PW * 6) * (PG_PW * 6); TRY(dv.alloc(c, &d_Wsw, wn)); TRY(dv.alloc(c, &d_Wrow, wn));
      HIPCHK(c, hipMemsetAsync(d_Wsw, 0, wn * sizeof(double), c->stream)); HIPCHK(c, hipMemsetAsync(d_Wrow, 0, wn * sizeof(double), c->stream)); }
    {   // pg_panel_diag_kernel keeps the packed L11 and W blocks (2 x 39 KB) in dynamic LDS
        static bool once = false;
        if (!once) {
            hipFuncSetAttribute((const void*)pg_panel_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PG_DIAG_LDS);
            hipFuncSetAttribute((const void*)pg_panel_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PG_BWD_LDS);
            (void)hipGetLastError();
            once = true;
        }
    }
    hipStream_t st = c->stream;
    const bool no_ahead = getenv("DSSS_PG_NO_LOOKAHEAD") != nullptr;
    hipEvent_t ev_bins = dv.event(), ev_far[2] = { dv.event(), dv.event() }, ev_trsm[2] = { dv.event(), dv.event() };
#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { c->err = std::string(#x) + ": " + hipGetErrorString(_e); dv.release(); return DSSS_E_HIP; } } while (0)
    auto error_of = [&](const pose_t* Xd, double* out) -> int {
        hipLaunchKernelGGL(pg_linearize_kernel, dim3(nblk), dim3(256), 0, st, n, ne, Xd, d_meas, W, d_ea, d_eb, d_emeas, d_ew, (double*)nullptr, (double*)nullptr, d_part);
        hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal);
        HCK(hipMemcpyAsync(out, d_scal, sizeof(double), hipMemcpyDeviceToHost, st));
        HCK(hipStreamSynchronize(st));
        return DSSS_OK;
    };
    {   // initial values
        double* d_norm = nullptr;
        if (c->pg.add_noise) {
            const long long need_pairs = 3LL * n;
            long long natt = (long long)(need_pairs * 1.32) + 4096;          // acceptance rate pi/4
            for (int attempt = 0;; ++attempt) {
                double* d_pairs; int* d_flags; int* d_bsum; int* d_total;
                const int nb = (int)((natt + 4095) / 4096);
                TRY(dv.alloc(c, &d_pairs, (size_t)natt * 2)); TRY(dv.alloc(c, &d_flags, (size_t)natt)); TRY(dv.alloc(c, &d_bsum, (size_t)nb)); TRY(dv.alloc(c, &d_total, 1));
                if (!d_norm) TRY(dv.alloc(c, &d_norm, (size_t)need_pairs * 2));
                const long long nthr = (natt + RNG_PER_THREAD - 1) / RNG_PER_THREAD;
                hipLaunchKernelGGL(pg_rng_attempts_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, natt, d_pairs, d_flags);
                hipLaunchKernelGGL(pg_flag_blocksum_kernel, dim3(nb), dim3(256), 0, st, d_flags, natt, d_bsum);
                hipLaunchKernelGGL(pg_flag_scan_kernel, dim3(1), dim3(256), 0, st, d_bsum, nb, d_total);
                hipLaunchKernelGGL(pg_flag_compact_kernel, dim3(nb), dim3(256), 0, st, d_flags, d_pairs, natt, d_bsum, need_pairs, d_norm);
                int total_ok = 0;
                HCK(hipMemcpyAsync(&total_ok, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
                HCK(hipStreamSynchronize(st));
                if (total_ok >= need_pairs) break;
                if (attempt > 3) { dv.release(); DSSS_FAIL(c, DSSS_E_NUMERIC, "normal generator: not enough accepted attempts"); }
                natt *= 2;
            }
        }
        hipLaunchKernelGGL(pg_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_dr6, d_norm, c->pg.add_noise, d_X, d_meas);
    }
    HCK(hipMemsetAsync(d_map, 0xff, (size_t)std::max<long long>(mapsz, 1) * sizeof(int), st));
    { const int nupd = (int)S.rlcol.size();
      if (nupd > 0) hipLaunchKernelGGL(pg_build_map_kernel, dim3((nupd + 255) / 256), dim3(256), 0, st, nupd, d_rlrow, d_rlptr, d_rlcol, d_rlpos, d_colptr, d_rowidx, d_mapptr, d_map); }
    const double t_up = ms_since(T2);
    const auto T3 = std::chrono::steady_clock::now();
    dsss_scope sc(c, DSSS_K_PG);
    double lambda = c->pg.lambda0, err = 0, err0 = 0, cur = 0;
    int iters = 0, nfact = 0;
    TRY(error_of(d_X, &err));
    err0 = err;
    if (err > 0 && c->pg.max_iters > 0) do {     // NonlinearOptimizer::defaultOptimize returns before iterating when maxIterations is reached
        cur = err;
        double oldLin = 0;
        hipLaunchKernelGGL(pg_linearize_kernel, dim3(nblk), dim3(256), 0, st, n, ne, d_X, d_meas, W, d_ea, d_eb, d_emeas, d_ew, d_r, d_Ji, d_part);
        hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal);
        HCK(hipMemcpyAsync(&oldLin, d_scal, sizeof(double), hipMemcpyDeviceToHost, st));
        for (;;) {
            // ---- solve (H + lambda I) delta = -g ; lambda lives in device memory
            HCK(hipMemcpyAsync(d_scal + 3, &lambda, sizeof(double), hipMemcpyHostToDevice, st));
            {   // (a hipGraph of this sequence costs more to instantiate than the 5 replays of one solve save: measured)
                hipMemsetAsync(d_fail, 0, sizeof(int), st);
                hipMemsetAsync(d_L, 0, nnzL * 36 * sizeof(double), st);
                hipLaunchKernelGGL(pg_assemble_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, W, d_r, d_Ji, d_adj_ptr, d_adj_edge, d_ew, d_scal + 3, d_D, d_C, d_g);
                hipLaunchKernelGGL(pg_segment_kernel, dim3((nseg + 63) / 64), dim3(64), 0, st, nseg, d_sep, d_D, d_C, d_g, d_E, d_Dl, d_gi, d_sDL, d_sDR, d_sGL, d_sGR, d_sS, d_fail);
                hipLaunchKernelGGL(pg_scatter_base_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, ns, d_sep, d_perm, d_D, d_g, d_sDL, d_sDR, d_sGL, d_sGR, d_sS, d_diag, d_ch, d_L, d_x);
                if (ne > 0) hipLaunchKernelGGL(pg_scatter_lc_kernel, dim3((ne + 255) / 256), dim3(256), 0, st, n, ne, d_Ji, d_ew, d_lc, d_L);
                if (nbins > 0) { dsss_scope s1(c, DSSS_K_PG_SUBTREE);
                                 hipLaunchKernelGGL(pg_factor_subtree_kernel, dim3(nbins), dim3(256), 0, st, d_binptr, d_bincols, d_colptr, d_rlptr, d_rlcol, d_rlpos, d_mapptr, d_map, d_L, d_x, d_fail); }
                // top of the tree: panel levels, forward substitution fused in.  Look-ahead: the far updates of level l
                // (sources finished by level l-2) run on the side stream while level l-1 is being factorised; the
                // main stream adds the near updates (level l-1) once its row solve is done.  Partial-sum buffers alternate.
                const bool ahead = !c->prof.on && !no_ahead;
                hipStream_t sf = ahead ? c->xs[0] : st;
                auto launch_far = [&](int l) {
                    if (!lv_far[l]) return;
                    const int ncl = S.lvptr[l + 1] - S.lvptr[l], stride = lv_chunks[l] * 256 * 6 + 8;
                    if (ahead) hipStreamWaitEvent(sf, l >= 2 ? ev_trsm[(l - 2) & 1] : ev_bins, 0);
                    hipLaunchKernelGGL(pg_factor_acc_kernel, dim3(ncl, lv_chunks[l], lv_slices[l]), dim3(256), 0, sf, d_lvcols + S.lvptr[l], d_colptr, d_rlptr, d_rlpos,
                                       d_mapptr, d_map, d_L, d_part2 + (size_t)(l & 1) * part2_doubles, stride, d_rlcol, d_x, (const int*)nullptr, d_tfar, (const double*)nullptr, 0);
                    if (ahead) hipEventRecord(ev_far[l & 1], sf);
                };
                if (ahead && nlev > 0) { hipEventRecord(ev_bins, st); launch_far(0); if (nlev > 1) launch_far(1); }
                for (int l = 0; l < nlev; ++l) {
                    const int ncl = S.lvptr[l + 1] - S.lvptr[l], npl = S.plvptr[l + 1] - S.plvptr[l];
                    const int stride = lv_chunks[l] * 256 * 6 + 8;
                    const int nsl = lv_far[l] ? lv_slices[l] : 1;
                    {   dsss_scope s2(c, DSSS_K_PG_ACC, fl_acc[l], (lv_far[l] ? 1 : 0) + (lv_near[l] ? 1 : 0));
                        if (!ahead) launch_far(l);
                        else if (lv_far[l]) hipStreamWaitEvent(st, ev_far[l & 1], 0);
                        // near updates in place; the same launch folds the far slices of this level (pg_fold_kernel only
                        // runs when a level has far slices but no near updates at all)
                        if (lv_near[l])
                            hipLaunchKernelGGL(pg_factor_acc_kernel, dim3(ncl, lv_chunks[l], 1), dim3(256), 0, st, d_lvcols + S.lvptr[l], d_colptr, d_rlptr, d_rlpos,
                                               d_mapptr, d_map, d_L, d_part2, stride, d_rlcol, d_x, d_tfar, d_tlim,
                                               (const double*)(d_part2 + (size_t)(l & 1) * part2_doubles), nsl > 1 ? nsl : 0);
                    }
                    if (nsl > 1 && !lv_near[l]) hipLaunchKernelGGL(pg_fold_kernel, dim3(ncl, lv_chunks[l]), dim3(256), 0, st, d_lvcols + S.lvptr[l], d_colptr, d_L, d_part2 + (size_t)(l & 1) * part2_doubles, nsl, stride, d_x);
                    { dsss_scope s3(c, DSSS_K_PG_DIAG, fl_diag[l]);
                      hipLaunchKernelGGL(pg_panel_diag_kernel, dim3(npl), dim3(256), PG_DIAG_LDS, st, d_plvpan + S.plvptr[l], d_pan_first, d_pan_w, d_colptr, d_L, d_x, d_fail, d_Wsw, d_Wrow); }
                    {   dsss_scope s4(c, DSSS_K_PG_TRSM, plv_rowchunks[l] > 0 ? fl_trsm[l] : 0.0);
                        if (plv_rowchunks[l] > 0)
                            hipLaunchKernelGGL(pg_panel_trsm_kernel, dim3(npl, plv_rowchunks[l]), dim3(256), 0, st, d_plvpan + S.plvptr[l], d_pan_first, d_pan_w, d_colptr, d_L, d_Wsw);
                    }
                    if (ahead && l + 2 < nlev) { hipEventRecord(ev_trsm[l & 1], st); launch_far(l + 2); }
                }
                for (int l = nlev - 1; l >= 0; --l) {
                    dsss_scope s5(c, DSSS_K_PG_BWD, fl_bwd[l]);
                    hipLaunchKernelGGL(pg_panel_bwd_kernel, dim3(S.plvptr[l + 1] - S.plvptr[l]), dim3(1024), PG_BWD_LDS, st, d_plvpan + S.plvptr[l], d_pan_first, d_pan_w, d_colptr, d_rowidx, d_L, d_x, d_Wrow);
                }
                if (nbins > 0) { dsss_scope s6(c, DSSS_K_PG_SUBTREE);
                    hipLaunchKernelGGL(pg_bwd_subtree_kernel, dim3(nbins), dim3(64), 0, st, d_binptr, d_bincols, d_colptr, d_rowidx, d_L, d_x); }
                hipLaunchKernelGGL(pg_sep_delta_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, ns, d_sep, d_perm, d_x, d_delta);
                hipLaunchKernelGGL(pg_backsub_kernel, dim3((nseg + 63) / 64), dim3(64), 0, st, nseg, d_sep, d_C, d_E, d_Dl, d_gi, d_delta);
                hipLaunchKernelGGL(pg_linerr_kernel, dim3(nblk), dim3(256), 0, st, n, ne, W, d_ea, d_eb, d_ew, d_r, d_Ji, d_delta, d_part);
                hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal + 1);
            }
            ++nfact;
            // X and Xn swap between trials, so these two stay outside the captured graph
            hipLaunchKernelGGL(pg_retract_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_X, d_delta, d_Xn);
            hipLaunchKernelGGL(pg_linearize_kernel, dim3(nblk), dim3(256), 0, st, n, ne, d_Xn, d_meas, W, d_ea, d_eb, d_emeas, d_ew, (double*)nullptr, (double*)nullptr, d_part);
            hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal + 2);
            HCK(hipGetLastError());
            double h[3]; int failed = 0;
            HCK(hipMemcpyAsync(h, d_scal, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
            HCK(hipMemcpyAsync(&failed, d_fail, sizeof(int), hipMemcpyDeviceToHost, st));
            HCK(hipStreamSynchronize(st));
            const bool ok = !failed && std::isfinite(h[1]);
            bool success = false, stop = false;
            double newErr = 0;
            if (ok) {
                const double linChange = oldLin - h[1];
                if (linChange >= 0) {
                    newErr = h[2];
                    const double costChange = err - newErr;
                    if (linChange > 2.220446049250313e-16 * oldLin) success = (costChange / linChange) > c->pg.min_fidelity;
                    if (std::fabs(costChange) < c->pg.rel_tol * err) stop = true;
                }
            }
            if (success) { std::swap(d_X, d_Xn); err = newErr; lambda /= c->pg.lambda_factor; ++iters; break; }
            else if (!stop) { lambda *= c->pg.lambda_factor; if (lambda >= c->pg.lambda_max) break; }
            else break;
        }
    } while (iters < c->pg.max_iters && !((err <= 0) || ((cur - err) / cur <= c->pg.rel_tol) || ((cur - err) <= c->pg.abs_tol)) && std::isfinite(cur));
    const auto T4 = std::chrono::steady_clock::now();
    const double t_lm = ms_since(T3);
    if (poses12) {      // pose_t is 12 contiguous doubles (R row-major, t): straight into the caller's buffer
        static_assert(sizeof(pose_t) == 12 * sizeof(double), "pose_t layout");
        HCK(hipMemcpyAsync(poses12, d_X, (size_t)n * sizeof(pose_t), hipMemcpyDeviceToHost, st));
    }
    if (rpy6) {
        double* d_rpy;
        TRY(dv.alloc(c, &d_rpy, (size_t)n * 6));
        hipLaunchKernelGGL(pg_rpy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_X, d_rpy);
        HCK(hipMemcpyAsync(rpy6, d_rpy, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    HCK(hipStreamSynchronize(st));
    if (stats4) { stats4[0] = iters; stats4[1] = err0; stats4[2] = err; stats4[3] = lambda; }
    dv.release();
    if (verbose) fprintf(stderr, "[dsss pg] LM iterations %d  factorisations %d  err %.6g -> %.6g | host prep %.1f ms, symbolic %.1f ms, alloc+upload %.1f ms, LM loop %.1f ms, download %.1f ms\n",
                         iters, nfact, err0, err, t_prep, t_sym, t_up, t_lm, ms_since(T4));
#undef TRY
#undef HCK
    return DSSS_OK;
}

// ------------------------------------------------------------------ LC selection (optimizer.cpp:203-258)
// For target frame t, ping j: the LAST pair (s, t) in pair order holding a kp whose target ping is j wins, and
// within it the FIRST such kp.  One 64-bit atomicMax per kp on key = (pair rank << 32) | (~index in pair).
__global__ __launch_bounds__(256) void lc_select_kernel(const double* __restrict__ kp7, int n, const int* __restrict__ kp7_pair,
                                                        const int* __restrict__ kp7_off, const int* __restrict__ act_t,
                                                        const int* __restrict__ frame_off, unsigned long long* __restrict__ slot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = kp7_pair[i];
    const int t = act_t[p];
    const int ping = (int)kp7[(size_t)i * 7 + 3];
    const unsigned long long key = ((unsigned long long)(unsigned)(p + 1) << 32) | (unsigned long long)(0xffffffffu - (unsigned)(i - kp7_off[p]));
    atomicMax(&slot[frame_off[t] + ping], key);
}

// flag of global pose g: it won a loop closure and that measurement scored > 0 (optimizer.cpp:234)
__global__ __launch_bounds__(256) void lc_edge_flag_kernel(const unsigned long long* __restrict__ slot, int total, const int* __restrict__ kp7_off,
                                                           const dsss_lc* __restrict__ lcs, int* __restrict__ flags)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const unsigned long long key = slot[g];
    int f = 0;
    if (key) {
        const int p = (int)(key >> 32) - 1, k = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
        const dsss_lc& m = lcs[kp7_off[p] + k];
        // score > 0 (optimizer.cpp:234); a non-finite score (fin == 0) or a variance that is not finite and positive
        // (the undamped 15x15 marginal failed: GTSAM would throw) drops the measurement instead of poisoning the batch LM
        f = m.score > 0 && isfinite(m.score);
        for (int q = 0; q < 6; ++q) f = f && m.var[q] > 0 && isfinite(m.var[q]);
    }
    flags[g] = f;
}
// edges in ascending target pose id (the reference's loop order), ordered compaction over blocks of 4096 poses
__global__ __launch_bounds__(256) void lc_edge_compact_kernel(const int* __restrict__ flags, const int* __restrict__ bsum, const unsigned long long* __restrict__ slot,
                                                              int total, const int* __restrict__ kp7_off, const double* __restrict__ kp7,
                                                              const dsss_lc* __restrict__ lcs, const int* __restrict__ act_s, const int* __restrict__ frame_off,
                                                              int cap, dsss_lc_edge* __restrict__ edges)
{
    __shared__ int s_w[4];
    __shared__ int s_run;
    const int i0 = blockIdx.x * 4096;
    if (threadIdx.x == 0) s_run = bsum[blockIdx.x];
    __syncthreads();
    for (int c0 = 0; c0 < 4096; c0 += 256) {
        const int g = i0 + c0 + threadIdx.x;
        const int f = g < total ? flags[g] : 0;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int inc = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        __syncthreads();
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int base = s_run;
        for (int k = 0; k < w; ++k) base += s_w[k];
        const int pos = base + inc - f;
        if (f && pos < cap) {
            const unsigned long long key = slot[g];
            const int p = (int)(key >> 32) - 1, k = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
            const int i = kp7_off[p] + k;
            dsss_lc_edge ed;
            ed.a = frame_off[act_s[p]] + (int)kp7[(size_t)i * 7 + 0];
            ed.b = g;
            for (int q = 0; q < 12; ++q) ed.rel[q] = lcs[i].rel[q];
            for (int q = 0; q < 6; ++q) ed.var[q] = lcs[i].var[q];
            edges[pos] = ed;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

extern "C" {

int dsss_posegraph_select(dsss_ctx* c, int nframes, dsss_lc_edge* edges, int cap, int* n_edges)
{
    if (!c || nframes <= 0 || nframes > c->max_frames) return DSSS_E_ARG;
    if (!c->has_lc) DSSS_FAIL(c, DSSS_E_STATE, "dsss_lc_solve_all has not run");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int> off(nframes + 1, 0);
    for (int f = 0; f < nframes; ++f) {
        if (!c->frames[f].has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no geometry", f);
        off[f + 1] = off[f] + c->frames[f].N;
    }
    const int total = off[nframes], n = c->total_kp7;
    int ne = 0;
    if (n > 0) {
        // the reference's "last pair wins" ranks pairs in (i<j) loop order == the caller's pair order, provided the
        // active pairs were listed in that order (they are: dsss_match_pairs keeps the caller's order)
        for (int p = 0; p < c->npairs; ++p)
            if (c->pair_s[p] >= nframes || c->pair_t[p] >= nframes) DSSS_FAIL(c, DSSS_E_ARG, "pair %d references a frame >= nframes", p);
        // everything on the device: winner per target pose (atomicMax), score filter, ordered compaction; only the
        // edge records come back.  Scratch comes from the solver arena (reset by the solve that follows).
        pg_dev dv;
        unsigned long long* d_slot; int *d_off, *d_flags, *d_bsum, *d_total; dsss_lc_edge* d_edges;
        const int nb = (total + 4095) / 4096;
        int rc = dv.alloc(c, &d_slot, (size_t)total); if (rc) return rc;
        rc = dv.alloc(c, &d_off, (size_t)nframes + 1); if (rc) return rc;
        rc = dv.alloc(c, &d_flags, (size_t)total); if (rc) return rc;
        rc = dv.alloc(c, &d_bsum, (size_t)nb); if (rc) return rc;
        rc = dv.alloc(c, &d_total, 1); if (rc) return rc;
        rc = dv.alloc(c, &d_edges, (size_t)cap); if (rc) return rc;
        hipStream_t st = c->stream;
        HIPCHK(c, hipMemsetAsync(d_slot, 0, (size_t)total * sizeof(unsigned long long), st));
        HIPCHK(c, hipMemcpyAsync(d_off, off.data(), (nframes + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lc_select_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->kp7, n, c->kp7_pair, c->kp7_off, c->act_t, d_off, d_slot);
        hipLaunchKernelGGL(lc_edge_flag_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d_slot, total, c->kp7_off, c->lcs, d_flags);
        hipLaunchKernelGGL(pg_flag_blocksum_kernel, dim3(nb), dim3(256), 0, st, d_flags, (long long)total, d_bsum);
        hipLaunchKernelGGL(pg_flag_scan_kernel, dim3(1), dim3(256), 0, st, d_bsum, nb, d_total);
        hipLaunchKernelGGL(lc_edge_compact_kernel, dim3(nb), dim3(256), 0, st, d_flags, d_bsum, d_slot, total, c->kp7_off, c->kp7, c->lcs, c->act_s, d_off, cap, d_edges);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&ne, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (ne > cap) { dv.release(); DSSS_FAIL(c, DSSS_E_CAPACITY, "more than %d LC edges", cap); }
        if (ne > 0) HIPCHK(c, hipMemcpy(edges, d_edges, (size_t)ne * sizeof(dsss_lc_edge), hipMemcpyDeviceToHost));
        dv.release();
    }
    if (n_edges) *n_edges = ne;
    return DSSS_OK;
}

int dsss_posegraph_solve_edges(dsss_ctx* c, const double* dr6, int total, const dsss_lc_edge* edges, int ne, double* poses12, double* stats4)
{
    if (!c || !dr6 || total <= 0 || ne < 0 || (ne > 0 && !edges)) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<double> h_dr((size_t)total * 6);
    HIPCHK(c, hipMemcpy(h_dr.data(), dr6, h_dr.size() * sizeof(double), hipMemcpyDefault));
    std::vector<dsss_lc_edge> h_e(ne);
    if (ne) HIPCHK(c, hipMemcpy(h_e.data(), edges, (size_t)ne * sizeof(dsss_lc_edge), hipMemcpyDefault));
    return pg_solve_impl(c, h_dr.data(), total, h_e.data(), ne, poses12, stats4);
}

int dsss_posegraph_solve(dsss_ctx* c, int nframes, double* poses12, double* rpy6, double* stats4)
{
    if (!c || nframes <= 0 || nframes > c->max_frames) return DSSS_E_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    size_t total = 0;
    for (int f = 0; f < nframes; ++f) {
        if (!c->frames[f].has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no geometry", f);
        total += (size_t)c->frames[f].N;
    }
    std::vector<dsss_lc_edge> edges((size_t)std::max(c->total_kp7, 1));
    int ne = 0;
    const double t_dr = ms(t0);
    const auto t1 = std::chrono::steady_clock::now();
    int rc = dsss_posegraph_select(c, nframes, edges.data(), (int)edges.size(), &ne);
    if (rc) return rc;
    const double t_sel = ms(t1);
    const auto t2 = std::chrono::steady_clock::now();
    rc = pg_solve_impl(c, nullptr, (int)total, edges.data(), ne, poses12, stats4, rpy6, nframes);
    if (rc) return rc;
    if (getenv("DSSS_PG_VERBOSE")) fprintf(stderr, "[dsss pg] DR rows %.1f ms, LC selection %.1f ms, solve + download %.1f ms\n", t_dr, t_sel, ms(t2));
    return DSSS_OK;
}

} // extern "C"
