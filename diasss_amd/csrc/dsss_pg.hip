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
//      Cholesky: nested-dissection ordering and symbolic analysis on the host (dsss_pg_sym.cpp, once per solve);
//      small subtrees column by column inside one workgroup each, everything above them MULTIFRONTAL: dense fronts,
//      extend-add of the children's update matrices, panel Cholesky / row solve / trailing update on the f64 matrix cores;
//   4. back-substitution through the segments.
#include "dsss_internal.h"
#include <utility>
#include "dsss_pose.h"
#include "dsss_pg_sym.h"
#include "dsss_pg_nd.h"
#include <algorithm>
#include <numeric>
#include <random>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <mutex>
#include <future>

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
// the same with the reciprocal of a correctly rounded square root (one sqrt and one division per pivot): for the bins, whose 17 k columns
// carry the whole dynamic range of the chain condensation -- with rsqrt here two elimination orders of the C3 graph end 1.6e-6 apart, with
// this 3e-7 (test_config_C4_full_size_8_partitions_and_2_ranks)
__device__ inline int chol6_recip(double* A, double* ri)
{
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0) || !isfinite(d)) { bad = 1; d = 1.0; }
        const double sq = sqrt(d), r = 1.0 / sq;
        A[j * 6 + j] = sq; ri[j] = r;
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
// the same with 1 / L[j][j] left ON the diagonal (what the solves multiply by): no separate reciprocal array, twelve registers less
__device__ inline int chol6_rdiag(double* A)
{
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0) || !isfinite(d)) { bad = 1; d = 1.0; }
        const double r = rsqrt(d);
        A[j * 6 + j] = r;
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

// Ownership with several ranks (dsss_comm.hip): a rank owns the poses [mp0, mp1); chain factor k belongs to the owner of pose k,
// LC edge (a, b) to the owner of its HIGHER pose eo = max(a, b) -- the rule the analysis builds the interface on (dsss_pg_sym.cpp:
// "a factor belongs to the rank of its higher pose and adds to the diagonal block of the lower one").  The pipeline's own edges
// have a < b, so eo = b there; dsss_posegraph_solve_edges also takes a > b.  Every kernel below skips what the rank does not
// own; with one rank [mp0, mp1) is everything.
__device__ inline bool pg_owned_factor(int k, int n, const int* __restrict__ eo, int mp0, int mp1)
{
    const int p = k < n ? k : eo[k - n];
    return p >= mp0 && p < mp1;
}
__global__ __launch_bounds__(256) void pg_linearize_kernel(int n, int ne, const pose_t* __restrict__ X, const pose_t* __restrict__ meas,
                                                           pg_weights W, const int* __restrict__ ea, const int* __restrict__ eb, const int* __restrict__ eo,
                                                           const pose_t* __restrict__ emeas, const double* __restrict__ ew,
                                                           double* __restrict__ r, double* __restrict__ Ji, double* __restrict__ partial, int mp0, int mp1)
{
    __shared__ double s_w[4];
    // (round 4) A thread's 36 Jacobian entries are 288 contiguous bytes and the threads of a wavefront lie 288 bytes apart: stored
    // directly, every store instruction touched 64 cache lines for 8 bytes each.  The 64 Jacobians of a wavefront are ONE contiguous
    // 18 KB range: they go through the wavefront's own slice of LDS (half a Jacobian at a time, rows padded to 19) and leave as whole lines.
    __shared__ double s_j[4][64 * 19];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double e2 = 0;
    const bool own = k < n + ne && pg_owned_factor(k, n, eo, mp0, mp1);
    double rr[6], J[36];
    if (own) {
        factor_eval(k, n, X, meas, W, ea, eb, emeas, ew, rr, Ji ? J : nullptr);
        for (int a = 0; a < 6; ++a) { e2 += rr[a] * rr[a]; if (r) r[(size_t)k * 6 + a] = rr[a]; }
    }
    if (Ji) {                                                   // (uniform over the grid)
        const unsigned long long owned = __ballot(own);
        const size_t k0 = (size_t)(blockIdx.x * 256 + wv * 64);   // first factor of this wavefront
        double* __restrict__ sj = s_j[wv];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (own) {
#pragma unroll
                for (int a = 0; a < 18; ++a) sj[lane * 19 + a] = J[18 * h + a];
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const int e = i * 64 + lane, f = e / 18, a = e - 18 * f;
                if ((owned >> f) & 1ull) Ji[(k0 + f) * 36 + 18 * h + a] = sj[f * 19 + a];
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
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
// pose's incidence list in a fixed order (no atomics).  SIX LANES PER POSE: lane a builds row a of D and C and g[a] (one
// thread per pose held 2 x 36 accumulators + a Jacobian: 280 registers, one wavefront per SIMD); the six lanes read the same
// Jacobian, which the memory pipeline broadcasts.  Every entry is summed in the order of the one-thread form.
#define PG_ASM_POSES 32                              // poses per workgroup of 192 threads
__global__ __launch_bounds__(6 * PG_ASM_POSES) void pg_assemble_kernel(int n, pg_weights W, const double* __restrict__ r, const double* __restrict__ Ji,
                                                          const int* __restrict__ adj_ptr, const int* __restrict__ adj_edge,
                                                          const double* __restrict__ ew, const double* __restrict__ lambda_ptr,
                                                          double* __restrict__ D, double* __restrict__ C, double* __restrict__ g,
                                                          const int* __restrict__ eo, int mp0, int mp1)
{
    // (a thread's six values of a block row are 48 contiguous bytes, the threads of a wavefront 48 bytes apart: stored directly, every
    // store instruction touched 24 cache lines for a sixth each.  The rows go through LDS and leave as 16-byte stores of whole lines.)
    __shared__ double s_dc[2][PG_ASM_POSES * 36];
    __shared__ double s_j[PG_ASM_POSES * 36];            // Jacobians of the chain factors i + 1 of the workgroup's poses: one contiguous 9 KB read
    const int i = blockIdx.x * PG_ASM_POSES + threadIdx.x / 6, a = threadIdx.x % 6;
    const bool live = i < n;
    {
        const size_t jb = ((size_t)blockIdx.x * PG_ASM_POSES + 1) * 36, lim = (size_t)n * 36;      // factor k lives at Ji + 36 k, k < n
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = 2 * (u * (6 * PG_ASM_POSES) + (int)threadIdx.x);
            double2 v = make_double2(0.0, 0.0);
            if (jb + e + 1 < lim) v = *reinterpret_cast<const double2*>(Ji + jb + e);
            s_j[e] = v.x; s_j[e + 1] = v.y;
        }
    }
    __syncthreads();
    double Dd[6] = { 0, 0, 0, 0, 0, 0 }, Cc[6] = { 0, 0, 0, 0, 0, 0 }, gg = 0;
    if (live) {
    const bool own_i = i >= mp0 && i < mp1, own_next = i + 1 >= mp0 && i + 1 < mp1;
    // factor i with this pose as the second variable (Jacobian W)
    const double* w2 = i == 0 ? W.prior : W.odo;
    if (own_i) {
        const double wa = w2[a];
#pragma unroll
        for (int b = 0; b < 6; ++b) if (b == a) Dd[b] += wa * wa;
        gg += wa * r[(size_t)i * 6 + a];
    }
    if (i + 1 < n && own_next) {   // factor i+1 with this pose as the first variable
        const double* J = s_j + (threadIdx.x / 6) * 36; const double* rr = r + (size_t)(i + 1) * 6;
        double ja[6], sb[6] = { 0, 0, 0, 0, 0, 0 };            // column a of J; row a of J^T J
#pragma unroll
        for (int q = 0; q < 6; ++q) ja[q] = J[q * 6 + a];
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int b = 0; b < 6; ++b) sb[b] += ja[q] * J[q * 6 + b];
#pragma unroll
        for (int b = 0; b < 6; ++b) { Dd[b] += sb[b]; Cc[b] = ja[b] * W.odo[b]; }       // C = Ji^T W
        double s = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) s += ja[q] * rr[q];
        gg += s;
    }
    for (int p = adj_ptr[i]; p < adj_ptr[i + 1]; ++p) {
        const int code = adj_edge[p], e = code >> 1, second = code & 1;
        if (eo[e] < mp0 || eo[e] >= mp1) continue;             // the edge belongs to another rank
        const double* rr = r + (size_t)(n + e) * 6;
        if (second) {
            const double wa = ew[(size_t)e * 6 + a];
#pragma unroll
            for (int b = 0; b < 6; ++b) if (b == a) Dd[b] += wa * wa;
            gg += wa * rr[a];
        } else {
            const double* J = Ji + (size_t)(n + e) * 36;
            double ja[6], sb[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < 6; ++q) ja[q] = J[q * 6 + a];
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int b = 0; b < 6; ++b) sb[b] += ja[q] * J[q * 6 + b];
#pragma unroll
            for (int b = 0; b < 6; ++b) Dd[b] += sb[b];
            double s = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) s += ja[q] * rr[q];
            gg += s;
        }
    }
    const double lambda = *lambda_ptr;
    if (own_i) {                                               // the damping of a pose is added once, by its owner
#pragma unroll
        for (int b = 0; b < 6; ++b) if (b == a) Dd[b] += lambda;
    }
    g[(size_t)i * 6 + a] = gg;
    }       // live
#pragma unroll
    for (int b = 0; b < 6; ++b) { s_dc[0][threadIdx.x * 6 + b] = Dd[b]; s_dc[1][threadIdx.x * 6 + b] = Cc[b]; }
    __syncthreads();
    {   // 32 poses x 36 doubles per array = 576 pairs of doubles: three 16-byte stores per thread and array, consecutive threads consecutive pairs
        const size_t base = (size_t)blockIdx.x * PG_ASM_POSES * 36;
        const size_t lim = (size_t)n * 36;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = 2 * (u * (6 * PG_ASM_POSES) + (int)threadIdx.x);
            if (base + e + 1 < lim) {                        // (n * 36 is even: a pair is either inside or outside)
                *reinterpret_cast<double2*>(D + base + e) = make_double2(s_dc[0][e], s_dc[0][e + 1]);
                *reinterpret_cast<double2*>(C + base + e) = make_double2(s_dc[1][e], s_dc[1][e + 1]);
            }
        }
    }
}

// Schur complement of the interior of segment s (poses L+1 .. R-1) onto its end points L, R (block Thomas recursion).
// Stores the Cholesky factor of every eliminated pivot (Dl), the fill block E_i = H(L, i) and the updated gradient for the
// back-substitution; outputs the end-point corrections.  SIXTEEN LANES PER SEGMENT (sixteen segments per workgroup): per pose the
// thirteen right-hand sides of D_i^-1 [E_i^T | C_i | g_i] go to thirteen lanes (each factorises the 6 x 6 pivot itself: cheaper
// than broadcasting the factor), and the products with E_i and C_i^T that follow are column-parallel as well.  One thread per
// segment took 24 us per pose (3 000 dependent f64 operations, 512 registers and scratch); this takes well under 1 us.
#define PG_SEG_LANES 16
struct pg_seg_lds { double E[2][36], D[2][36], G[2][6], C[36], L[36], pad[4]; };      // 232 doubles
// the sixteen lanes of a group sit in one wavefront, whose LDS operations execute in program order: waiting for the LDS queue
// (not for the global stores in flight -- a fence would) and keeping the compiler from moving memory operations across is enough
#define PG_COMPILER_FENCE() asm volatile("" ::: "memory")
#define PG_GROUP_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
__global__ __launch_bounds__(256, 2) void pg_segment_kernel(int nseg, const int* __restrict__ seg_order, const int* __restrict__ sep_pose, const double* __restrict__ D,
                                                        const double* __restrict__ C, const double* __restrict__ g,
                                                        double* __restrict__ E, double* __restrict__ Dl, double* __restrict__ gi,
                                                        double* __restrict__ segDL, double* __restrict__ segDR, double* __restrict__ segGL,
                                                        double* __restrict__ segGR, double* __restrict__ segS, int* __restrict__ fail, int mp0, int mp1)
{
    __shared__ pg_seg_lds sh_all[256 / PG_SEG_LANES];
    const int grp = threadIdx.x / PG_SEG_LANES, c = threadIdx.x % PG_SEG_LANES;
    const int slot = blockIdx.x * (256 / PG_SEG_LANES) + grp;
    if (slot >= nseg) return;                                   // whole groups leave together
    // segments are taken in descending order of length (host counting sort): the four segments of a wavefront run the same number
    // of dependent steps (lengths are spread evenly over 1..15, a wavefront of unsorted ones idles a third of its lanes) and the
    // longest start first
    const int s = seg_order[slot];
    pg_seg_lds& sh = sh_all[grp];
    const int L = sep_pose[s], R = sep_pose[s + 1];
    if (L + 1 < mp0 || L + 1 >= mp1) return;                    // a segment belongs to the owner of its poses (partitions end on a separator)
    if (R == L + 1) {
        for (int a = c; a < 36; a += PG_SEG_LANES) { segDL[(size_t)s * 36 + a] = 0; segDR[(size_t)s * 36 + a] = 0; segS[(size_t)s * 36 + a] = C[(size_t)L * 36 + a]; }
        if (c < 6) { segGL[(size_t)s * 6 + c] = 0; segGR[(size_t)s * 6 + c] = 0; }
        return;
    }
    for (int a = c; a < 36; a += PG_SEG_LANES) { sh.E[0][a] = C[(size_t)L * 36 + a]; sh.D[0][a] = D[(size_t)(L + 1) * 36 + a]; }
    if (c < 6) sh.G[0][c] = g[(size_t)(L + 1) * 6 + c];
    // Global memory is touched at ONE point of a step, its top: the blocks of the next step are requested there (nC = this lane's
    // share of C_(i+1); column c - 6 of D_(i+1) on lanes 6..11, g_(i+1) on lane 13) and consumed at the top of the next
    // step, and the records of the back-substitution (E_i, g_i, and the factor of the PREVIOUS step, parked in LDS) are stored
    // there, so the one wait on the memory counter per step finds everything a whole step old.  (Dependent loads inside the
    // step cost 8 000 of its 12 500 cycles.)  E, D, G are double-buffered in LDS: the next step's blocks are written as soon
    // as they are computed, which keeps the live registers under 168 (three wavefronts per SIMD) and saves a group sync.
    // pers[]: lanes 0..5 and 12 accumulate column c of DL / GL in it; lanes 6..11 and 13 keep their prefetched column in it.
    double nC[3], pers[6] = { 0, 0, 0, 0, 0, 0 };
    const bool is_acc = c < 6 || c == 12, is_nd = (c >= 6 && c < 12) || c == 13;
    const int nd_stride = c == 13 ? 1 : 6;
    const double* nd_src = c == 13 ? g : D + (c - 6);
    const int nd_lds = c == 13 ? (int)(&sh.G[0][0] - &sh.E[0][0]) : (int)(&sh.D[0][0] - &sh.E[0][0]) + (c - 6);     // offsets from sh.E[0] in doubles
    const int nd_flip = c == 13 ? 6 : 36;
    int cb = 0;
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int a = c + PG_SEG_LANES * u; nC[u] = a < 36 ? C[(size_t)(L + 1) * 36 + a] : 0.0; }
    for (int i = L + 1; i < R; ++i) {
        const bool last = (i + 1 == R);
        double* __restrict__ Ec = sh.E[cb]; double* __restrict__ Dc = sh.D[cb]; double* __restrict__ Gc = sh.G[cb];
        double* __restrict__ En = sh.E[cb ^ 1]; double* __restrict__ Dn = sh.D[cb ^ 1]; double* __restrict__ Gn = sh.G[cb ^ 1];
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int a = c + PG_SEG_LANES * u; if (a < 36) sh.C[a] = nC[u]; }
        if (is_nd) {                                             // completes the entries the last step left in D and G
            double* dst = &sh.E[0][0] + nd_lds + cb * nd_flip;
#pragma unroll
            for (int a = 0; a < 6; ++a) dst[a * nd_stride] += pers[a];
        }
        if (!last) {
#pragma unroll
            for (int u = 0; u < 3; ++u) { const int a = c + PG_SEG_LANES * u; if (a < 36) nC[u] = C[(size_t)(i + 1) * 36 + a]; }
            if (is_nd) {                                         // ONE load sequence for both kinds of lane: two divergent ones that
                const double* src = nd_src + (size_t)(i + 1) * (c == 13 ? 6 : 36);       // write the same registers are serialised by a full wait
#pragma unroll
                for (int a = 0; a < 6; ++a) pers[a] = src[a * nd_stride];
            }
        }
        if (i > L + 1) for (int a = c; a < 36; a += PG_SEG_LANES) Dl[(size_t)(i - 1) * 36 + a] = sh.L[a];
        PG_GROUP_SYNC();
        for (int a = c; a < 36; a += PG_SEG_LANES) E[(size_t)i * 36 + a] = Ec[a];
        if (c < 6) gi[(size_t)i * 6 + c] = Gc[c];
        // (the compiler-only barriers keep the LDS reads of the later phases from being hoisted to the top of the step)
        double Li[36], y[6];                                     // the factor with 1 / L_jj on its diagonal
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2) Li[a * 6 + b2] = b2 <= a ? Dc[a * 6 + b2] : 0.0;
        if (chol6_rdiag(Li)) { *fail = 1; return; }              // every lane of the group sees the same pivot: they leave together
        if (c == 0) {                                            // the factor, for the record (stored at the top of the next step): the
#pragma unroll
            for (int a = 0; a < 36; ++a) sh.L[a] = Li[a];        // back-substitution multiplies by its diagonal too
        }
        PG_COMPILER_FENCE();
        // right-hand side of this lane: c < 6 row c of E (column of E^T), 6 <= c < 12 column c - 6 of C, c == 12 the gradient
        {
            const double* ysrc = c < 6 ? Ec + c * 6 : c < 12 ? sh.C + (c - 6) : Gc;
            const int ystr = (c >= 6 && c < 12) ? 6 : 1;
#pragma unroll
            for (int q = 0; q < 6; ++q) y[q] = ysrc[q * ystr];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) { double t = y[a]; for (int k = 0; k < a; ++k) t -= Li[a * 6 + k] * y[k]; y[a] = t * Li[a * 7]; }
#pragma unroll
        for (int a = 5; a >= 0; --a) { double t = y[a]; for (int k = a + 1; k < 6; ++k) t -= Li[k * 6 + a] * y[k]; y[a] = t * Li[a * 7]; }
        PG_COMPILER_FENCE();
        {   // E y: column c - 6 of E_next = -E X_C on lanes 6..11; accumulated into DL / GL on lanes 0..5 and 12
            double eo[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double t = 0;
#pragma unroll
                for (int q = 0; q < 6; ++q) t += Ec[a * 6 + q] * y[q];
                eo[a] = t;
            }
            if (is_acc) {
#pragma unroll
                for (int a = 0; a < 6; ++a) pers[a] -= eo[a];
            } else if (c < 12) {
#pragma unroll
                for (int a = 0; a < 6; ++a) En[a * 6 + (c - 6)] = -eo[a];
            }
        }
        PG_COMPILER_FENCE();
        {   // C^T y: the next pivot less D_(i+1) (added at the top of the next step; the right separator's share when i + 1 == R)
            double co[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double u = 0;
#pragma unroll
                for (int q = 0; q < 6; ++q) u += sh.C[q * 6 + a] * y[q];
                co[a] = u;
            }
            if (c >= 6 && c < 12) {
#pragma unroll
                for (int a = 0; a < 6; ++a) Dn[a * 6 + (c - 6)] = -co[a];
            } else if (c == 12) {
#pragma unroll
                for (int a = 0; a < 6; ++a) Gn[a] = -co[a];
            }
        }
        if (last && is_nd) {
#pragma unroll
            for (int a = 0; a < 6; ++a) pers[a] = 0.0;
        }
        cb ^= 1;
        PG_GROUP_SYNC();
    }
    for (int a = c; a < 36; a += PG_SEG_LANES) Dl[(size_t)(R - 1) * 36 + a] = sh.L[a];
    if (c < 6) for (int a = 0; a < 6; ++a) segDL[(size_t)s * 36 + a * 6 + c] = pers[a];
    if (c == 12) for (int a = 0; a < 6; ++a) segGL[(size_t)s * 6 + a] = pers[a];
    for (int a = c; a < 36; a += PG_SEG_LANES) { segDR[(size_t)s * 36 + a] = sh.D[cb][a]; segS[(size_t)s * 36 + a] = sh.E[cb][a]; }
    if (c < 6) segGR[(size_t)s * 6 + c] = sh.G[cb][c];
}

// The same recursion on EIGHT lanes per segment (round 5; eight segments per wavefront).  With sixteen lanes a step cost the wavefront
// ~540 vector instructions -- the 6 x 6 factor (150, the same on every lane), one pair of triangular solves (42) and two products (72) --
// for FOUR segments, thirteen of sixteen lanes busy and half of the second product thrown away.  Here lane c < 6 carries TWO right-hand
// sides -- row c of E_i (its E y accumulates column c of DL) and column c of C_i (its E y is column c of E_(i+1), its C^T y column c of the
// next pivot) -- lane 6 the gradient, lane 7 only helps to move blocks: the factor is computed once per EIGHT segments, the two solves of a
// lane are independent chains (the kernel is bound by dependent f64 latency at two wavefronts per SIMD), and a step is ~520 instructions
// for eight segments.  Every right-hand side sees the arithmetic of pg_segment_kernel in the same order: the records are the same bits.
#define PG_SEG8_LANES 8
__global__ __launch_bounds__(256, 2) void pg_segment8_kernel(int nseg, const int* __restrict__ seg_order, const int* __restrict__ sep_pose, const double* __restrict__ D,
                                                         const double* __restrict__ C, const double* __restrict__ g,
                                                         double* __restrict__ E, double* __restrict__ Dl, double* __restrict__ gi,
                                                         double* __restrict__ segDL, double* __restrict__ segDR, double* __restrict__ segGL,
                                                         double* __restrict__ segGR, double* __restrict__ segS, int* __restrict__ fail, int mp0, int mp1)
{
    __shared__ pg_seg_lds sh_all[256 / PG_SEG8_LANES];
    const int grp = threadIdx.x / PG_SEG8_LANES, c = threadIdx.x % PG_SEG8_LANES;
    const int slot = blockIdx.x * (256 / PG_SEG8_LANES) + grp;
    if (slot >= nseg) return;                                   // whole groups leave together
    const int s = seg_order[slot];                              // descending length: the eight segments of a wavefront run the same number of steps
    pg_seg_lds& sh = sh_all[grp];
    const int L = sep_pose[s], R = sep_pose[s + 1];
    if (L + 1 < mp0 || L + 1 >= mp1) return;
    if (R == L + 1) {
        for (int a = c; a < 36; a += PG_SEG8_LANES) { segDL[(size_t)s * 36 + a] = 0; segDR[(size_t)s * 36 + a] = 0; segS[(size_t)s * 36 + a] = C[(size_t)L * 36 + a]; }
        if (c < 6) { segGL[(size_t)s * 6 + c] = 0; segGR[(size_t)s * 6 + c] = 0; }
        return;
    }
    for (int a = c; a < 36; a += PG_SEG8_LANES) { sh.E[0][a] = C[(size_t)L * 36 + a]; sh.D[0][a] = D[(size_t)(L + 1) * 36 + a]; }
    if (c < 6) sh.G[0][c] = g[(size_t)(L + 1) * 6 + c];
    // acc[]: column c of DL on lanes 0..5, GL on lane 6.  pre[]: the prefetched column c of D_(i+1) on lanes 0..5, g_(i+1) on lane 6 -- added
    // to the next step's pivot / gradient at its top (one load sequence for both kinds of lane, as in pg_segment_kernel).  nC: the lane's
    // share of C_(i+1).
    double nC[5], acc[6] = { 0, 0, 0, 0, 0, 0 }, pre[6] = { 0, 0, 0, 0, 0, 0 };
    const bool has_role = c < 7, is_col = c < 6;
    const int pre_stride = is_col ? 6 : 1;
    const double* pre_src = is_col ? D + c : g;
    const int pre_lds = is_col ? (int)(&sh.D[0][0] - &sh.E[0][0]) + c : (int)(&sh.G[0][0] - &sh.E[0][0]);     // offsets from sh.E[0] in doubles
    const int pre_flip = is_col ? 36 : 6;
    int cb = 0;
#pragma unroll
    for (int u = 0; u < 5; ++u) { const int a = c + PG_SEG8_LANES * u; nC[u] = a < 36 ? C[(size_t)(L + 1) * 36 + a] : 0.0; }
    for (int i = L + 1; i < R; ++i) {
        const bool last = (i + 1 == R);
        double* __restrict__ Ec = sh.E[cb]; double* __restrict__ Dc = sh.D[cb]; double* __restrict__ Gc = sh.G[cb];
        double* __restrict__ En = sh.E[cb ^ 1]; double* __restrict__ Dn = sh.D[cb ^ 1]; double* __restrict__ Gn = sh.G[cb ^ 1];
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int a = c + PG_SEG8_LANES * u; if (a < 36) sh.C[a] = nC[u]; }
        if (has_role) {                                          // completes the entries the last step left in D and G
            double* dst = &sh.E[0][0] + pre_lds + cb * pre_flip;
#pragma unroll
            for (int a = 0; a < 6; ++a) dst[a * pre_stride] += pre[a];
        }
        if (!last) {
#pragma unroll
            for (int u = 0; u < 5; ++u) { const int a = c + PG_SEG8_LANES * u; if (a < 36) nC[u] = C[(size_t)(i + 1) * 36 + a]; }
            if (has_role) {
                const double* src = pre_src + (size_t)(i + 1) * (is_col ? 36 : 6);
#pragma unroll
                for (int a = 0; a < 6; ++a) pre[a] = src[a * pre_stride];
            }
        } else if (has_role) {
#pragma unroll
            for (int a = 0; a < 6; ++a) pre[a] = 0.0;
        }
        if (i > L + 1) for (int a = c; a < 36; a += PG_SEG8_LANES) Dl[(size_t)(i - 1) * 36 + a] = sh.L[a];
        PG_GROUP_SYNC();
        for (int a = c; a < 36; a += PG_SEG8_LANES) E[(size_t)i * 36 + a] = Ec[a];
        if (c < 6) gi[(size_t)i * 6 + c] = Gc[c];
        double Li[36];                                           // the factor with 1 / L_jj on its diagonal
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2) Li[a * 6 + b2] = b2 <= a ? Dc[a * 6 + b2] : 0.0;
        if (chol6_rdiag(Li)) { *fail = 1; return; }              // every lane of the group sees the same pivot: they leave together
        if (c == 0) {
#pragma unroll
            for (int a = 0; a < 36; ++a) sh.L[a] = Li[a];
        }
        PG_COMPILER_FENCE();
        // right-hand sides of this lane: yA = row c of E (lanes 0..5) or the gradient (lanes 6, 7); yB = column c of C (lanes 0..5; the
        // gradient again on the others, unused)
        double yA[6], yB[6];
        {
            const double* ya = is_col ? Ec + c * 6 : Gc;
            const double* yb = is_col ? sh.C + c : Gc;
            const int bstr = is_col ? 6 : 1;
#pragma unroll
            for (int q = 0; q < 6; ++q) { yA[q] = ya[q]; yB[q] = yb[q * bstr]; }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double t = yA[a], u = yB[a];
            for (int k = 0; k < a; ++k) { t -= Li[a * 6 + k] * yA[k]; u -= Li[a * 6 + k] * yB[k]; }
            yA[a] = t * Li[a * 7]; yB[a] = u * Li[a * 7];
        }
#pragma unroll
        for (int a = 5; a >= 0; --a) {
            double t = yA[a], u = yB[a];
            for (int k = a + 1; k < 6; ++k) { t -= Li[k * 6 + a] * yA[k]; u -= Li[k * 6 + a] * yB[k]; }
            yA[a] = t * Li[a * 7]; yB[a] = u * Li[a * 7];
        }
        PG_COMPILER_FENCE();
        {   // E yA: accumulated into DL (lanes 0..5) / GL (lane 6).  E yB: column c of E_next = -E X_C
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double t = 0, u = 0;
#pragma unroll
                for (int q = 0; q < 6; ++q) { t += Ec[a * 6 + q] * yA[q]; u += Ec[a * 6 + q] * yB[q]; }
                acc[a] -= t;
                if (is_col) En[a * 6 + c] = -u;
            }
        }
        PG_COMPILER_FENCE();
        {   // C^T y: the next pivot less D_(i+1) from yB (lanes 0..5), the next gradient less g_(i+1) from yA (lane 6)
            double yc[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) yc[q] = is_col ? yB[q] : yA[q];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double u = 0;
#pragma unroll
                for (int q = 0; q < 6; ++q) u += sh.C[q * 6 + a] * yc[q];
                if (is_col) Dn[a * 6 + c] = -u;
                else if (c == 6) Gn[a] = -u;
            }
        }
        cb ^= 1;
        PG_GROUP_SYNC();
    }
    for (int a = c; a < 36; a += PG_SEG8_LANES) Dl[(size_t)(R - 1) * 36 + a] = sh.L[a];
    if (c < 6) for (int a = 0; a < 6; ++a) segDL[(size_t)s * 36 + a * 6 + c] = acc[a];
    if (c == 6) for (int a = 0; a < 6; ++a) segGL[(size_t)s * 6 + a] = acc[a];
    for (int a = c; a < 36; a += PG_SEG8_LANES) { segDR[(size_t)s * 36 + a] = sh.D[cb][a]; segS[(size_t)s * 36 + a] = sh.E[cb][a]; }
    if (c < 6) segGR[(size_t)s * 6 + c] = sh.G[cb][c];
}

// the level-1 chain after pass 1: diagonal block, coupling to the next entry and gradient of every chunk end / true separator
// (what pass 2 of pg_segment_kernel condenses; same meaning as D, C, g of the pose chain).  Partial sums on interface entries.
__global__ __launch_bounds__(256) void pg_chain1_kernel(int ns1, const int* __restrict__ sep1, const double* __restrict__ D, const double* __restrict__ g,
                                                        const double* __restrict__ segDL, const double* __restrict__ segDR,
                                                        const double* __restrict__ segGL, const double* __restrict__ segGR, const double* __restrict__ segS,
                                                        double* __restrict__ D1, double* __restrict__ C1, double* __restrict__ g1, int mp0, int mp1)
{
    // one thread per element (36 of D1 / C1 + 6 of g1 per node): a thread per node read its eight 288-byte rows alone (114 us)
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int k = (int)(t / 42), a = (int)(t - 42LL * k);
    if (k >= ns1) return;
    const int p = sep1[k];
    const bool segl = k > 0 && sep1[k - 1] + 1 >= mp0 && sep1[k - 1] + 1 < mp1, segr = k + 1 < ns1 && p + 1 >= mp0 && p + 1 < mp1;
    if (a < 36) {
        double v = D[(size_t)p * 36 + a];
        if (segl) v += segDR[(size_t)(k - 1) * 36 + a];
        if (segr) v += segDL[(size_t)k * 36 + a];
        D1[(size_t)k * 36 + a] = v;
        C1[(size_t)k * 36 + a] = segr ? segS[(size_t)k * 36 + a] : 0.0;
    } else {
        const int b = a - 36;
        double v = g[(size_t)p * 6 + b];
        if (segl) v += segGR[(size_t)(k - 1) * 6 + b];
        if (segr) v += segGL[(size_t)k * 6 + b];
        g1[(size_t)k * 6 + b] = v;
    }
}

// reduced system: diagonal blocks, chain couplings and right-hand side (one thread per separator, chain order).  Value index
// k = diagonal block of separator k, ns + k = chain coupling S(k, k+1), 2 ns - 1 + e = LC edge e (dsss_pg_sym.h).  A value
// whose destination column is binned goes straight into the block-sparse factor (dest >= 0: position << 1 | transpose); the
// others go, untransposed, into the value array the fronts assemble from.
__global__ __launch_bounds__(256) void pg_scatter_base_kernel(int ns, const int* __restrict__ sep_pose, const int* __restrict__ perm,
                                                              const double* __restrict__ D, const double* __restrict__ g,
                                                              const double* __restrict__ segDL, const double* __restrict__ segDR,
                                                              const double* __restrict__ segGL, const double* __restrict__ segGR,
                                                              const double* __restrict__ segS, const int* __restrict__ dest,
                                                              double* __restrict__ Lvals, double* __restrict__ aval, double* __restrict__ rhs,
                                                              const int* __restrict__ if_slot, double* __restrict__ aval_if, double* __restrict__ x_if, int mp0, int mp1)
{
    // one thread per element: 36 of the diagonal block, 6 of the right-hand side, 36 of the coupling S(k, k+1)
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int k = (int)(t / 78), el = (int)(t - 78LL * k);
    if (k >= ns) return;
    const int p = sep_pose[k];
    // segment k-1 ends in this separator, segment k starts in it; each belongs to the owner of its first interior pose
    const bool segl = k > 0 && sep_pose[k - 1] + 1 >= mp0 && sep_pose[k - 1] + 1 < mp1, segr = k + 1 < ns && p + 1 >= mp0 && p + 1 < mp1;
    const int code = dest[k];
    const bool iface = code <= -2, own = p >= mp0 && p < mp1;
    if (el < 42) {
        if (!(iface || own)) return;
        // an interface separator takes a partial sum from every rank (summed by the all-reduce), an interior one is complete
        if (el < 36) {
            double* dst = code >= 0 ? Lvals + (size_t)(code >> 1) * 36 : (iface ? aval_if + (size_t)(-2 - code) * 36 : aval + (size_t)k * 36);
            double v = D[(size_t)p * 36 + el];
            if (segl) v += segDR[(size_t)(k - 1) * 36 + el];
            if (segr) v += segDL[(size_t)k * 36 + el];
            dst[el] = v;
        } else {
            const int a = el - 36;
            double* rr = iface ? x_if + (size_t)if_slot[k] * 6 : rhs + (size_t)perm[k] * 6;
            double v = g[(size_t)p * 6 + a];
            if (segl) v += segGR[(size_t)(k - 1) * 6 + a];
            if (segr) v += segGL[(size_t)k * 6 + a];
            rr[a] = -v;
        }
    } else if (segr) {     // S(k, k+1), written by the owner of segment k: the factor holds the (larger index, smaller index) block
        const int e = el - 42, a = e / 6, b = e - 6 * a;
        const int cc = dest[ns + k];
        const double* S = segS + (size_t)k * 36;
        if (cc >= 0) Lvals[(size_t)(cc >> 1) * 36 + e] = (cc & 1) ? S[b * 6 + a] : S[e];
        else (cc <= -2 ? aval_if + (size_t)(-2 - cc) * 36 : aval + (size_t)(ns + k) * 36)[e] = S[e];
    }
}
// LC off-diagonal blocks H(a, b) = Ji^T W, added after the chain couplings.  The pipeline's per-ping selection gives every unordered
// pose pair at most one loop closure; a caller of dsss_posegraph_solve_edges may pass several (in either direction).  Those land on
// ONE block of the factor: the host chains them in edge order (lc_link[2 e] = e is the first of its group, lc_link[2 e + 1] = the next
// member or -1; NULL when no pair repeats) and the first member's threads add the whole group -- one writer per element, the sum in
// edge order whatever the number of duplicates: same bits every run.  (Round 4 added them atomically, which is order-independent for
// two addends on an empty block only.)
__global__ __launch_bounds__(256) void pg_scatter_lc_kernel(int n, int ne, int ns, const double* __restrict__ Ji, const double* __restrict__ ew,
                                                            const int* __restrict__ dest, double* __restrict__ Lvals, double* __restrict__ aval,
                                                            double* __restrict__ aval_if, const int* __restrict__ eo, int mp0, int mp1, const int* __restrict__ lc_link)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;          // one thread per element of the 6 x 6 block
    const int e = (int)(t / 36), el = (int)(t - 36LL * e);
    if (e >= ne) return;
    if (eo[e] < mp0 || eo[e] >= mp1) return;
    const int code = dest[2 * ns - 1 + e];
    if (code >= 0) {                                                        // element el of the factor's block (a group shares code >> 1)
        if (lc_link && !lc_link[2 * e]) return;
        double* dst = Lvals + (size_t)(code >> 1) * 36 + el;
        double v = *dst;
        for (int q = e; q >= 0; q = lc_link ? lc_link[2 * q + 1] : -1) {
            const int tr = dest[2 * ns - 1 + q] & 1;                        // this member's block is stored transposed
            const int a = tr ? el % 6 : el / 6, b = tr ? el / 6 : el % 6;
            v += Ji[(size_t)(n + q) * 36 + b * 6 + a] * ew[(size_t)q * 6 + b];      // (Ji^T W)(a, b)
        }
        *dst = v;
    } else {                                                                // the fronts' value array: a slot per edge, summed by the extend-add
        const int a = el / 6, b = el - 6 * a;
        (code <= -2 ? aval_if + (size_t)(-2 - code) * 36 : aval + (size_t)(2 * ns - 1 + e) * 36)[el] = Ji[(size_t)(n + e) * 36 + b * 6 + a] * ew[(size_t)e * 6 + b];
    }
}

// update matrices that cross from this rank's interior into the interface, packed (6b x 6b lower block triangle, then 6b of
// right-hand side) into the buffer the all-reduce sums; one workgroup per (child, block row)
struct pg_pack { const double* U; const double* g; double* dst; int cld, cb; };
__global__ __launch_bounds__(256) void pg_comm_pack_kernel(const int* __restrict__ it_child, const int* __restrict__ it_row, const pg_pack* __restrict__ PK)
{
    const pg_pack pk = PK[it_child[blockIdx.x]];
    const int i = it_row[blockIdx.x], b6 = 6 * pk.cb, wcols = 6 * (i + 1);
    for (int cc = threadIdx.x; cc < wcols; cc += 256) {
#pragma unroll
        for (int a = 0; a < 6; ++a) pk.dst[(size_t)(6 * i + a) * b6 + cc] = pk.U[(size_t)(6 * i + a) * pk.cld + cc];
    }
    if (threadIdx.x < 6) pk.dst[(size_t)b6 * b6 + 6 * i + threadIdx.x] = pk.g[6 * i + threadIdx.x];
}
// interface right-hand sides out of the summed buffer into the solver's vector; three scalars + the failure flag into the
// little buffer of the second all-reduce
__global__ __launch_bounds__(256) void pg_comm_xif_kernel(int nif, const int* __restrict__ if_sep, const int* __restrict__ perm, const double* __restrict__ x_if, double* __restrict__ x)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nif) return;
    for (int a = 0; a < 6; ++a) x[(size_t)perm[if_sep[q]] * 6 + a] = x_if[(size_t)q * 6 + a];
}
__global__ void pg_comm_scal_kernel(const double* __restrict__ scal, const int* __restrict__ fail, double* __restrict__ red)
{
    if (threadIdx.x < 3) red[threadIdx.x] = scal[threadIdx.x];
    if (threadIdx.x == 3) red[3] = (double)*fail;
}
__global__ __launch_bounds__(256) void pg_mask_own_kernel(int n, pose_t* __restrict__ X, int mp0, int mp1)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || (i >= mp0 && i < mp1)) return;
    for (int a = 0; a < 9; ++a) X[i].R[a] = 0.0;
    for (int a = 0; a < 3; ++a) X[i].t[a] = 0.0;
}

// The factorisation kernels below (down to pg_sep_delta_kernel) are compared with the oracle at 1e-6 on the poses, not
// bit for bit, so they may fuse multiply-adds; everything else in the library stays at -ffp-contract=off.
#pragma clang fp contract(fast)
// ---- sparse block Cholesky of the binned columns, left-looking.
// Column j holds blocks L(i, j), i in rowidx[colptr[j] .. colptr[j+1]) ascending, first the diagonal.
// rowlist(j) = binned columns k < j with L(j, k) != 0 and the position of that block.
// upd_map (built once per solve by pg_build_map_kernel): for update t of column j and target block q the position of
// L(i_q, k_t) or -1; layout [mapptr[j] + t * m_j + q], so the factor kernel has no dependent index search.
__global__ __launch_bounds__(256) void pg_build_map_kernel(int nupd, const int* __restrict__ rlrow, const int* __restrict__ rlptr,
                                                           const int* __restrict__ rlcol, const int* __restrict__ rlpos,
                                                           const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                           const long long* __restrict__ mapptr, int* __restrict__ upd_map, const int* __restrict__ nupd_dev)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= (nupd_dev ? *nupd_dev : nupd)) return;             // (lists built on the device: their total stays there, the grid covers the bound)
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

// ---- The bins' index tables built ON THE DEVICE (round 3): the update lists (per target column j the binned source columns k < j
// with L(j, k) != 0, ascending, and the position of that block), the offsets of the update map, and for the rows of a binned column
// beyond its subtree root their index in the root's boundary list.  On the host they were the last 0.9 ms of the analysis before the
// first trial could touch the bins (35 ms at the 4 M-pose graph of config 5); they are independent of everything the analysis does
// afterwards.  Same content, same order (a list is sorted by source column), so the factor is bit-identical to the host-built one
// (DSSS_PG_LISTS=host keeps the host path; the host twin of the CPU tests always uses it).
__global__ __launch_bounds__(256) void pg_rl_count_kernel(int ns, const int* __restrict__ colptr, const int* __restrict__ rowidx, const char* __restrict__ binned,
                                                          int* __restrict__ cnt)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns || !binned[k]) return;
    for (int q = colptr[k] + 1; q < colptr[k + 1]; ++q) { const int j = rowidx[q]; if (binned[j]) atomicAdd(&cnt[j], 1); }
}
// exclusive scan in three launches (1024 values per workgroup, up to 1024 x 1024 values): VAL 0 = the counts themselves (int),
// VAL 1 = count x blocks of the column (long long: the update map holds one entry per (update, target block))
template <typename T, int VAL>
__global__ __launch_bounds__(1024) void pg_scan_block_kernel(int n, const int* __restrict__ cnt, const int* __restrict__ colptr, T* __restrict__ out, T* __restrict__ block_sum)
{
    __shared__ T s_w[16];
    const int i = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T v = 0;
    if (i < n) v = VAL == 0 ? (T)cnt[i] : (T)cnt[i] * (T)(colptr[i + 1] - colptr[i]);
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const T t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const T t = s_w[k]; if (k < w) base += t; tot += t; }
    if (i < n) out[i] = base + inc - v;
    if (threadIdx.x == 0) block_sum[blockIdx.x] = tot;
}
template <typename T>
__global__ __launch_bounds__(1024) void pg_scan_tops_kernel(int nblocks, T* __restrict__ block_sum, T* __restrict__ total)
{
    __shared__ T s_w[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const T v = i < nblocks ? block_sum[i] : (T)0;
        T inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const T t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        __syncthreads();
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        T base = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const T t = s_w[k]; if (k < w) base += t; tot += t; }
        if (i < nblocks) block_sum[i] = carry + base + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}
template <typename T>
__global__ __launch_bounds__(1024) void pg_scan_add_kernel(int n, T* __restrict__ out, const T* __restrict__ block_sum, const T* __restrict__ total)
{
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += block_sum[blockIdx.x];
    if (i == 0) out[n] = *total;
}
__global__ __launch_bounds__(256) void pg_rl_fill_kernel(int ns, const int* __restrict__ colptr, const int* __restrict__ rowidx, const char* __restrict__ binned,
                                                         const int* __restrict__ rlptr, int* __restrict__ cur, int* __restrict__ rlcol, int* __restrict__ rlpos)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns || !binned[k]) return;
    for (int q = colptr[k] + 1; q < colptr[k + 1]; ++q) {
        const int j = rowidx[q];
        if (!binned[j]) continue;
        const int at = rlptr[j] + atomicAdd(&cur[j], 1);          // any order: pg_rl_sort_kernel sorts the list by source column
        rlcol[at] = k; rlpos[at] = q;
    }
}
// one wavefront per target column: its list sorted by source column (the keys are distinct: rank = number of smaller keys)
#define PG_RL_SORT_CAP 1024
__global__ __launch_bounds__(256) void pg_rl_sort_kernel(int ns, const int* __restrict__ rlptr, int* __restrict__ rlcol, int* __restrict__ rlpos, int* __restrict__ rlrow,
                                                         int* __restrict__ fail)
{
    __shared__ int s_k[4][PG_RL_SORT_CAP], s_q[4][PG_RL_SORT_CAP];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wv;
    if (j >= ns) return;
    const int b = rlptr[j], len = rlptr[j + 1] - b;
    if (len <= 0) return;
    if (len > PG_RL_SORT_CAP) { if (lane == 0) *fail = 2; return; }      // (a bin's lists sum to a few hundred entries: cannot happen; never silent)
    for (int e = lane; e < len; e += 64) { s_k[wv][e] = rlcol[b + e]; s_q[wv][e] = rlpos[b + e]; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < len; e += 64) {
        const int key = s_k[wv][e];
        int rank = 0;
        for (int f = 0; f < len; ++f) rank += s_k[wv][f] < key;
        rlcol[b + rank] = key; rlpos[b + rank] = s_q[wv][e]; rlrow[b + rank] = j;
    }
}
__global__ __launch_bounds__(256) void pg_fill_map_kernel(int* __restrict__ upd_map, const long long* __restrict__ total)
{
    const long long n = *total;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) upd_map[i] = -1;
}
__global__ __launch_bounds__(256) void pg_anc_rel_kernel(int ns, const int* __restrict__ colptr, const int* __restrict__ rowidx, const char* __restrict__ binned,
                                                         const int* __restrict__ root_of, int* __restrict__ anc_first, int* __restrict__ anc_rel)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns || !binned[k]) return;
    const int r = root_of[k], c0 = colptr[k], m = colptr[k + 1] - c0;
    int q = 0;
    while (q < m && rowidx[c0 + q] <= r) ++q;
    anc_first[k] = q;
    const int* __restrict__ rb = rowidx + colptr[r] + 1; const int nb = colptr[r + 1] - colptr[r] - 1;
    int w = 0;
    for (; q < m; ++q) { const int row = rowidx[c0 + q]; while (w < nb && rb[w] < row) ++w; anc_rel[c0 + q] = (w < nb && rb[w] == row) ? w : -1; }
}

#define PG_TCH 128
// ---- multifrontal top of the tree (dsss_pg_sym.h).  A front is a dense ld x ld lower-triangular image
//          [ F11            ]   s6 own scalar columns          assembled from the original entries + the update matrices of its
//          [ F21   F22      ]   n6 - s6 boundary rows          children (extend-add), factorised in 96-column panel steps:
//     pg_front_asm_kernel      zero + original entries + children, parent rows owned by workgroups, children in fixed order
//     pg_front_diag4_kernel    L11 = chol(A11) in 4-column pivot blocks, their inverses Linv, y = L11^-1 b      one workgroup per panel
//     pg_front_trsm2_kernel    L21 = A21 L11^-T, b2 -= L21 y                              one wavefront per 16 rows
//     pg_front_syrk_kernel     A22 -= L21 L21^T                                           64 x 64 tiles of the trailing part
//     pg_front_bwd2_kernel     x1 = L11^-T (y1 - L21^T x2)
// all dense products on v_mfma_f64_16x16x4_f64.  What is left in F22 after the last panel is the front's update matrix.
struct pg_front {
    long long off, roff;            // front / right-hand-side arena offsets (doubles)
    int ld, n6, s6, c0;             // leading dimension, scalar rows, own scalar columns, first column (elimination index)
    int rowptr, pan0;               // offset of its block-row list, first global panel id
    int ch0, ch1, fa0, fa1;         // children and original entries (CSR ranges)
};
struct pg_child { const double* U; const double* g; long long relptr; int cld, cb; };

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
#define PG_NB4 24                                   // 4-column blocks per panel
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
// The panel factorisation on FOUR wavefronts: tile column I (= tile row I of the matrix) belongs to wavefront I mod 4, so a
// wavefront holds at most nine tiles and a quarter of the updates.  Per 4-column block:
//     pivot wavefront (owner of tile (t, t)):  M44 -> Cholesky + inverse -> Linv into LDS                     barrier
//     every wavefront:  LP_I = Linv x P_I^T (one MFMA) for its tile columns I >= t, LP_I into LDS              barrier
//     every wavefront:  tile (T', I') -= LP_T' LP_I'^T for its tiles, A operand from LDS, B operand its own LP_I'
// LDS buffers alternate between blocks, so two barriers per block order everything.  The body is instantiated once per
// wavefront index, so which tiles a wavefront owns is known at compile time: straight-line code, tiles in fixed registers.
struct pg_d3_lds { double lp[2][7][64]; double li[2][16]; int bad; };
template <int W>
__device__ __forceinline__ void pg_diag3_body(pg_d3_lds& sh, double* __restrict__ A, double* __restrict__ rr, double* __restrict__ tout, int n, int ld, int l)
{
    constexpr int I0 = W, I1 = W + 4;
    constexpr bool has1 = I1 < 7;
    const int j = l & 15, q = l >> 4;
    pg_d4 S0[I0 + 1], S1[has1 ? I1 + 1 : 1];      // tiles (T, I0), T <= I0 and (T, I1), T <= min(I1, 5)
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
    __syncthreads();
    int bad = 0;
    const pg_d4 zero4 = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        if (16 * t >= n) break;                     // uniform over the workgroup: the rest is identity padding
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int buf = b & 1;
            if ((t & 3) == W) {                     // this wavefront owns tile (t, t): S0[t] for t < 4, S1[t] for t = 4, 5
                double m[10], li[10];
                double dv;
                if (t < 4) dv = S0[t < 4 ? (t <= I0 ? t : 0) : 0][b]; else dv = S1[has1 ? t : 0][b];
                int e = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c2 = 0; c2 <= r; ++c2) m[e++] = pg_readlane(dv, (4 * b + r) + 16 * c2);
                bad |= pg_chol4_inv(m, li);
                if (l == 0) {                        // Linv row-major 4 x 4: LDS for this block, global for the kernels below the panel
                    double* __restrict__ to = tout + (4 * t + b) * 16;
                    e = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c2 = 0; c2 < 4; ++c2) { const double v = c2 <= r ? li[e++] : 0.0; if (c2 <= r) sh.li[buf][r * 4 + c2] = v; to[r * 4 + c2] = v; }
                }
            }
            __syncthreads();
            const double lop = j < 4 ? sh.li[buf][j * 4 + q] : 0.0;       // A operand of Linv x P^T: lane (i, m) = Linv[i][m]
            double LP0 = 0.0, LP1 = 0.0;
            if (I0 >= t) {
                const pg_d4 r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(lop, S0[t <= I0 ? t : 0][b], zero4, 0, 0, 0);
                double v = r4[0];
                if (I0 == t) { const int rj = j - 4 * b; if (rj < 0 || (rj < 4 && q > rj)) v = 0.0; }
                LP0 = v; S0[t <= I0 ? t : 0][b] = v;
                sh.lp[buf][I0][l] = v;
            }
            if (has1 && I1 >= t) {
                const pg_d4 r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(lop, S1[has1 ? t : 0][b], zero4, 0, 0, 0);
                double v = r4[0];
                if (I1 == t) { const int rj = j - 4 * b; if (rj < 0 || (rj < 4 && q > rj)) v = 0.0; }
                LP1 = v; S1[has1 ? t : 0][b] = v;
                sh.lp[buf][I1][l] = v;
            }
            __syncthreads();
            // updates of the own tiles (T', I0), t <= T' <= I0, and (T', I1), t <= T' <= min(I1, 5); the operands come in first
            double aop[6];
#pragma unroll
            for (int T2 = t; T2 < 6; ++T2) {
                const bool need = T2 <= I0 || (has1 && T2 <= I1);
                aop[T2] = need ? -sh.lp[buf][T2][l] : 0.0;
            }
            if (j < 4 * b + 4) aop[t] = 0.0;                      // pivot tile row: only the rows below the pivot block are updated
#pragma unroll
            for (int T2 = t; T2 < 6; ++T2) {
                if (T2 <= I0) S0[T2 <= I0 ? T2 : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[T2], LP0, S0[T2 <= I0 ? T2 : 0], 0, 0, 0);
                if (has1 && T2 <= I1) S1[has1 ? T2 : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[T2], LP1, S1[has1 ? T2 : 0], 0, 0, 0);
            }
        }
    }
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
// workgroup barrier that orders LDS traffic only (__syncthreads() also waits for the global stores in flight)
#define PG_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// ---- The panel factorisation once more, as a PIPELINE (default).  In pg_diag3_body every 4-column block costs two workgroup barriers
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
// Every tile receives the same updates in the same order as in pg_diag3_body: the result is bit-identical.
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

__global__ __launch_bounds__(256) void pg_front_diag3_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD,
                                                             double* __restrict__ F, double* __restrict__ R, int* __restrict__ fail, double* __restrict__ Tinv,
                                                             unsigned long long* __restrict__ stamps)
{
    __shared__ pg_d3_lds sh;
    const unsigned long long ts0 = stamps ? __builtin_amdgcn_s_memtime() : 0;
    const pg_front fd = FD[it_front[blockIdx.x]];
    const int step = it_step[blockIdx.x], col0 = 96 * step;
    const int n = min(96, fd.s6 - col0), p = fd.pan0 + step, ld = fd.ld;
    double* __restrict__ A = F + fd.off + (size_t)col0 * ld + col0;
    double* __restrict__ rr = R + fd.roff + col0;
    double* __restrict__ tout = Tinv + (size_t)p * PG_NB4 * 16;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (threadIdx.x == 0) sh.bad = 0;
    if (threadIdx.x < 32) sh.li[threadIdx.x >> 4][threadIdx.x & 15] = 0.0;     // the zeros above the diagonal of Linv stay
    if (w == 0) pg_diag3_body<0>(sh, A, rr, tout, n, ld, l);
    else if (w == 1) pg_diag3_body<1>(sh, A, rr, tout, n, ld, l);
    else if (w == 2) pg_diag3_body<2>(sh, A, rr, tout, n, ld, l);
    else pg_diag3_body<3>(sh, A, rr, tout, n, ld, l);
    __syncthreads();
    if (sh.bad && threadIdx.x == 0) *fail = 1;
    if (stamps && threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = ts0; stamps[1] = ts0; stamps[2] = __builtin_amdgcn_s_memtime(); stamps[3] = stamps[2]; }
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
#define PG_RSU_MAX_TILES 320
#define PG_RSU32_MAX_TILES 80
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
#define PG_BWD2_LD 97
#define PG_BWD2_SX 8192                         // rows of x2 the LDS stages; taller fronts read x2 through the row map
// Tall fronts (the 4 M-pose graph of BASELINE config 5 has fronts of 7 000 rows): L21^T x2 of a panel is a 5 MB stream, and one
// workgroup pulls it through one compute unit at 30 - 50 GB/s -- 100 us and more per panel on the levels where the root front is alone.
// For panels with more than PG_BWD_SPLIT rows below them the product is split over workgroups of PG_BWD_RC rows each (this kernel:
// partial column sums, folded in slot order), and pg_front_bwd2_kernel adds the partial sums in chunk order instead of streaming L21.
#define PG_BWD_SPLIT 2048
#define PG_BWD_RC 512
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

// ---- bottom of the elimination tree: whole subtrees per workgroup (no grid-wide level barriers).
// A bin is a list of columns in ascending order whose sources all lie in the same bin, so the workgroup can run them back to
// back with workgroup barriers only (left-looking, block-sparse, update map).  Only columns with at most 42 blocks
// (6m <= 256 rows: one pass) are binned.  Every finished column also adds its outer product over the rows BEYOND its subtree
// root to the root's update matrix U_root (and L y to its right-hand side part): what the first front above the bin
// extend-adds, exactly like the F22 of a child front.  One workgroup owns a bin, columns in fixed order: deterministic.
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
__global__ __launch_bounds__(256) void pg_factor_subtree_kernel(const int* __restrict__ bin_perm, const int* __restrict__ binptr, const int* __restrict__ bincols,
                                                                const int* __restrict__ colptr, const int* __restrict__ rlptr,
                                                                const int* __restrict__ rlcol, const int* __restrict__ rlpos,
                                                                const long long* __restrict__ mapptr, const int* __restrict__ upd_map,
                                                                double* __restrict__ Lvals, double* __restrict__ x, int* __restrict__ fail,
                                                                const int* __restrict__ binroot_ptr, const int* __restrict__ binroot_idx,
                                                                const int* __restrict__ broot_b, const long long* __restrict__ broot_uoff,
                                                                const int* __restrict__ broot_of_col, const int* __restrict__ anc_first,
                                                                const int* __restrict__ anc_rel, double* __restrict__ ubin, double* __restrict__ rdiag)
{
    __shared__ double s_Ljk[PG_TCH * 36];          // update staging; reused for the column's ancestor blocks (42 x 36)
    __shared__ double s_yk[PG_TCH * 6];
    __shared__ double s_diag[36];
    __shared__ int s_arel[48];                     // boundary indices of the column's ancestor rows (at most 42)
    __shared__ double s_ri[6];                     // 1 / L(j, j)[a][a]: the solves below multiply (a dependent f64 division is ~15 instructions)
    __shared__ double s_y[6];
    __shared__ double s_xj[6];
    __shared__ int s_ok;
    const int bin = bin_perm[blockIdx.x];          // bins in descending order of work
    for (int q = binroot_ptr[bin]; q < binroot_ptr[bin + 1]; ++q) {        // zero the update matrices of this bin's roots
        const int ri = binroot_idx[q]; const long long b6 = 6LL * broot_b[ri]; double* U = ubin + broot_uoff[ri];
        for (long long e = threadIdx.x; e < b6 * b6 + b6; e += 256) U[e] = 0.0;
    }
    for (int ci = binptr[bin]; ci < binptr[bin + 1]; ++ci) {
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
        // (round 5) the updated pivot block and right-hand side reach the one-thread section below through LDS: stored to global memory
        // and read back by another thread they were a round trip through the cache hierarchy on every column's chain.  The pivot thread
        // writes both back (the factor and the solved right-hand side), as before.
        if (rhs) s_xj[rs_] = x[(size_t)j * 6 + rs_] - accy;               // (accy = 0 without updates)
        if (act && idx < 6) { for (int s = 0; s < 6; ++s) s_diag[r * 6 + s] = Lvals[(size_t)c0 * 36 + r * 6 + s] - acc[s]; }
        else if (act && T > 0) for (int s = 0; s < 6; ++s) Lvals[(size_t)(c0 + q) * 36 + r * 6 + s] -= acc[s];
        __syncthreads();
        if (threadIdx.x == 0) {
            // (round 4: in-kernel stamps put this one-thread section at 4 200 cycles per column, 28 % of the kernel -- a square root and
            // 21 dependent divisions; one reciprocal square root per pivot and multiplications by it from here on)
            double A[36], xj[6], ri[6];
#pragma unroll
            for (int a = 0; a < 36; ++a) A[a] = s_diag[a];
#pragma unroll
            for (int a = 0; a < 6; ++a) xj[a] = s_xj[a];
            const int bad = chol6_recip(A, ri);
            if (bad) *fail = 1;
            s_ok = !bad;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) { const double v = b <= a ? A[a * 6 + b] : 0.0; s_diag[a * 6 + b] = v; Lvals[(size_t)c0 * 36 + a * 6 + b] = v; }
#pragma unroll
            for (int a = 0; a < 6; ++a) { s_ri[a] = ri[a]; rdiag[(size_t)j * 6 + a] = ri[a]; }
            if (!bad) {
                double v[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    double t = xj[a];
#pragma unroll
                    for (int b = 0; b < 6; ++b) if (b < a) t -= A[a * 6 + b] * v[b];
                    v[a] = t * ri[a];
                }
#pragma unroll
                for (int a = 0; a < 6; ++a) { x[(size_t)j * 6 + a] = v[a]; s_y[a] = v[a]; }
            }
        }
        __syncthreads();
        if (!s_ok) return;
        const int af = anc_first[j], ta = m - af;                       // block rows beyond the subtree root (a suffix of the column)
        if (act && idx >= 6) {
            double* row = Lvals + (size_t)(c0 + q) * 36 + r * 6;
            double xr[6];
#pragma unroll
            for (int s = 0; s < 6; ++s) { double v = row[s];
#pragma unroll
                                          for (int c = 0; c < 6; ++c) if (c < s) v -= xr[c] * s_diag[s * 6 + c];
                                          xr[s] = v * s_ri[s]; }
            for (int s = 0; s < 6; ++s) row[s] = xr[s];
            if (q >= af) for (int s = 0; s < 6; ++s) s_Ljk[(q - af) * 36 + r * 6 + s] = xr[s];       // keep the ancestor rows for the update matrix
        }
        __syncthreads();
        const int ri = broot_of_col[j];
        if (ri >= 0 && ta > 0) {
            const int b6 = 6 * broot_b[ri];
            double* __restrict__ U = ubin + broot_uoff[ri];
            double* __restrict__ g = U + (size_t)b6 * b6;
            // (round 4) the boundary indices of the column's ancestor rows go to LDS once, and the block pairs ib <= ia are ONE flat loop:
            // per ancestor row the pass was a dependent index load, a load and a store of U in global memory, and the rows followed each
            // other (in-kernel stamps: 3 200 cycles per column, a quarter of the kernel).  Every entry of U still receives one term per column.
            if ((int)threadIdx.x < ta) s_arel[threadIdx.x] = anc_rel[c0 + af + threadIdx.x];
            __syncthreads();
            if ((int)threadIdx.x < 6 * ta) {                            // right-hand side: g[ia] -= L_a y_j
                const int pa = threadIdx.x / 6, a = threadIdx.x - pa * 6;
                const double* La = s_Ljk + pa * 36 + a * 6;
                g[s_arel[pa] * 6 + a] -= La[0] * s_y[0] + La[1] * s_y[1] + La[2] * s_y[2] + La[3] * s_y[3] + La[4] * s_y[4] + La[5] * s_y[5];
            }
            const int npair = ta * (ta + 1) / 2;                        // U[ia][ib] -= L_a L_b^T for the block pairs ib <= ia
            for (int e = threadIdx.x; e < 36 * npair; e += 256) {
                const int pr = e / 36, ab = e - 36 * pr;
                int pa = (int)((sqrtf(8.0f * (float)pr + 1.0f) - 1.0f) * 0.5f);      // pr = pa (pa + 1) / 2 + pb, pb <= pa
                while (pa * (pa + 1) / 2 > pr) --pa;
                while ((pa + 1) * (pa + 2) / 2 <= pr) ++pa;
                const int pb = pr - pa * (pa + 1) / 2, a = ab / 6, b = ab - 6 * a;
                const double* La = s_Ljk + pa * 36 + a * 6; const double* Lb = s_Ljk + pb * 36 + b * 6;
                U[(size_t)(s_arel[pa] * 6 + a) * b6 + s_arel[pb] * 6 + b] -= La[0] * Lb[0] + La[1] * Lb[1] + La[2] * Lb[2] + La[3] * Lb[3] + La[4] * Lb[4] + La[5] * Lb[5];
            }
        }
        __syncthreads();
        __threadfence_block();
    }
}
// backward substitution through a bin, columns in descending order, one wave per bin
__global__ __launch_bounds__(64) void pg_bwd_subtree_kernel(const int* __restrict__ bin_perm, const int* __restrict__ binptr, const int* __restrict__ bincols,
                                                            const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                            const double* __restrict__ Lvals, double* __restrict__ x, const double* __restrict__ rdiag)
{
    const int lane = threadIdx.x;
    const int bin = bin_perm[blockIdx.x];
    for (int ci = binptr[bin + 1] - 1; ci >= binptr[bin]; --ci) {
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
            double v[6], ld[36], xj[6], ri[6];
#pragma unroll
            for (int a = 0; a < 36; ++a) ld[a] = Ld[a];
#pragma unroll
            for (int a = 0; a < 6; ++a) { xj[a] = x[(size_t)j * 6 + a]; ri[a] = rdiag[(size_t)j * 6 + a]; }
#pragma unroll
            for (int a = 5; a >= 0; --a) { double s = xj[a] - acc[a];
#pragma unroll
                                           for (int b = 0; b < 6; ++b) if (b > a) s -= ld[b * 6 + a] * v[b];
                                           v[a] = s * ri[a]; }
#pragma unroll
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
    const int src = perm ? perm[k] : k;
    for (int a = 0; a < 6; ++a) delta[(size_t)sep_pose[k] * 6 + a] = x[(size_t)src * 6 + a];
}

// interiors, right to left: delta_i = D_i^-1 (-g_i - E_i^T delta_L - C_i delta_{i+1}).  EIGHT LANES PER SEGMENT: lane a < 6 forms
// component a of the right-hand side (column a of E_i, row a of C_i: the group reads the 288-byte blocks together; one thread
// per segment read them alone, 8.7 us per pose), the six components are exchanged, and every lane solves the 6 x 6 system itself,
// which leaves delta_i in all of them for the next step.  The blocks of pose i - 1 are requested before pose i is computed.
#define PG_BS_LANES 8
struct pg_bs_blk { double Ec[6], Cr[6], gv, Lm[21]; };
__device__ __forceinline__ void pg_bs_load(pg_bs_blk& B, int i, int aa, const double* __restrict__ C, const double* __restrict__ E,
                                           const double* __restrict__ Dl, const double* __restrict__ gi)
{
#pragma unroll
    for (int q = 0; q < 6; ++q) { B.Ec[q] = E[(size_t)i * 36 + q * 6 + aa]; B.Cr[q] = C[(size_t)i * 36 + aa * 6 + q]; }
    B.gv = gi[(size_t)i * 6 + aa];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int k = 0; k <= r; ++k) B.Lm[r * (r + 1) / 2 + k] = Dl[(size_t)i * 36 + r * 6 + k];
}
__global__ __launch_bounds__(256) void pg_backsub_kernel(int nseg, const int* __restrict__ seg_order, const int* __restrict__ sep_pose, const double* __restrict__ C,
                                                         const double* __restrict__ E, const double* __restrict__ Dl, const double* __restrict__ gi,
                                                         double* __restrict__ delta, int mp0, int mp1)
{
    const int slot = blockIdx.x * (256 / PG_BS_LANES) + threadIdx.x / PG_BS_LANES, a = threadIdx.x % PG_BS_LANES;
    if (slot >= nseg) return;                                   // whole groups leave together
    const int s = seg_order[slot];                              // descending length, as in pg_segment_kernel
    const int L = sep_pose[s], R = sep_pose[s + 1];
    if (L + 1 < mp0 || L + 1 >= mp1 || R == L + 1) return;
    const int aa = a < 6 ? a : 5;                               // lanes 6 and 7 shadow lane 5 and store nothing
    double dL[6], dn[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { dL[q] = delta[(size_t)L * 6 + q]; dn[q] = delta[(size_t)R * 6 + q]; }
    pg_bs_blk cur, nxt;
    pg_bs_load(cur, R - 1, aa, C, E, Dl, gi);
    for (int i = R - 1; i > L; --i) {
        if (i - 1 > L) pg_bs_load(nxt, i - 1, aa, C, E, Dl, gi);
        double t = -cur.gv;
#pragma unroll
        for (int q = 0; q < 6; ++q) { t -= cur.Ec[q] * dL[q]; t -= cur.Cr[q] * dn[q]; }
        double b[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) b[q] = __shfl(t, q, PG_BS_LANES);
        // (L L^T) x = b; the record holds 1 / L_jj on the diagonal (pg_segment_kernel)
#pragma unroll
        for (int r = 0; r < 6; ++r) { double v = b[r]; for (int k = 0; k < r; ++k) v -= cur.Lm[r * (r + 1) / 2 + k] * b[k]; b[r] = v * cur.Lm[r * (r + 1) / 2 + r]; }
#pragma unroll
        for (int r = 5; r >= 0; --r) { double v = b[r]; for (int k = r + 1; k < 6; ++k) v -= cur.Lm[k * (k + 1) / 2 + r] * b[k]; b[r] = v * cur.Lm[r * (r + 1) / 2 + r]; }
        const double mine = a == 0 ? b[0] : a == 1 ? b[1] : a == 2 ? b[2] : a == 3 ? b[3] : a == 4 ? b[4] : b[5];
        if (a < 6) delta[(size_t)i * 6 + a] = mine;
#pragma unroll
        for (int q = 0; q < 6; ++q) dn[q] = b[q];
        cur = nxt;
    }
}

// 0.5 * || J delta + r ||^2 over all factors (linear.error(delta))
__global__ __launch_bounds__(256) void pg_linerr_kernel(int n, int ne, pg_weights W, const int* __restrict__ ea, const int* __restrict__ eb, const int* __restrict__ eo,
                                                        const double* __restrict__ ew, const double* __restrict__ r, const double* __restrict__ Ji,
                                                        const double* __restrict__ delta, double* __restrict__ partial, int mp0, int mp1)
{
    __shared__ double s_w[4];
    // a block covers 256 factors; a thread takes residual component (factor, a) six times over, so that consecutive threads read
    // consecutive rows of the Jacobians (a thread per factor read its 288-byte block alone)
    double e2 = 0;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int el = it * 256 + threadIdx.x, k = blockIdx.x * 256 + el / 6, a = el % 6;
        if (k < n + ne && pg_owned_factor(k, n, eo, mp0, mp1)) {
            int i = -1, j; double wa;
            if (k == 0) { j = 0; wa = W.prior[a]; }
            else if (k < n) { i = k - 1; j = k; wa = W.odo[a]; }
            else { i = ea[k - n]; j = eb[k - n]; wa = ew[(size_t)(k - n) * 6 + a]; }
            double s = r[(size_t)k * 6 + a] + wa * delta[(size_t)j * 6 + a];
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
// DR rows of the frames (device copies kept by dsss_frame_set) into one array, frame after frame
__global__ __launch_bounds__(256) void pg_gather_dr_kernel(const unsigned long long* __restrict__ fptr, const int* __restrict__ foff, double* __restrict__ out)
{
    const int f = blockIdx.y;
    const double* __restrict__ src = reinterpret_cast<const double*>(fptr[f]);
    const size_t n6 = (size_t)(foff[f + 1] - foff[f]) * 6;
    double* __restrict__ dst = out + (size_t)foff[f] * 6;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n6; i += (size_t)gridDim.x * 256) dst[i] = src[i];
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

// ------------------------------------------------------------------ host: device memory of one solve
namespace {

inline int sym_threads() { const int env = getenv("DSSS_SYM_THREADS") ? atoi(getenv("DSSS_SYM_THREADS")) : 0; if (env > 0) return env;
                          const unsigned hc = std::thread::hardware_concurrency(); return (int)std::min(8u, std::max(1u, hc)); }      // ranges per parallel phase of the analysis (its worker pool has 7 threads); 16 gain another 10 % on an idle 128-core host

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
    // The tables of the analysis (some forty arrays, a few MB) go up as ONE copy: `later` books a slice of a block, `flush` copies
    // the arrays into the context's page-locked staging area on the worker pool, issues one asynchronous upload on `st` and sets the
    // device pointers.  (Forty synchronous copies from pageable memory cost 0.7 ms per solve.)  The vectors must live until flush.
    struct pend { void** p; const void* src; size_t bytes, off; };
    std::vector<pend> pending; size_t pend_total = 0;
    template <typename T> void later(T** p, const std::vector<T>& v) {
        *p = nullptr;
        pending.push_back({ (void**)p, v.data(), v.size() * sizeof(T), pend_total });
        pend_total += (std::max<size_t>(v.size(), 1) * sizeof(T) + 255) & ~(size_t)255;
    }
    hipEvent_t stage_ev = nullptr; bool stage_ev_live = false;      // recorded behind the upload of a flush: the staging area is busy until it fires
    int flush(dsss_ctx* c, hipStream_t st) {
        if (pending.empty()) return DSSS_OK;
        if (stage_ev_live) { hipEventSynchronize(stage_ev); stage_ev_live = false; }
        char* dev = nullptr;
        int rc = alloc(c, &dev, pend_total); if (rc) return rc;
        if (c->pg_stage_cap < pend_total) {
            if (c->pg_stage) hipHostFree(c->pg_stage);
            c->pg_stage = nullptr; c->pg_stage_cap = 0;
            const size_t cap = pend_total + pend_total / 4;
            HIPCHK(c, hipHostMalloc(&c->pg_stage, cap, hipHostMallocDefault));
            c->pg_stage_cap = cap;
        }
        if (stage_ev_live) { hipEventSynchronize(stage_ev); stage_ev_live = false; }      // the previous upload out of the staging area has left it
        char* stage = static_cast<char*>(c->pg_stage);
        const int T = pend_total > ((size_t)1 << 20) ? 4 : 1;
        dsss_pool_run(T, [&](int t) { for (size_t k = t; k < pending.size(); k += T) if (pending[k].bytes) memcpy(stage + pending[k].off, pending[k].src, pending[k].bytes); });
        HIPCHK(c, hipMemcpyAsync(dev, stage, pend_total, hipMemcpyHostToDevice, st));
        if (!stage_ev) stage_ev = event();
        if (stage_ev) { hipEventRecord(stage_ev, st); stage_ev_live = true; }
        for (const pend& q : pending) *q.p = dev + q.off;
        pending.clear(); pend_total = 0;
        return DSSS_OK;
    }
    std::vector<hipEvent_t> events;
    hipEvent_t event() { hipEvent_t e = nullptr; hipEventCreateWithFlags(&e, hipEventDisableTiming); events.push_back(e); return e; }
    void release() { stage_ev_live = false; stage_ev = nullptr; if (ctx) { hipStreamSynchronize(ctx->stream); ctx->pg_chunk_cur = 0; ctx->pg_chunk_off = 0; } for (hipEvent_t e : events) if (e) hipEventDestroy(e); events.clear(); pending.clear(); pend_total = 0; }
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
    // ranks: contiguous blocks of frames (of poses when the DR chain comes without frames) per partition, contiguous partitions
    // per rank.  A rank owns the poses [mp0, mp1): their chain factors, the LC edges that end in them, their segments.
    const int world = dsss_comm_world(c), rank = dsss_comm_rank(c);
    int nparts = std::max(world, c->pg_parts > 0 ? c->pg_parts : world);
    nparts = std::min(nparts, dr6 ? std::max(1, n / 4) : std::max(1, nframes));
    if (nparts < world) DSSS_FAIL(c, DSSS_E_ARG, "%d ranks need at least %d frames", world, world);
    std::vector<int> pbound(nparts + 1, n);
    for (int p = 0; p < nparts; ++p) pbound[p] = dr6 ? (int)((long long)n * p / nparts) : foff[(int)((long long)nframes * p / nparts)];
    for (int p = 0; p < nparts; ++p) if (pbound[p + 1] <= pbound[p]) DSSS_FAIL(c, DSSS_E_ARG, "empty pose-graph partition %d", p);
    const int part_lo = (int)((long long)nparts * rank / world), part_hi = (int)((long long)nparts * (rank + 1) / world);
    const int mp0 = pbound[part_lo], mp1 = pbound[part_hi];
    const auto T0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    const double PI = DSSS_PI_REF;
    pg_weights W;
    { const double wgt1 = 0.001, wgt2 = 10;                                   // optimizer.cpp:24,28
      const double so[6] = { wgt1 * PI / 180, wgt1 * PI / 180, 0.1 * wgt1 * wgt2 * PI / 180, wgt1 * wgt2, wgt1 * wgt2, wgt1 };
      for (int k = 0; k < 6; ++k) { W.prior[k] = 1.0 / 0.000001; W.odo[k] = 1.0 / so[k]; } }
    // DR poses, measurements and initial values are produced on the device (pg_init_kernel) further down
    std::vector<int> ea(ne), eb(ne), eo(ne); std::vector<pose_t> emeas; std::vector<double> ew;      // (the measurements' 1.7 MB are allocated where they are filled, beside the analysis: touching fresh pages here is time the GPU waits for)
    for (int e = 0; e < ne; ++e) {                  // the end points first: they are all the analysis needs (the measurements are unpacked beside it, below)
        ea[e] = edges[e].a; eb[e] = edges[e].b; eo[e] = std::max(edges[e].a, edges[e].b);
        if (ea[e] < 0 || ea[e] >= n || eb[e] < 0 || eb[e] >= n || ea[e] == eb[e]) DSSS_FAIL(c, DSSS_E_ARG, "LC edge %d out of range", e);
    }
    // Two levels of chain elimination.  TRUE separators (the unknowns of the sparse factorisation): LC-touched poses, the first and
    // the last pose, the last pose of every partition.  CHUNK ends: every PG_CHUNK-th pose as well, which bounds the sequential
    // depth of the per-segment block-Thomas recursion (one thread per segment).  Pass 1 condenses every chunk onto its two ends
    // (poses -> "level-1" chain of true separators + chunk ends); pass 2 condenses the runs of chunk ends between two true
    // separators the same way (same kernel, on the level-1 chain).  Exact: only the elimination order changes.
    // (All of this in time proportional to the separators, not to the poses: three passes over 400 k poses cost a millisecond of the
    // solve's serial host preparation.)  True separators are marked in a bit set and read back in order.
    const double t_p1 = ms_since(T0);
    std::vector<unsigned long long> tbits(((size_t)n + 63) / 64, 0ull);
    auto mark = [&](int i) { tbits[(size_t)i >> 6] |= 1ull << (i & 63); };
    mark(0); mark(n - 1);
    for (int e = 0; e < ne; ++e) { mark(ea[e]); mark(eb[e]); }
    for (int p = 1; p < nparts; ++p) mark(pbound[p] - 1);                // a partition ends on a separator: segments never straddle ranks
    const char* chunk_env = getenv("DSSS_PG_CHUNK"); const int chunk = chunk_env ? std::max(2, atoi(chunk_env)) : 16;
    // pass 2 is sequential over the chunk ends between two true separators: a gap of more than 16 chunks (frame boundaries
    // without keypoints reach 900 poses) gets true separators of its own, the first chunk end at least 16 chunks after the last one
    const int run = 16 * chunk;
    std::vector<int> sep1, sep_pose, t2;                                 // level-1 chain (poses); true separators (poses; positions in sep1)
    sep_pose.reserve((size_t)2 * ne + nparts + n / run + 8);
    {
        int last = 0;
        for (size_t w = 0; w < tbits.size(); ++w)
            for (unsigned long long bits = tbits[w]; bits; bits &= bits - 1) {
                const int b = (int)(w * 64) + __builtin_ctzll(bits);
                for (;;) {                                               // fill the gap (last, b)
                    const long long nx = ((long long)last + run + chunk - 1) / chunk * chunk;
                    if (nx >= b) break;
                    sep_pose.push_back((int)nx); last = (int)nx;
                }
                sep_pose.push_back(b); last = b;
            }
    }
    const double t_q1 = ms_since(T0);
    // index of a true separator among the true separators = number of marked poses below it: word prefix + popcount (a binary search
    // per loop-closure end point cost 0.6 ms)
    for (int v : sep_pose) mark(v);                                      // the gap fillers too
    std::vector<int> tpre(tbits.size() + 1, 0);
    for (size_t w = 0; w < tbits.size(); ++w) tpre[w + 1] = tpre[w] + __builtin_popcountll(tbits[w]);
    auto sidx = [&](int pose) { return tpre[(size_t)pose >> 6] + __builtin_popcountll(tbits[(size_t)pose >> 6] & ((1ull << (pose & 63)) - 1ull)); };
    const int ns = (int)sep_pose.size(), nseg = ns - 1;
    std::vector<std::pair<int, int>> redges;
    redges.reserve((size_t)ns + ne);
    const double t_q3 = ms_since(T0);
    const double t_p2 = ms_since(T0);                   // (the reduced edges themselves are written by the analysis thread, first thing)
    const auto T1 = std::chrono::steady_clock::now();
    // The analysis of the reduced system runs on a host thread of its own while this thread sets up everything that does not depend
    // on it -- device arrays of the pose chain, initial values, the first linearisation and the chain part of the first LM trial
    // (assembly, both segment passes): the device works through those while the host orders and analyses.
    const bool lists_on_device = !(getenv("DSSS_PG_LISTS") && !strcmp(getenv("DSSS_PG_LISTS"), "host"));      // A/B: the bins' update lists built by the analysis (host) instead of on the device
    std::vector<double> cx(ns), cy(ns);                       // separator coordinates: filled below, read by the analysis after its adjacency pass
    pg_sym S;
    pg_sched SO, SI;
    std::vector<int> sym_part(ns);
    for (int k = 0; k < ns; ++k) sym_part[k] = (int)(std::upper_bound(pbound.begin(), pbound.end(), sep_pose[k]) - pbound.begin()) - 1;
    std::promise<void> bottom_prom; std::future<void> bottom_fut = bottom_prom.get_future();
    std::promise<void> lists_prom; std::future<void> lists_fut = lists_prom.get_future(); bool lists_signalled = false;
    std::promise<void> coords_prom; std::future<void> coords_fut = coords_prom.get_future();
    bool bottom_signalled = false;
    // the ordering on the device (dsss_pg_nd.hip): one partition, at most 65 536 separators; DSSS_PG_ND=host keeps it on the host
    const bool dev_nd = nparts == 1 && ns >= 2 && ns <= 65536 && !(getenv("DSSS_PG_ND") && !strcmp(getenv("DSSS_PG_ND"), "host"));
    pg_nd_buffers ndB; bool nd_ok = false, nd_started = false; size_t nd_sets_cap = 0; int nd_edges_cap = 0;
    std::promise<void> nd_ready_prom; std::future<void> nd_ready_fut = nd_ready_prom.get_future();
    std::thread sym_thread([&] {
        pg_sym_opts opt; opt.threads = sym_threads();
        opt.on_bottom_ready = [&] { bottom_signalled = true; bottom_prom.set_value(); };
        opt.on_lists_ready = [&] { lists_signalled = true; lists_prom.set_value(); };
        opt.before_order = [&] { coords_fut.wait(); };
        opt.lists_on_device = lists_on_device;
        if (dev_nd) {
            opt.device_order_start = [&](const std::vector<std::pair<int, int>>& ed, int leaf, int both_axes) {
                nd_ready_fut.wait();                             // the main thread has allocated the buffers and queued the coordinates
                if (!nd_ok) return;
                static_assert(sizeof(std::pair<int, int>) == 2 * sizeof(int), "reduced edges are pairs of ints");
                if (hipSetDevice(c->device) != hipSuccess || hipStreamWaitEvent(c->pg_nd_stream, c->pg_nd_dep, 0) != hipSuccess) { (void)hipGetLastError(); return; }
                ndB.nlev = pg_nd_levels(ns, leaf);
                if (pg_nd_set_count(ndB.nlev) > nd_sets_cap || (int)ed.size() > nd_edges_cap) return;
                nd_started = pg_nd_start(c, c->pg_nd_stream, ndB, reinterpret_cast<const int*>(ed.data()), (int)ed.size(), leaf, both_axes) == DSSS_OK;
            };
            opt.device_order_finish = [&](std::vector<int>& order, std::vector<int>& top6) -> bool {
                if (!nd_started) return false;
                if (hipEventSynchronize(ndB.done) != hipSuccess) { (void)hipGetLastError(); return false; }
                if (ndB.h_sets[0].lo != 0) return false;              // (the failure flag of the level kernels)
                {   // the order must be a permutation of the separators before the symbolic phase may index with it: a position nobody
                    // filled (the levels did not suffice) or a separator placed twice sends the solve to the host ordering
                    std::vector<unsigned long long> seen(((size_t)ns + 63) / 64, 0ull);
                    for (int i = 0; i < ns; ++i) {
                        const unsigned v = (unsigned)ndB.h_order[i];
                        if (v >= (unsigned)ns) return false;
                        unsigned long long& w = seen[v >> 6]; const unsigned long long bit = 1ull << (v & 63);
                        if (w & bit) return false;
                        w |= bit;
                    }
                }
                order.assign(ndB.h_order, ndB.h_order + ns);
                top6.assign(reinterpret_cast<const int*>(ndB.h_sets), reinterpret_cast<const int*>(ndB.h_sets) + 64 * 6);
                return true;
            };
        }
        for (int k = 0; k + 1 < ns; ++k) redges.push_back({ k, k + 1 });      // the reduced graph: the chain of the separators, then the loop closures
        for (int e = 0; e < ne; ++e) redges.push_back({ sidx(ea[e]), sidx(eb[e]) });
        const double bin_cost = getenv("DSSS_PG_BIN_COST") ? atof(getenv("DSSS_PG_BIN_COST")) : 600;   // ~ update-list iterations + 20 per column; measured optimum at C3 (500-700)
        opt.bin_cost = bin_cost; opt.pack_cost = getenv("DSSS_PG_PACK_COST") ? atof(getenv("DSSS_PG_PACK_COST")) : 0; pg_sym_opts_env(opt);
        pg_symbolic(ns, redges, nseg, cx.data(), cy.data(), nparts > 1 ? sym_part.data() : nullptr, nparts, opt, S);
        // launch lists: this rank's interior fronts, then (after the all-reduce) the replicated interface fronts
        pg_build_schedule(S, part_lo, part_hi, SO);
        if (nparts > 1) pg_build_schedule(S, -1, 0, SI);
        if (!lists_signalled) lists_prom.set_value();
        if (!bottom_signalled) bottom_prom.set_value();      // (several partitions: nothing is ready early)
    });
    struct pg_joiner { std::thread& t; ~pg_joiner() { if (t.joinable()) t.join(); } } sym_join{ sym_thread };      // every return path waits for the thread before its data goes away
    struct pg_coords_guard { std::promise<void>& p; bool done = false; void set() { if (!done) { done = true; p.set_value(); } } ~pg_coords_guard() { set(); } } coords_guard{ coords_prom };
    pg_coords_guard nd_guard{ nd_ready_prom };                 // (an error exit before the buffers exist releases the analysis thread too: nd_ok stays false)      // (an error exit must not leave the thread waiting)
    // device state
    pg_dev dv;
    int rc = DSSS_OK;
    // every error exit: let the analysis thread go, wait for it AND for what it queued on the ordering's stream, and only then hand the
    // arena back (released first, the next solve could reuse memory the ordering's kernels of this one still write)
    auto abandon = [&] {
        coords_guard.set(); nd_guard.set();
        if (sym_thread.joinable()) sym_thread.join();
        if (c->pg_nd_stream && hipStreamSynchronize(c->pg_nd_stream) != hipSuccess) (void)hipGetLastError();
        dv.release();
    };
#define TRY(x) do { rc = (x); if (rc) { abandon(); return rc; } } while (0)
    // DR rows on the device first: the separator coordinates for the ordering come back from there (host reads of the
    // frames' pinned copies are slow).  The analysis thread is started BEFORE they are back and before the rest of this thread's
    // preparation (the level-1 chain of the device's chain condensation, the segment orders): it builds its adjacency first and waits for
    // the coordinates where it first needs them (pg_sym_opts::before_order).
    double* d_dr6; double* d_sxy; int* d_sep; int* d_sep1; int* d_t2; int* d_ord1; int* d_ord2;
    TRY(dv.alloc(c, &d_dr6, (size_t)n * 6)); TRY(dv.alloc(c, &d_sxy, (size_t)ns * 2)); dv.later(&d_sep, sep_pose);
    std::vector<int> ord1, ord2;                            // (alive until the second flush below)
    std::vector<unsigned long long> fp;
    unsigned long long* d_fp = nullptr; int* d_foff = nullptr;
    std::vector<double> sxy((size_t)ns * 2);
    {
        if (!dr6) {
            fp.resize(nframes);
            for (int f = 0; f < nframes; ++f) fp[f] = (unsigned long long)(uintptr_t)c->frames[f].pose6;
            dv.later(&d_fp, fp); dv.later(&d_foff, foff);
        }
        TRY(dv.flush(c, c->stream));                          // true separators, frame pointers: one upload
        hipError_t e = hipSuccess;
        if (dr6) e = hipMemcpyAsync(d_dr6, dr6, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice, c->stream);
        else {      // one gather launch over the frames' device copies (a device-to-device copy per frame cost 0.5 ms of launches at 200 frames)
            hipLaunchKernelGGL(pg_gather_dr_kernel, dim3(8, nframes), dim3(256), 0, c->stream, d_fp, d_foff, d_dr6);
            e = hipGetLastError();
        }
        if (e == hipSuccess) { hipLaunchKernelGGL(pg_sep_xy_kernel, dim3((ns + 255) / 256), dim3(256), 0, c->stream, ns, d_sep, d_dr6, d_sxy); e = hipGetLastError(); }
        if (e != hipSuccess) { abandon(); HIPCHK(c, e); }
    }
    if (dev_nd) {   // buffers of the device ordering (this thread owns the arena), the event its stream waits for, then the analysis thread may queue it
        const int ne_red = (ns - 1) + ne, nlev_max = pg_nd_levels(ns, 4);       // (levels: sized for a leaf of 4, the smallest DSSS_PG_LEAF makes sense with)
        ndB.n = ns; ndB.sxy = d_sxy;
        TRY(dv.alloc(c, &ndB.edges, (size_t)2 * ne_red)); TRY(dv.alloc(c, &ndB.deg, (size_t)ns)); TRY(dv.alloc(c, &ndB.adj_ptr, (size_t)ns + 1)); TRY(dv.alloc(c, &ndB.adj_cur, (size_t)ns));
        TRY(dv.alloc(c, &ndB.adj_idx, (size_t)2 * ne_red)); TRY(dv.alloc(c, &ndB.rank_x, (size_t)ns)); TRY(dv.alloc(c, &ndB.rank_y, (size_t)ns)); TRY(dv.alloc(c, &ndB.perm0, (size_t)ns));
        TRY(dv.alloc(c, &ndB.perm1, (size_t)ns)); TRY(dv.alloc(c, &ndB.setid, (size_t)ns)); TRY(dv.alloc(c, &ndB.order, (size_t)ns)); TRY(dv.alloc(c, &ndB.cut0, (size_t)ns)); TRY(dv.alloc(c, &ndB.cut1, (size_t)ns * 4));      // (cut1: the packed ranks, four bytes per node)
        nd_sets_cap = pg_nd_set_count(nlev_max); nd_edges_cap = ne_red;
        TRY(dv.alloc(c, &ndB.sets, nd_sets_cap + 4096));      // (+ 96 KB behind the sets: the histograms of the rank kernels, the per-set records of the big levels)
        const size_t host_ints = (size_t)ns + 64 * 6 + 16;
        if (c->pg_nd_host_cap < host_ints) {
            if (c->pg_nd_host) hipHostFree(c->pg_nd_host);
            c->pg_nd_host = nullptr; c->pg_nd_host_cap = 0;
            hipError_t e = hipHostMalloc((void**)&c->pg_nd_host, (host_ints + host_ints / 2) * sizeof(int), hipHostMallocDefault);
            if (e != hipSuccess) { abandon(); HIPCHK(c, e); }
            c->pg_nd_host_cap = host_ints + host_ints / 2;
        }
        ndB.h_sets = reinterpret_cast<pg_nd_set*>(c->pg_nd_host); ndB.h_order = c->pg_nd_host + 64 * 6 + 16;
        hipError_t e = hipSuccess;
        if (!c->pg_nd_stream) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); e = hipStreamCreateWithPriority(&c->pg_nd_stream, hipStreamNonBlocking, hi); }
        if (e == hipSuccess && !c->pg_nd_dep) e = hipEventCreateWithFlags(&c->pg_nd_dep, hipEventDisableTiming);
        if (e == hipSuccess && !c->pg_nd_done) e = hipEventCreateWithFlags(&c->pg_nd_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(c->pg_nd_dep, c->stream);        // behind pg_sep_xy_kernel
        if (e != hipSuccess) { abandon(); HIPCHK(c, e); }
        ndB.done = c->pg_nd_done;
        nd_ok = true;
    }
    nd_guard.set();
    const double t_prep0 = ms_since(T0);
    {   // the coordinates come back while the analysis builds its adjacency: hand them over
        hipError_t e = hipMemcpyAsync(sxy.data(), d_sxy, sxy.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // the library's stream does not synchronise with the null stream
        if (e != hipSuccess) { abandon(); HIPCHK(c, e); }
        for (int k = 0; k < ns; ++k) { cx[k] = sxy[2 * (size_t)k]; cy[k] = sxy[2 * (size_t)k + 1]; }
        coords_guard.set();
    }
    emeas.resize(ne); ew.resize((size_t)ne * 6);
    for (int e = 0; e < ne; ++e) {                  // measurements and weights of the loop closures (the analysis is running and has its coordinates)
        for (int k = 0; k < 9; ++k) emeas[e].R[k] = edges[e].rel[k];
        for (int k = 0; k < 3; ++k) emeas[e].t[k] = edges[e].rel[9 + k];
        for (int k = 0; k < 6; ++k) {
            if (!(edges[e].var[k] > 0) || !std::isfinite(edges[e].var[k])) { abandon(); DSSS_FAIL(c, DSSS_E_NUMERIC, "LC edge %d: variance %d is not finite and positive", e, k); }
            ew[(size_t)e * 6 + k] = 1.0 / std::sqrt(edges[e].var[k]);
        }
        for (int k = 0; k < 12; ++k) if (!std::isfinite(edges[e].rel[k])) { abandon(); DSSS_FAIL(c, DSSS_E_NUMERIC, "LC edge %d: relative pose is not finite", e); }
    }
    // level-1 chain = true separators merged with the chunk ends 0, chunk, 2 chunk ...; segment orders: only the device reads them
    const double t_m0 = ms_since(T0);
    sep1.reserve(sep_pose.size() + n / chunk + 2); t2.reserve(sep_pose.size());
    for (size_t it = 0, m = 0; it < sep_pose.size() || m < (size_t)n;) {
        const long long a = it < sep_pose.size() ? sep_pose[it] : (1LL << 40), bm = m < (size_t)n ? (long long)m : (1LL << 40);
        if (a <= bm) { t2.push_back((int)sep1.size()); sep1.push_back((int)a); ++it; if (a == bm) m += chunk; }
        else { sep1.push_back((int)bm); m += chunk; }
    }
    const double t_q2 = ms_since(T0) - t_m0;
    const int ns1 = (int)sep1.size(), nseg1 = ns1 - 1;
    // this rank's range of the level-1 chain (its poses are [mp0, mp1))
    const int kp0 = (int)(std::lower_bound(sep1.begin(), sep1.end(), mp0) - sep1.begin()), kp1 = (int)(std::lower_bound(sep1.begin(), sep1.end(), mp1) - sep1.begin());
    dv.later(&d_sep1, sep1); dv.later(&d_t2, t2);
    {   // segments of both passes in descending order of length (stable counting sort: ties stay in chain order)
        auto by_length = [](const std::vector<int>& ends, std::vector<int>& ord) {
            const int m = (int)ends.size() - 1;
            ord.resize(std::max(m, 1));
            int maxlen = 0;
            for (int k = 0; k < m; ++k) maxlen = std::max(maxlen, ends[k + 1] - ends[k]);
            std::vector<int> cnt(maxlen + 2, 0);
            for (int k = 0; k < m; ++k) cnt[maxlen - (ends[k + 1] - ends[k]) + 1]++;
            for (int l = 0; l <= maxlen; ++l) cnt[l + 1] += cnt[l];
            for (int k = 0; k < m; ++k) ord[cnt[maxlen - (ends[k + 1] - ends[k])]++] = k;
        };
        by_length(sep1, ord1); by_length(t2, ord2);
        dv.later(&d_ord1, ord1); dv.later(&d_ord2, ord2);
    }
    TRY(dv.flush(c, c->stream));                              // level-1 chain and segment orders: one upload
    const double t_prep = ms_since(T0);
    if (getenv("DSSS_PG_VERBOSE")) fprintf(stderr, "[dsss pg prep] edges %.2f ms, separators %.2f ms (bits %.2f mid %.2f redges %.2f), coordinates launched at %.2f ms; beside the analysis: level-1 chain %.2f ms, rest %.2f ms\n", t_p1, t_p2 - t_p1, t_q1 - t_p1, t_q3 - t_q1, t_p2 - t_q3, t_prep0, t_q2, t_prep - t_prep0 - t_q2);
    const bool verbose = getenv("DSSS_PG_VERBOSE") != nullptr;

    // incidence lists of the poses (edge order): only the device kernels read them, so they are built while the analysis runs
    std::vector<int> adj_ptr(n + 1, 0), adj_edge(2 * (size_t)ne);
    for (int e = 0; e < ne; ++e) { adj_ptr[ea[e] + 1]++; adj_ptr[eb[e] + 1]++; }
    for (int i = 0; i < n; ++i) adj_ptr[i + 1] += adj_ptr[i];
    { std::vector<int> fill(adj_ptr.begin(), adj_ptr.end() - 1);
      for (int e = 0; e < ne; ++e) { adj_edge[fill[ea[e]]++] = e << 1; adj_edge[fill[eb[e]]++] = (e << 1) | 1; } }
    // loop closures that repeat an unordered pose pair (see pg_scatter_lc_kernel).  The pipeline's edges -- a < b, b strictly
    // ascending -- cannot: they skip the sort
    std::vector<int> lc_link;
    {
        bool plain = true;
        for (int e = 0; e < ne && plain; ++e) plain = ea[e] < eb[e] && (e == 0 || eb[e - 1] < eb[e]);
        if (!plain) {
            std::vector<std::pair<unsigned long long, int>> key(ne);
            for (int e = 0; e < ne; ++e) key[e] = { ((unsigned long long)(unsigned)std::min(ea[e], eb[e]) << 32) | (unsigned)std::max(ea[e], eb[e]), e };
            std::sort(key.begin(), key.end());                           // (pair, edge index): the members of a group in edge order
            bool dups = false;
            for (int k = 0; k + 1 < ne && !dups; ++k) dups = key[k].first == key[k + 1].first;
            if (dups) {
                lc_link.assign((size_t)2 * ne, -1);
                for (int k = 0; k < ne; ++k) {
                    lc_link[2 * (size_t)key[k].second] = (k == 0 || key[k - 1].first != key[k].first) ? 1 : 0;
                    if (k + 1 < ne && key[k + 1].first == key[k].first) lc_link[2 * (size_t)key[k].second + 1] = key[k + 1].second;
                }
            }
        }
    }
    // ---- early device set-up (nothing here reads S)
    pose_t *d_X, *d_Xn, *d_meas, *d_emeas; int *d_ea, *d_eb, *d_eo, *d_adj_ptr, *d_adj_edge, *d_perm;
    double *d_ew, *d_r, *d_Ji, *d_D, *d_C, *d_g, *d_delta, *d_E, *d_Dl, *d_gi, *d_sDL, *d_sDR, *d_sGL, *d_sGR, *d_sS, *d_L, *d_x, *d_part, *d_scal;
    double *d_F, *d_R, *d_ubin, *d_aval;
    int* d_binperm = nullptr;
    int *d_colptr, *d_rowidx, *d_rlptr, *d_rlcol, *d_rlpos, *d_rlrow, *d_binptr, *d_bincols, *d_dest, *d_fail, *d_map; long long* d_mapptr;
    int *d_binroot_ptr, *d_binroot_idx, *d_broot_b, *d_broot_of_col, *d_anc_first, *d_anc_rel, *d_rel, *d_fa_src, *d_fa_col, *d_fa_tr, *d_frows, *d_xr_ptr, *d_xr_child, *d_xr_row, *d_fa_rowptr;
    long long* d_broot_uoff; pg_front* d_FD; pg_child* d_CH;
    double* d_red;
    const int nf = n + ne, nblk = (nf + 255) / 256;
    TRY(dv.alloc(c, &d_X, n)); TRY(dv.alloc(c, &d_Xn, n)); TRY(dv.alloc(c, &d_meas, n)); TRY(dv.upload(c, &d_emeas, emeas));
    TRY(dv.upload(c, &d_ea, ea)); TRY(dv.upload(c, &d_eb, eb)); TRY(dv.upload(c, &d_eo, eo)); TRY(dv.upload(c, &d_ew, ew));
    TRY(dv.upload(c, &d_adj_ptr, adj_ptr)); TRY(dv.upload(c, &d_adj_edge, adj_edge));
    int* d_lc_link = nullptr; if (!lc_link.empty()) TRY(dv.upload(c, &d_lc_link, lc_link));
    TRY(dv.alloc(c, &d_r, (size_t)nf * 6)); TRY(dv.alloc(c, &d_Ji, (size_t)nf * 36));
    // a second set of residuals and Jacobians: the linearisation that measures a trial's error at X (+) delta IS the next iteration's
    // linearisation when the trial is accepted (same kernel, same point, same bits) -- it writes them here and the sets swap
    double *d_r2, *d_Ji2; TRY(dv.alloc(c, &d_r2, (size_t)nf * 6)); TRY(dv.alloc(c, &d_Ji2, (size_t)nf * 36));
    TRY(dv.alloc(c, &d_D, (size_t)n * 36)); TRY(dv.alloc(c, &d_C, (size_t)n * 36)); TRY(dv.alloc(c, &d_g, (size_t)n * 6)); TRY(dv.alloc(c, &d_delta, (size_t)n * 6));
    TRY(dv.alloc(c, &d_E, (size_t)n * 36)); TRY(dv.alloc(c, &d_Dl, (size_t)n * 36)); TRY(dv.alloc(c, &d_gi, (size_t)n * 6));
    TRY(dv.alloc(c, &d_sDL, (size_t)nseg1 * 36)); TRY(dv.alloc(c, &d_sDR, (size_t)nseg1 * 36)); TRY(dv.alloc(c, &d_sGL, (size_t)nseg1 * 6));
    TRY(dv.alloc(c, &d_sGR, (size_t)nseg1 * 6)); TRY(dv.alloc(c, &d_sS, (size_t)nseg1 * 36));
    // level-1 chain (true separators + chunk ends) and its condensation onto the true separators (pass 2)
    double *d_D1, *d_C1, *d_g1, *d_E1, *d_Dl1, *d_gi1, *d_delta1, *d_s2DL, *d_s2DR, *d_s2GL, *d_s2GR, *d_s2S;
    TRY(dv.alloc(c, &d_D1, (size_t)ns1 * 36)); TRY(dv.alloc(c, &d_C1, (size_t)ns1 * 36)); TRY(dv.alloc(c, &d_g1, (size_t)ns1 * 6)); TRY(dv.alloc(c, &d_delta1, (size_t)ns1 * 6));
    TRY(dv.alloc(c, &d_E1, (size_t)ns1 * 36)); TRY(dv.alloc(c, &d_Dl1, (size_t)ns1 * 36)); TRY(dv.alloc(c, &d_gi1, (size_t)ns1 * 6));
    TRY(dv.alloc(c, &d_s2DL, (size_t)std::max(nseg, 1) * 36)); TRY(dv.alloc(c, &d_s2DR, (size_t)std::max(nseg, 1) * 36)); TRY(dv.alloc(c, &d_s2GL, (size_t)std::max(nseg, 1) * 6));
    TRY(dv.alloc(c, &d_s2GR, (size_t)std::max(nseg, 1) * 6)); TRY(dv.alloc(c, &d_s2S, (size_t)std::max(nseg, 1) * 36));
    double* d_rdiag; TRY(dv.alloc(c, &d_rdiag, (size_t)ns * 6));      // reciprocal diagonals of the binned columns' pivots (forward -> backward substitution)
    TRY(dv.alloc(c, &d_x, (size_t)ns * 6)); TRY(dv.alloc(c, &d_part, (size_t)nblk)); TRY(dv.alloc(c, &d_scal, 8)); TRY(dv.alloc(c, &d_fail, 1)); TRY(dv.alloc(c, &d_red, 8));
    hipStream_t st = c->stream;
#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { c->err = std::string(#x) + ": " + hipGetErrorString(_e); abandon(); return DSSS_E_HIP; } } while (0)
    // sums over the factors are partial on every rank: one small all-reduce makes them global (and identical everywhere)
    auto reduce_scalars = [&](double* host3, int* failed) -> int {
        if (world > 1) {
            hipLaunchKernelGGL(pg_comm_scal_kernel, dim3(1), dim3(64), 0, st, d_scal, d_fail, d_red);
            int rc2 = dsss_comm_allreduce(c, d_red, 4, st); if (rc2) { abandon(); return rc2; }
            double h4[4];
            HCK(hipMemcpyAsync(h4, d_red, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
            HCK(hipStreamSynchronize(st));
            host3[0] = h4[0]; host3[1] = h4[1]; host3[2] = h4[2]; *failed = h4[3] != 0.0;
        } else {
            // into page-locked memory: a copy to the caller's stack is staged by the runtime and waits for it twice per trial
            if (!c->pg_scal_host) HCK(hipHostMalloc((void**)&c->pg_scal_host, 8 * sizeof(double), hipHostMallocDefault));
            HCK(hipMemcpyAsync(c->pg_scal_host, d_scal, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
            HCK(hipMemcpyAsync(c->pg_scal_host + 4, d_fail, sizeof(int), hipMemcpyDeviceToHost, st));
            HCK(hipStreamSynchronize(st));
            host3[0] = c->pg_scal_host[0]; host3[1] = c->pg_scal_host[1]; host3[2] = c->pg_scal_host[2];
            *failed = *reinterpret_cast<const int*>(c->pg_scal_host + 4);
        }
        return DSSS_OK;
    };
    auto error_of = [&](const pose_t* Xd, double* out) -> int {
        hipLaunchKernelGGL(pg_linearize_kernel, dim3(nblk), dim3(256), 0, st, n, ne, Xd, d_meas, W, d_ea, d_eb, d_eo, d_emeas, d_ew, (double*)nullptr, (double*)nullptr, d_part, mp0, mp1);
        hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal);
        double h3[3]; int f0 = 0;
        HCK(hipMemsetAsync(d_fail, 0, sizeof(int), st));
        int rc2 = reduce_scalars(h3, &f0); if (rc2) return rc2;
        *out = h3[0];
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
                if (attempt > 3) { abandon(); DSSS_FAIL(c, DSSS_E_NUMERIC, "normal generator: not enough accepted attempts"); }
                natt *= 2;
            }
        }
        hipLaunchKernelGGL(pg_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_dr6, d_norm, c->pg.add_noise, d_X, d_meas);
        // online use (dsss_posegraph_update): the pings the previous update covered start from its estimate, the new ones where
        // the reference puts them (dead reckoning o noise, optimizer.cpp:150-160)
        if (c->pg_online && c->pg_warm_n > 0)
            HCK(hipMemcpyAsync(d_X, c->pg_warm, (size_t)std::min(n, c->pg_warm_n) * sizeof(pose_t), hipMemcpyDeviceToDevice, st));
    }
    double lambda = c->pg.lambda0, err = 0, err0 = 0, cur = 0;
    int iters = 0, nfact = 0;
    TRY(error_of(d_X, &err));
    err0 = err;
    // the chain part of a trial: per-pose blocks, pass 1 (chunks of poses onto their ends), the level-1 chain, pass 2 (runs of
    // chunk ends onto the true separators)
    const bool seg8 = !(getenv("DSSS_PG_SEG") && atoi(getenv("DSSS_PG_SEG")) == 16);      // A/B: sixteen lanes per segment (rounds 2 - 4)
    auto chain_part = [&]() {
        hipMemsetAsync(d_fail, 0, sizeof(int), st);
        hipLaunchKernelGGL(pg_assemble_kernel, dim3((n + PG_ASM_POSES - 1) / PG_ASM_POSES), dim3(6 * PG_ASM_POSES), 0, st, n, W, d_r, d_Ji, d_adj_ptr, d_adj_edge, d_ew, d_scal + 3, d_D, d_C, d_g, d_eo, mp0, mp1);
        if (seg8) hipLaunchKernelGGL(pg_segment8_kernel, dim3((nseg1 + 31) / 32), dim3(256), 0, st, nseg1, d_ord1, d_sep1, d_D, d_C, d_g, d_E, d_Dl, d_gi, d_sDL, d_sDR, d_sGL, d_sGR, d_sS, d_fail, mp0, mp1);
        else hipLaunchKernelGGL(pg_segment_kernel, dim3((nseg1 + 15) / 16), dim3(256), 0, st, nseg1, d_ord1, d_sep1, d_D, d_C, d_g, d_E, d_Dl, d_gi, d_sDL, d_sDR, d_sGL, d_sGR, d_sS, d_fail, mp0, mp1);
        hipLaunchKernelGGL(pg_chain1_kernel, dim3((unsigned)(((long long)ns1 * 42 + 255) / 256)), dim3(256), 0, st, ns1, d_sep1, d_D, d_g, d_sDL, d_sDR, d_sGL, d_sGR, d_sS, d_D1, d_C1, d_g1, mp0, mp1);
        if (nseg > 0 && seg8) hipLaunchKernelGGL(pg_segment8_kernel, dim3((nseg + 31) / 32), dim3(256), 0, st, nseg, d_ord2, d_t2, d_D1, d_C1, d_g1, d_E1, d_Dl1, d_gi1, d_s2DL, d_s2DR, d_s2GL, d_s2GR, d_s2S, d_fail, kp0, kp1);
        else if (nseg > 0) hipLaunchKernelGGL(pg_segment_kernel, dim3((nseg + 15) / 16), dim3(256), 0, st, nseg, d_ord2, d_t2, d_D1, d_C1, d_g1, d_E1, d_Dl1, d_gi1, d_s2DL, d_s2DR, d_s2GL, d_s2GR, d_s2S, d_fail, kp0, kp1);
    };
    const bool will_iterate = err > 0 && c->pg.max_iters > 0;
    bool pre_lin = false, pre_chain = false;
    if (will_iterate) {                              // first linearisation and the chain part of the first trial, before the analysis is in
        hipLaunchKernelGGL(pg_linearize_kernel, dim3(nblk), dim3(256), 0, st, n, ne, d_X, d_meas, W, d_ea, d_eb, d_eo, d_emeas, d_ew, d_r, d_Ji, d_part, mp0, mp1);
        hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal);
        HCK(hipMemcpyAsync(d_scal + 3, &lambda, sizeof(double), hipMemcpyHostToDevice, st));
        chain_part();
        pre_lin = pre_chain = true;
    }
    // ---- The bottom of the tree (ordering, column structures, bins, update lists, destinations) is final about a millisecond before the
    // fronts and the schedule are: with one partition its tables go up as soon as the analysing thread says so, and the scatter and
    // the bins of the FIRST trial run while the host finishes the analysis.
    size_t nnzL = 0; int nval = 0;
    size_t ncv = 0, nif = 0, comm_total = 8;
    double *d_comm = nullptr, *d_avalif = nullptr, *d_xif = nullptr, *d_commU = nullptr;
    int *d_ifslot = nullptr, *d_ifsep = nullptr;
    std::vector<int> ifslot;
    int bin_lo = 0, bin_hi = 0, nbins = 0;
    bool pre_bins = false;
    // The bins' update lists, update-map offsets and root-boundary indices are built on the device from the column structures
    // (pg_rl_count_kernel ... pg_build_map_kernel): everything they read -- colptr, rowidx, the binned flags, the subtree roots -- is final
    // BEFORE the host packs the bins (pg_sym_opts::on_lists_ready), so they are uploaded and the ten kernels run while it does.
    bool lists_done = false;
    auto upload_lists = [&]() -> int {
        nnzL = S.rowidx.size();
        int rc2 = DSSS_OK;
        char* d_binned = nullptr; int* d_rootof = nullptr;
        dv.later(&d_colptr, S.colptr); dv.later(&d_rowidx, S.rowidx); dv.later(&d_binned, S.binned); dv.later(&d_rootof, S.root_of);
        if ((rc2 = dv.flush(c, st))) return rc2;
        int *d_cnt, *d_cur, *d_bs32, *d_tot32; long long *d_bs64, *d_tot64;
        const int nsb = (ns + 1023) / 1024;
        if ((rc2 = dv.alloc(c, &d_cnt, (size_t)ns)) || (rc2 = dv.alloc(c, &d_cur, (size_t)ns)) || (rc2 = dv.alloc(c, &d_bs32, (size_t)nsb)) || (rc2 = dv.alloc(c, &d_tot32, 1)) ||
            (rc2 = dv.alloc(c, &d_bs64, (size_t)nsb)) || (rc2 = dv.alloc(c, &d_tot64, 1)) || (rc2 = dv.alloc(c, &d_rlptr, (size_t)ns + 1)) || (rc2 = dv.alloc(c, &d_mapptr, (size_t)ns + 1)) ||
            (rc2 = dv.alloc(c, &d_rlcol, nnzL)) || (rc2 = dv.alloc(c, &d_rlpos, nnzL)) || (rc2 = dv.alloc(c, &d_rlrow, nnzL)) ||      // (an entry of L is on at most one list)
            (rc2 = dv.alloc(c, &d_anc_first, (size_t)ns)) || (rc2 = dv.alloc(c, &d_anc_rel, nnzL))) return rc2;
        if (nsb > 1024 * 1024) DSSS_FAIL(c, DSSS_E_CAPACITY, "%d separators", ns);
        hipMemsetAsync(d_cnt, 0, sizeof(int) * (size_t)ns, st); hipMemsetAsync(d_cur, 0, sizeof(int) * (size_t)ns, st);
        hipMemsetAsync(d_anc_first, 0, sizeof(int) * (size_t)ns, st); hipMemsetAsync(d_anc_rel, 0xff, sizeof(int) * nnzL, st);
        const dim3 gcol((ns + 255) / 256);
        hipLaunchKernelGGL(pg_rl_count_kernel, gcol, dim3(256), 0, st, ns, d_colptr, d_rowidx, d_binned, d_cnt);
        hipLaunchKernelGGL((pg_scan_block_kernel<int, 0>), dim3(nsb), dim3(1024), 0, st, ns, d_cnt, d_colptr, d_rlptr, d_bs32);
        hipLaunchKernelGGL((pg_scan_tops_kernel<int>), dim3(1), dim3(1024), 0, st, nsb, d_bs32, d_tot32);
        hipLaunchKernelGGL((pg_scan_add_kernel<int>), dim3(nsb), dim3(1024), 0, st, ns, d_rlptr, d_bs32, d_tot32);
        hipLaunchKernelGGL((pg_scan_block_kernel<long long, 1>), dim3(nsb), dim3(1024), 0, st, ns, d_cnt, d_colptr, d_mapptr, d_bs64);
        hipLaunchKernelGGL((pg_scan_tops_kernel<long long>), dim3(1), dim3(1024), 0, st, nsb, d_bs64, d_tot64);
        hipLaunchKernelGGL((pg_scan_add_kernel<long long>), dim3(nsb), dim3(1024), 0, st, ns, d_mapptr, d_bs64, d_tot64);
        hipLaunchKernelGGL(pg_rl_fill_kernel, gcol, dim3(256), 0, st, ns, d_colptr, d_rowidx, d_binned, d_rlptr, d_cur, d_rlcol, d_rlpos);
        hipLaunchKernelGGL(pg_rl_sort_kernel, dim3((ns + 3) / 4), dim3(256), 0, st, ns, d_rlptr, d_rlcol, d_rlpos, d_rlrow, d_fail);
        hipLaunchKernelGGL(pg_anc_rel_kernel, gcol, dim3(256), 0, st, ns, d_colptr, d_rowidx, d_binned, d_rootof, d_anc_first, d_anc_rel);
        // the totals stay on the device: a binned column has at most 42 blocks and an entry of L is on at most one list, so the map
        // has at most 42 nnz(L) entries -- allocated to that bound (54 MB at C3), filled and built up to the device-side totals
        const long long mapsz = 42LL * (long long)nnzL; const int nupd = (int)nnzL;
        if (mapsz > (1LL << 31)) DSSS_FAIL(c, DSSS_E_CAPACITY, "update map bound of %lld entries", mapsz);
        if ((rc2 = dv.alloc(c, &d_map, (size_t)mapsz))) return rc2;
        hipLaunchKernelGGL(pg_fill_map_kernel, dim3(2048), dim3(256), 0, st, d_map, d_tot64);
        if (nupd > 0) hipLaunchKernelGGL(pg_build_map_kernel, dim3((nupd + 255) / 256), dim3(256), 0, st, nupd, d_rlrow, d_rlptr, d_rlcol, d_rlpos, d_colptr, d_rowidx, d_mapptr, d_map, d_tot32);
        lists_done = true;
        return DSSS_OK;
    };
    auto upload_bottom = [&]() -> int {
        nnzL = S.rowidx.size(); nval = (int)S.dest_bin.size();
        ncv = S.comm_vals.size(); nif = S.iface_seps.size();
        comm_total = ncv * 36 + nif * 6 + (size_t)S.comm_doubles + 8;
        int rc2 = DSSS_OK;
        if (lists_on_device && !lists_done && (rc2 = upload_lists())) return rc2;
        dv.later(&d_perm, S.perm);
        if ((rc2 = dv.alloc(c, &d_L, nnzL * 36))) return rc2;
        if ((rc2 = dv.alloc(c, &d_ubin, (size_t)S.ubin_doubles))) return rc2;
        // value array of the fronts; its tail IS the buffer the all-reduce sums: [interface values | interface right-hand sides |
        // update matrices that cross into the interface | 8 scalars]
        if ((rc2 = dv.alloc(c, &d_aval, (size_t)nval * 36 + comm_total))) return rc2;
        d_comm = d_aval + (size_t)nval * 36; d_avalif = d_comm; d_xif = d_comm + ncv * 36; d_commU = d_xif + nif * 6;
        ifslot.assign(ns, -1); for (size_t q = 0; q < nif; ++q) ifslot[S.iface_seps[q]] = (int)q;
        dv.later(&d_ifslot, ifslot); dv.later(&d_ifsep, S.iface_seps);
        if (!lists_on_device) { dv.later(&d_colptr, S.colptr); dv.later(&d_rowidx, S.rowidx); }
        dv.later(&d_binptr, S.binptr); dv.later(&d_bincols, S.bincols); dv.later(&d_binperm, S.bin_perm);
        dv.later(&d_dest, S.dest_bin);
        dv.later(&d_binroot_ptr, S.binroot_ptr); dv.later(&d_binroot_idx, S.binroot_idx); dv.later(&d_broot_b, S.broot_b); dv.later(&d_broot_uoff, S.broot_uoff);
        dv.later(&d_broot_of_col, S.broot_of_col);
        if (!lists_on_device) { dv.later(&d_rlptr, S.rlptr); dv.later(&d_rlcol, S.rlcol); dv.later(&d_rlpos, S.rlpos); dv.later(&d_rlrow, S.rlrow); dv.later(&d_mapptr, S.mapptr);
                                dv.later(&d_anc_first, S.anc_first); dv.later(&d_anc_rel, S.anc_rel); }
        if ((rc2 = dv.flush(c, st))) return rc2;
        // this rank's bins are one contiguous range (bins never straddle partitions, partitions are ascending in the order)
        { const int nb_all = (int)S.binptr.size() - 1; bin_lo = 0; while (bin_lo < nb_all && S.bin_part[bin_lo] < part_lo) ++bin_lo; bin_hi = bin_lo; while (bin_hi < nb_all && S.bin_part[bin_hi] < part_hi) ++bin_hi; }
        nbins = bin_hi - bin_lo;
        if (lists_on_device) return DSSS_OK;
        const long long mapsz = S.mapptr[ns]; const int nupd = (int)S.rlcol.size();
        if (mapsz > (1LL << 31)) DSSS_FAIL(c, DSSS_E_CAPACITY, "update map of %lld entries", mapsz);
        if ((rc2 = dv.alloc(c, &d_map, (size_t)mapsz))) return rc2;
        if (hipMemsetAsync(d_map, 0xff, (size_t)std::max<long long>(mapsz, 1) * sizeof(int), st) != hipSuccess) DSSS_FAIL(c, DSSS_E_HIP, "hipMemsetAsync(update map)");
        if (nupd > 0) hipLaunchKernelGGL(pg_build_map_kernel, dim3((nupd + 255) / 256), dim3(256), 0, st, nupd, d_rlrow, d_rlptr, d_rlcol, d_rlpos, d_colptr, d_rowidx, d_mapptr, d_map, (const int*)nullptr);
        return DSSS_OK;
    };
    // the bottom part of a trial: reduced system into the factor / the value array, then the bins
    auto bottom_trial = [&](double bins_flops) {
        hipMemsetAsync(d_L, 0, nnzL * 36 * sizeof(double), st);
        if (nparts > 1) hipMemsetAsync(d_comm, 0, comm_total * sizeof(double), st);
        hipLaunchKernelGGL(pg_scatter_base_kernel, dim3((unsigned)(((long long)ns * 78 + 255) / 256)), dim3(256), 0, st, ns, d_t2, d_perm, d_D1, d_g1, d_s2DL, d_s2DR, d_s2GL, d_s2GR, d_s2S, d_dest, d_L, d_aval, d_x,
                           d_ifslot, d_avalif, d_xif, kp0, kp1);
        if (ne > 0) hipLaunchKernelGGL(pg_scatter_lc_kernel, dim3((unsigned)(((long long)ne * 36 + 255) / 256)), dim3(256), 0, st, n, ne, ns, d_Ji, d_ew, d_dest, d_L, d_aval, d_avalif, d_eo, mp0, mp1, d_lc_link);
        if (nbins > 0) { dsss_scope s1(c, DSSS_K_PG_SUBTREE, bins_flops);      // flops of the binned columns
                         hipLaunchKernelGGL(pg_factor_subtree_kernel, dim3(nbins), dim3(256), 0, st, d_binperm + bin_lo, d_binptr, d_bincols, d_colptr, d_rlptr, d_rlcol, d_rlpos, d_mapptr, d_map, d_L, d_x, d_fail,
                                            d_binroot_ptr, d_binroot_idx, d_broot_b, d_broot_uoff, d_broot_of_col, d_anc_first, d_anc_rel, d_ubin, d_rdiag); }
    };
    const bool early_ok = !(getenv("DSSS_PG_EARLY") && atoi(getenv("DSSS_PG_EARLY")) == 0);      // A/B switch
    const bool early_bottom = early_ok && nparts == 1 && will_iterate && pre_chain;
    if (early_bottom) {
        if (lists_on_device) { lists_fut.wait(); TRY(upload_lists()); }
        bottom_fut.wait();
        TRY(upload_bottom());
        bottom_trial(0.0);                                   // (its flop count is known when the analysis has finished: added below)
        pre_bins = true;
    }
    sym_thread.join();
    if (S.ownership_violations) { abandon(); DSSS_FAIL(c, DSSS_E_STATE, "pose-graph analysis: %d separators with a higher-rank neighbour are not interface", S.ownership_violations); }
    const double t_sym = ms_since(T1);
    const auto T2 = std::chrono::steady_clock::now();
    const int nfr = (int)S.f_c0.size(), npan = S.npanels;
    if (!early_bottom) TRY(upload_bottom());
    else if (c->prof.on) c->prof.work[DSSS_K_PG_SUBTREE] += std::max(0.0, S.flops_factor - S.flops_fronts);
    if (verbose)
        fprintf(stderr, "[dsss pg] rank %d/%d parts %d (own %d..%d, poses %d..%d)  poses %d  LC edges %d  separators %d (interface %zu)  nnz(L) blocks %zu  bins %d (%d cols)  fronts %d (largest %d block rows, arena %.0f MB)  panels %d in %d levels  all-reduce %.1f MB\n",
                rank, world, nparts, part_lo, part_hi, mp0, mp1, n, ne, ns, S.iface_seps.size(), nnzL, (int)S.binptr.size() - 1, (int)S.bincols.size(), nfr, S.max_front_n, S.front_doubles * 8e-6, npan, S.nlev,
                (S.comm_doubles + 36.0 * S.comm_vals.size() + 6.0 * S.iface_seps.size()) * 8e-6);

    TRY(dv.alloc(c, &d_F, (size_t)S.front_doubles)); TRY(dv.alloc(c, &d_R, (size_t)S.frhs_doubles));
    const int rsu_max = getenv("DSSS_PG_RSU") ? atoi(getenv("DSSS_PG_RSU")) : PG_RSU_MAX_TILES;      // A/B: 0 = separate row solve and update launches on every level; n = tile limit
    const bool use_rsu = rsu_max > 0;
    const int rsu32_max = getenv("DSSS_PG_RSU32") ? atoi(getenv("DSSS_PG_RSU32")) : PG_RSU32_MAX_TILES;      // levels with at most this many 64 x 64 tiles run them as 32 x 32 quarters (0 = never)
    double* d_FL = nullptr;                              // second front arena: L21 of the levels that run the fused kernel
    if (use_rsu) TRY(dv.alloc(c, &d_FL, (size_t)S.front_doubles));
    int *d_pk_child, *d_pk_row; pg_pack* d_PK;
    dv.later(&d_rel, S.rel); dv.later(&d_fa_src, S.fa_src); dv.later(&d_fa_col, S.fa_col); dv.later(&d_fa_tr, S.fa_tr);
    dv.later(&d_frows, S.f_rows); dv.later(&d_xr_ptr, S.xr_ptr); dv.later(&d_xr_child, S.xr_child);
    dv.later(&d_xr_row, S.xr_row); dv.later(&d_fa_rowptr, S.fa_rowptr);
    struct dsched { int *lv_front, *lv_step, *asm_front, *asm_row, *tile_item, *tile_ij; } DO = {}, DI = {};
    for (int w2 = 0; w2 < 2; ++w2) {
        const pg_sched& H = w2 ? SI : SO; dsched& Dv = w2 ? DI : DO;
        dv.later(&Dv.lv_front, H.lv_front); dv.later(&Dv.lv_step, H.lv_step); dv.later(&Dv.asm_front, H.asmrow_front); dv.later(&Dv.asm_row, H.asmrow_row);
        dv.later(&Dv.tile_item, H.tile_item); dv.later(&Dv.tile_ij, H.tile_ij);
    }
    int n_pack = 0;
    {   // front and child descriptors (the children point straight at the update matrices: F22 of a front, U of a bin root)
        std::vector<pg_front> FD(nfr); std::vector<pg_child> CH(S.ch_kind.size());
        for (int f = 0; f < nfr; ++f) {
            pg_front& d = FD[f];
            d.off = S.f_off[f]; d.roff = S.f_roff[f]; d.ld = S.f_ld[f]; d.n6 = 6 * S.f_n[f]; d.s6 = 6 * S.f_s[f]; d.c0 = S.f_c0[f];
            d.rowptr = S.f_rowptr[f]; d.pan0 = S.f_pan0[f]; d.ch0 = S.ch_ptr[f]; d.ch1 = S.ch_ptr[f + 1]; d.fa0 = S.fa_ptr[f]; d.fa1 = S.fa_ptr[f + 1];
        }
        for (size_t q = 0; q < CH.size(); ++q) {
            pg_child& d = CH[q]; d.relptr = S.ch_relptr[q];
            if (S.ch_kind[q]) { const int ri = S.ch_id[q]; d.cb = S.broot_b[ri]; d.cld = 6 * d.cb; d.U = d_ubin + S.broot_uoff[ri]; d.g = d.U + (size_t)d.cld * d.cld; }
            else { const int gf = S.ch_id[q]; d.cb = S.f_n[gf] - S.f_s[gf]; d.cld = S.f_ld[gf]; d.U = d_F + S.f_off[gf] + (size_t)(6 * S.f_s[gf]) * d.cld + 6 * S.f_s[gf]; d.g = d_R + S.f_roff[gf] + 6 * S.f_s[gf]; }
        }
        // children that cross from an interior into the interface are read from the summed buffer; the owner packs them there
        std::vector<int> comm_of_front(nfr, -1), comm_of_broot(S.broot.size(), -1);
        for (size_t q = 0; q < S.comm_kind.size(); ++q) (S.comm_kind[q] ? comm_of_broot[S.comm_id[q]] : comm_of_front[S.comm_id[q]]) = (int)q;
        std::vector<pg_pack> PK(S.comm_kind.size()); std::vector<int> pk_child, pk_row;
        for (int f = 0; f < nfr; ++f) {
            if (S.f_part[f] >= 0) continue;
            for (int q = S.ch_ptr[f]; q < S.ch_ptr[f + 1]; ++q) {
                const int cq = S.ch_kind[q] ? comm_of_broot[S.ch_id[q]] : comm_of_front[S.ch_id[q]];
                if (cq < 0) continue;
                pg_child& d = CH[q];
                pg_pack& k = PK[cq]; k.U = d.U; k.g = d.g; k.cld = d.cld; k.cb = d.cb; k.dst = d_commU + S.comm_off[cq];
                d.U = k.dst; d.cld = 6 * d.cb; d.g = d.U + (size_t)d.cld * d.cld;
                if (S.comm_part[cq] >= part_lo && S.comm_part[cq] < part_hi) for (int r2 = 0; r2 < d.cb; ++r2) { pk_child.push_back(cq); pk_row.push_back(r2); }
            }
        }
        dv.later(&d_FD, FD); dv.later(&d_CH, CH); dv.later(&d_PK, PK); dv.later(&d_pk_child, pk_child); dv.later(&d_pk_row, pk_row);
        TRY(dv.flush(c, st));                                   // FD, CH, PK and the lists above are still alive here
        n_pack = (int)pk_child.size();
    }
    const int max_n6 = std::max(SO.max_n6, SI.max_n6);
    unsigned long long* d_stamps = nullptr; if (getenv("DSSS_PG_STAMPS")) TRY(dv.alloc(c, &d_stamps, 16));
    const bool diag3_panel = getenv("DSSS_PG_PANEL") && !strcmp(getenv("DSSS_PG_PANEL"), "diag3");          // A/B: two barriers per block
    double* d_Tinv; TRY(dv.alloc(c, &d_Tinv, (size_t)std::max(npan, 1) * PG_NB4 * 16));
    double* d_bwp = nullptr;                                // partial sums of the split back-substitution products (tall fronts only)
    {
        size_t need = 0;
        for (const pg_sched* H : { &SO, &SI })
            for (int l = 0; l < H->nlev; ++l)
                if (64 * H->trsm_chunks[l] > PG_BWD_SPLIT) need = std::max(need, (size_t)(H->lv_ptr[l + 1] - H->lv_ptr[l]) * ((64 * H->trsm_chunks[l] + PG_BWD_RC - 1) / PG_BWD_RC) * 96);
        if (need > 0) TRY(dv.alloc(c, &d_bwp, need));
    }
    const int bwd_lds = (int)(((PG_PW * 6) * PG_BWD2_LD + PG_NB4 * 16 + 10 * (PG_PW * 6) + std::min(max_n6, PG_BWD2_SX) + 16) * sizeof(double));
    if (bwd_lds > 160 * 1024) { abandon(); DSSS_FAIL(c, DSSS_E_CAPACITY, "front of %d scalar rows: back-substitution needs %d B of LDS", max_n6, bwd_lds); }
    {   // the back-substitution keeps L11, the slot sums and x2 in dynamic LDS
        hipFuncSetAttribute((const void*)pg_front_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    const double t_up = ms_since(T2);
    const auto T3 = std::chrono::steady_clock::now();
    dsss_scope sc(c, DSSS_K_PG);
    if (will_iterate) do {                        // NonlinearOptimizer::defaultOptimize returns before iterating when maxIterations is reached
        cur = err;
        double oldLin = err;                                                   // linear error at delta = 0 == the error at X (same sum, already global)
        if (!pre_lin) {
            hipLaunchKernelGGL(pg_linearize_kernel, dim3(nblk), dim3(256), 0, st, n, ne, d_X, d_meas, W, d_ea, d_eb, d_eo, d_emeas, d_ew, d_r, d_Ji, d_part, mp0, mp1);
            hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal);
        }
        pre_lin = false;
        for (;;) {
            // ---- solve (H + lambda I) delta = -g ; lambda lives in device memory
            if (!pre_chain) HCK(hipMemcpyAsync(d_scal + 3, &lambda, sizeof(double), hipMemcpyHostToDevice, st));
            {
                if (!pre_chain) chain_part();
                pre_chain = false;
                if (!pre_bins) bottom_trial(std::max(0.0, S.flops_factor - S.flops_fronts));
                pre_bins = false;
                // fronts, level by level: assemble the fronts that start here, then one panel step of every active front
                auto run_levels = [&](const pg_sched& H, const dsched& Dv) {
                    for (int l = 0; l < H.nlev; ++l) {
                        const int nas = H.asmrow_ptr[l + 1] - H.asmrow_ptr[l], nit = H.lv_ptr[l + 1] - H.lv_ptr[l], ntl = H.tile_ptr[l + 1] - H.tile_ptr[l];
                        const int* itf = Dv.lv_front + H.lv_ptr[l]; const int* its = Dv.lv_step + H.lv_ptr[l];
                        if (nas > 0) { dsss_scope s2(c, DSSS_K_PG_ASM);
                            hipLaunchKernelGGL(pg_front_asm_kernel, dim3(nas), dim3(256), 0, st, Dv.asm_front + H.asmrow_ptr[l], Dv.asm_row + H.asmrow_ptr[l], d_FD, d_CH, d_rel, d_xr_ptr, d_xr_child, d_xr_row,
                                               d_fa_rowptr, d_fa_src, d_fa_col, d_fa_tr, d_aval, d_x, d_F, d_R); }
                        if (nit == 0) continue;
                        { dsss_scope s3(c, DSSS_K_PG_DIAG, H.fl_diag[l]);
                          if (!diag3_panel) hipLaunchKernelGGL(pg_front_diag4_kernel, dim3(nit), dim3(256), 0, st, itf, its, d_FD, d_F, d_R, d_fail, d_Tinv);
                          else {
                              hipLaunchKernelGGL(pg_front_diag3_kernel, dim3(nit), dim3(256), 0, st, itf, its, d_FD, d_F, d_R, d_fail, d_Tinv, d_stamps);
                              if (d_stamps && l == H.nlev - 1) { unsigned long long hs[16]; hipMemcpyAsync(hs, d_stamps, sizeof hs, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
                                  fprintf(stderr, "[dsss pg stamps] diag3 last level: %llu cycles\n", hs[2] - hs[0]); }
                          } }
                        if (use_rsu && ntl > 0 && ntl <= rsu_max) {
                            dsss_scope s45(c, DSSS_K_PG_RSU, H.fl_trsm[l] + H.fl_syrk[l]);
                            if (ntl <= rsu32_max) hipLaunchKernelGGL(pg_front_rsu_kernel<32>, dim3(4 * ntl), dim3(256), 0, st, itf, its, d_FD, Dv.tile_item + H.tile_ptr[l], Dv.tile_ij + H.tile_ptr[l], d_F, d_FL, d_R, d_Tinv);
                            else hipLaunchKernelGGL(pg_front_rsu_kernel<64>, dim3(ntl), dim3(512), 0, st, itf, its, d_FD, Dv.tile_item + H.tile_ptr[l], Dv.tile_ij + H.tile_ptr[l], d_F, d_FL, d_R, d_Tinv);
                        } else if (H.trsm_chunks[l] > 0) {
                            { dsss_scope s4(c, DSSS_K_PG_TRSM, H.fl_trsm[l]);
                              hipLaunchKernelGGL(pg_front_trsm2_kernel, dim3(nit, H.trsm_chunks[l]), dim3(256), 0, st, itf, its, d_FD, d_F, d_R, d_Tinv); }
                            { dsss_scope s5(c, DSSS_K_PG_ACC, H.fl_syrk[l]);
                              if (ntl > 0) hipLaunchKernelGGL(pg_front_syrk_kernel, dim3(ntl), dim3(256), 0, st, itf, its, d_FD, Dv.tile_item + H.tile_ptr[l], Dv.tile_ij + H.tile_ptr[l], d_F); }
                        }
                    }
                };
                auto run_levels_bwd = [&](const pg_sched& H, const dsched& Dv) {
                    for (int l = H.nlev - 1; l >= 0; --l) {
                        const int nit = H.lv_ptr[l + 1] - H.lv_ptr[l];
                        if (nit == 0) continue;
                        dsss_scope s6(c, DSSS_K_PG_BWD, H.fl_bwd[l], 2);
                        const int ntl = H.tile_ptr[l + 1] - H.tile_ptr[l];
                        const double* Fl = (use_rsu && ntl > 0 && ntl <= rsu_max) ? d_FL : d_F;
                        const int nch = 64 * H.trsm_chunks[l] > PG_BWD_SPLIT ? (64 * H.trsm_chunks[l] + PG_BWD_RC - 1) / PG_BWD_RC : 0;      // tall fronts on this level: their L21^T x2 by many workgroups
                        if (nch > 0) hipLaunchKernelGGL(pg_front_bwd_part_kernel, dim3(nch, nit), dim3(1024), 0, st, Dv.lv_front + H.lv_ptr[l], Dv.lv_step + H.lv_ptr[l], d_FD, d_frows, Fl, d_x, d_bwp, nch);
                        hipLaunchKernelGGL(pg_front_bwd2_kernel, dim3(nit), dim3(1024), bwd_lds, st, Dv.lv_front + H.lv_ptr[l], Dv.lv_step + H.lv_ptr[l], d_FD, d_frows, d_F, Fl, d_R, d_x, d_Tinv,
                                           nch > 0 ? (const double*)d_bwp : (const double*)nullptr, nch);
                    }
                };
                run_levels(SO, DO);
                if (nparts > 1) {
                    // the reduced Hessian on the interface: this rank's update matrices next to its share of the interface values and
                    // right-hand sides, summed over the ranks by ONE all-reduce; then the small replicated interface factorisation
                    if (n_pack > 0) hipLaunchKernelGGL(pg_comm_pack_kernel, dim3(n_pack), dim3(256), 0, st, d_pk_child, d_pk_row, d_PK);
                    { dsss_scope s8(c, DSSS_K_PG_COMM, (double)comm_total * 8);
                      int rc2 = dsss_comm_allreduce(c, d_comm, comm_total - 8, st); if (rc2) { abandon(); return rc2; } }
                    if (nif > 0) hipLaunchKernelGGL(pg_comm_xif_kernel, dim3(((int)nif + 255) / 256), dim3(256), 0, st, (int)nif, d_ifsep, d_perm, d_xif, d_x);
                    run_levels(SI, DI);
                    run_levels_bwd(SI, DI);
                }
                run_levels_bwd(SO, DO);
                if (nbins > 0) { dsss_scope s7(c, DSSS_K_PG_SUBTREE);
                    hipLaunchKernelGGL(pg_bwd_subtree_kernel, dim3(nbins), dim3(64), 0, st, d_binperm + bin_lo, d_binptr, d_bincols, d_colptr, d_rowidx, d_L, d_x, d_rdiag); }
                hipLaunchKernelGGL(pg_sep_delta_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, ns, d_t2, d_perm, d_x, d_delta1);
                if (nseg > 0) hipLaunchKernelGGL(pg_backsub_kernel, dim3((nseg + 31) / 32), dim3(256), 0, st, nseg, d_ord2, d_t2, d_C1, d_E1, d_Dl1, d_gi1, d_delta1, kp0, kp1);
                hipLaunchKernelGGL(pg_sep_delta_kernel, dim3((ns1 + 255) / 256), dim3(256), 0, st, ns1, d_sep1, (const int*)nullptr, d_delta1, d_delta);
                hipLaunchKernelGGL(pg_backsub_kernel, dim3((nseg1 + 31) / 32), dim3(256), 0, st, nseg1, d_ord1, d_sep1, d_C, d_E, d_Dl, d_gi, d_delta, mp0, mp1);
                hipLaunchKernelGGL(pg_linerr_kernel, dim3(nblk), dim3(256), 0, st, n, ne, W, d_ea, d_eb, d_eo, d_ew, d_r, d_Ji, d_delta, d_part, mp0, mp1);
                hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal + 1);
            }
            ++nfact;
            // X and Xn swap between trials, so these two stay outside the captured graph
            hipLaunchKernelGGL(pg_retract_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_X, d_delta, d_Xn);
            hipLaunchKernelGGL(pg_linearize_kernel, dim3(nblk), dim3(256), 0, st, n, ne, d_Xn, d_meas, W, d_ea, d_eb, d_eo, d_emeas, d_ew, d_r2, d_Ji2, d_part, mp0, mp1);
            hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal + 2);
            HCK(hipGetLastError());
            double h[3]; int failed = 0;
            { int rc2 = reduce_scalars(h, &failed); if (rc2) return rc2; }
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
            if (success) { std::swap(d_X, d_Xn); std::swap(d_r, d_r2); std::swap(d_Ji, d_Ji2); pre_lin = true; err = newErr; lambda /= c->pg.lambda_factor; ++iters; break; }
            else if (!stop) { lambda *= c->pg.lambda_factor; if (lambda >= c->pg.lambda_max) break; }
            else break;
        }
    } while (iters < c->pg.max_iters && !((err <= 0) || ((cur - err) / cur <= c->pg.rel_tol) || ((cur - err) <= c->pg.abs_tol)) && std::isfinite(cur));
    const auto T4 = std::chrono::steady_clock::now();
    const double t_lm = ms_since(T3);
    if (world > 1) {    // every rank holds its own poses (and the interface): zero the rest, sum -> the whole trajectory everywhere
        hipLaunchKernelGGL(pg_mask_own_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_X, mp0, mp1);
        int rc2 = dsss_comm_allreduce(c, (double*)d_X, (size_t)n * 12, st); if (rc2) { abandon(); return rc2; }
    }
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
    if (c->pg_online) {
        if (c->pg_warm_cap < (size_t)n) {
            HCK(hipStreamSynchronize(st));
            if (c->pg_warm) hipFree(c->pg_warm);
            c->pg_warm = nullptr; c->pg_warm_cap = 0; c->pg_warm_n = 0;
            const size_t cap = (size_t)n + (size_t)n / 2 + 1024;            // the graph grows by a frame per update
            HCK(hipMalloc(&c->pg_warm, cap * sizeof(pose_t)));
            c->pg_warm_cap = cap;
        }
        HCK(hipMemcpyAsync(c->pg_warm, d_X, (size_t)n * sizeof(pose_t), hipMemcpyDeviceToDevice, st));
        c->pg_warm_n = n;
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
    // the selected edges come back into a page-locked buffer the context keeps (a fresh 7 MB vector per solve cost a millisecond of
    // page faults, and a pageable destination halves the copy rate)
    const size_t ecap = (size_t)std::max(c->total_kp7, 1);
    if (c->pg_edges_cap < ecap) {
        if (c->pg_edges_host) hipHostFree(c->pg_edges_host);
        c->pg_edges_host = nullptr; c->pg_edges_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->pg_edges_host, ecap * sizeof(dsss_lc_edge), hipHostMallocDefault));
        c->pg_edges_cap = ecap;
    }
    dsss_lc_edge* edges_p = static_cast<dsss_lc_edge*>(c->pg_edges_host);
    std::vector<dsss_lc_edge> edges;              // only used when the edges of several ranks are merged
    int ne = 0;
    const double t_dr = ms(t0);
    const auto t1 = std::chrono::steady_clock::now();
    int rc = dsss_posegraph_select(c, nframes, edges_p, (int)ecap, &ne);
    if (rc) return rc;
    const int world = dsss_comm_world(c), rank = dsss_comm_rank(c);
    if (world > 1) {
        // every rank selected the loop closures of the pairs it matched (pairs go to the owner of the TARGET frame, so a target
        // ping's "last pair wins" choice is rank-local): exchange them with two small all-reduces (counts, then the records in
        // rank order = ascending target pose, the reference's loop order)
        std::vector<double> cnt(world, 0.0); cnt[rank] = ne;
        auto xch = [&](size_t n) -> int {                    // device scratch of the exchange, kept by the context (a hipMalloc / hipFree pair per call cost 0.3 ms)
            if (c->xch_cap >= n) return DSSS_OK;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(c->xch_dev); c->xch_dev = nullptr; c->xch_cap = 0;
            const size_t cap = n + n / 2 + 1024;
            HIPCHK(c, hipMalloc(&c->xch_dev, cap * sizeof(double))); c->xch_cap = cap;
            return DSSS_OK;
        };
        rc = xch(world); if (rc) return rc;
        double* d_tmp = c->xch_dev;
        hipError_t e = hipMemcpyAsync(d_tmp, cnt.data(), world * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) { rc = dsss_comm_allreduce(c, d_tmp, world, c->stream); if (rc) return rc; }
        if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(), d_tmp, world * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        HIPCHK(c, e);
        size_t off = 0, tot = 0;
        for (int r = 0; r < world; ++r) { if (r < rank) off += (size_t)cnt[r]; tot += (size_t)cnt[r]; }
        std::vector<double> rec(std::max<size_t>(tot, 1) * 20, 0.0);
        for (int i = 0; i < ne; ++i) {
            double* q = rec.data() + (off + i) * 20;
            q[0] = edges_p[i].a; q[1] = edges_p[i].b;
            for (int k = 0; k < 12; ++k) q[2 + k] = edges_p[i].rel[k];
            for (int k = 0; k < 6; ++k) q[14 + k] = edges_p[i].var[k];
        }
        if (tot > 0) {
            rc = xch(rec.size()); if (rc) return rc;
            d_tmp = c->xch_dev;
            e = hipMemcpyAsync(d_tmp, rec.data(), rec.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) { rc = dsss_comm_allreduce(c, d_tmp, tot * 20, c->stream); if (rc) return rc; }
            if (e == hipSuccess) e = hipMemcpyAsync(rec.data(), d_tmp, rec.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            HIPCHK(c, e);
        }
        edges.resize(std::max<size_t>(tot, 1)); ne = (int)tot;
        for (int i = 0; i < ne; ++i) {
            const double* q = rec.data() + (size_t)i * 20;
            edges[i].a = (int)q[0]; edges[i].b = (int)q[1];
            for (int k = 0; k < 12; ++k) edges[i].rel[k] = q[2 + k];
            for (int k = 0; k < 6; ++k) edges[i].var[k] = q[14 + k];
        }
        std::stable_sort(edges.begin(), edges.begin() + ne, [](const dsss_lc_edge& x, const dsss_lc_edge& y) { return x.b < y.b; });
        edges_p = edges.data();
    }
    const double t_sel = ms(t1);
    const auto t2 = std::chrono::steady_clock::now();
    rc = pg_solve_impl(c, nullptr, (int)total, edges_p, ne, poses12, stats4, rpy6, nframes);
    if (rc) return rc;
    if (getenv("DSSS_PG_VERBOSE")) fprintf(stderr, "[dsss pg] DR rows %.1f ms, LC selection %.1f ms, solve + download %.1f ms\n", t_dr, t_sel, ms(t2));
    return DSSS_OK;
}

/* N3, the online use of the solver (optimizer.cpp:134-139 ISAM2, :262-266 isam.update per ping + calculateEstimate): the
 * graph grows frame by frame.  Each call consumes the LC result set the context holds (if it has not been consumed yet) into
 * the accumulated edge list, and solves frames 0..nframes-1 by the batch LM STARTED FROM THE PREVIOUS ESTIMATE -- no factor is
 * carried over, the analysis and the factorisation are redone (they take milliseconds), but a converged prefix needs one
 * or two trials instead of the cold start's five.                                                                     */
int dsss_posegraph_reset(dsss_ctx* c)
{
    if (!c) return DSSS_E_ARG;
    c->pg_inc_edges.clear(); c->pg_inc_gen = c->lc_gen; c->pg_warm_n = 0;
    return DSSS_OK;
}

int dsss_posegraph_online_edges(dsss_ctx* c) { return c ? (int)c->pg_inc_edges.size() : DSSS_E_ARG; }

int dsss_posegraph_update(dsss_ctx* c, int nframes, double* poses12, double* rpy6, double* stats4)
{
    if (!c || nframes <= 0 || nframes > c->max_frames) return DSSS_E_ARG;
    if (dsss_comm_world(c) > 1) DSSS_FAIL(c, DSSS_E_STATE, "dsss_posegraph_update is a single-rank call (the online use is one vehicle, one GPU)");
    size_t total = 0;
    for (int f = 0; f < nframes; ++f) {
        if (!c->frames[f].has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no geometry", f);
        total += (size_t)c->frames[f].N;
    }
    if (c->has_lc && c->pg_inc_gen != c->lc_gen && c->total_kp7 > 0) {
        std::vector<dsss_lc_edge> fresh((size_t)c->total_kp7);
        int ne = 0;
        const int rc = dsss_posegraph_select(c, nframes, fresh.data(), (int)fresh.size(), &ne);
        if (rc) return rc;
        // a target ping keeps ONE loop closure, the latest (optimizer.cpp:203-258 within a call; across calls the later set wins)
        // (the accumulated edges are range-checked BEFORE they index anything: nframes may have gone down since the last update)
        for (const dsss_lc_edge& e : c->pg_inc_edges)
            if (e.a >= (int)total || e.b >= (int)total) DSSS_FAIL(c, DSSS_E_ARG, "an accumulated LC edge references ping %d of %zu: nframes went down; dsss_posegraph_reset first", std::max(e.a, e.b), total);
        if (ne > 0 && !c->pg_inc_edges.empty()) {
            std::vector<char> hit(total, 0);
            for (int e = 0; e < ne; ++e) hit[fresh[e].b] = 1;
            size_t w = 0;
            for (size_t e = 0; e < c->pg_inc_edges.size(); ++e) if (!hit[c->pg_inc_edges[e].b]) c->pg_inc_edges[w++] = c->pg_inc_edges[e];
            c->pg_inc_edges.resize(w);
        }
        c->pg_inc_edges.insert(c->pg_inc_edges.end(), fresh.begin(), fresh.begin() + ne);
        std::stable_sort(c->pg_inc_edges.begin(), c->pg_inc_edges.end(), [](const dsss_lc_edge& x, const dsss_lc_edge& y) { return x.b < y.b; });
    }
    c->pg_inc_gen = c->lc_gen;
    for (const dsss_lc_edge& e : c->pg_inc_edges)
        if (e.a >= (int)total || e.b >= (int)total) DSSS_FAIL(c, DSSS_E_ARG, "an accumulated LC edge references ping %d of %zu: nframes went down; dsss_posegraph_reset first", std::max(e.a, e.b), total);
    c->pg_online = true;
    const int rc = pg_solve_impl(c, nullptr, (int)total, c->pg_inc_edges.data(), (int)c->pg_inc_edges.size(), poses12, stats4, rpy6, nframes);
    c->pg_online = false;
    return rc;
}

} // extern "C"
