// diasss_amd/csrc/dsss_pg_chain.hip -- the pose CHAIN of the batch LM (dsss_pg.hip): factors and their linearisation, per-pose Hessian
// blocks, the two-level condensation of the chain onto the separators and its back-substitution, the scatter of the reduced system into
// the factor / the fronts' value array, what the ranks pack for the all-reduce, initial values (optimizer.cpp:150-160) and the
// trajectory rows (optimizer.cpp:1164-1214).  Everything here runs at -ffp-contract=off.
#include "dsss_pg_kernels.h"
#include "dsss_pg_dev.h"
// ------------------------------------------------------------------ factors
// factor k < n: k == 0 prior on X0 (measurement DR0), else Between(X_{k-1}, X_k); factor n + e: LC edge e.
// r = whitened residual, Ji = whitened Jacobian wrt the first pose (-W Ad(h^-1)); the Jacobian wrt the second
// pose is W itself (BetweenFactor with GTSAM_SLOW_BUT_CORRECT_BETWEENFACTOR off, PriorFactor H = I).
// WJ is a template parameter and the arrays are indexed by unrolled loops only: r and Ji stay in registers (with a run-time `Ji != nullptr` the
// compiler kept the 36 + 2 doubles in 304 bytes of private memory per thread -- 125 MB of scratch traffic per launch at C3, the reason the
// kernel wrote 243 MB for 138 MB of residuals and Jacobians).  Same operations in the same order: same bits.
template <bool WJ>
__device__ __forceinline__ void factor_eval(int k, int n, const pose_t* X, const pose_t* meas, const pg_weights& W,
                                            const int* ea, const int* eb, const pose_t* emeas, const double* ew,
                                            double (&r)[6], double (&Ji)[36])
{
    double xi[6];
    if (k == 0) {
        pose_t d;
        pose_between(&meas[0], &X[0], &d);
        pose_log(&d, xi);
#pragma unroll
        for (int a = 0; a < 6; ++a) r[a] = xi[a] * W.prior[a];
        if (WJ) {
#pragma unroll
            for (int a = 0; a < 36; ++a) Ji[a] = 0.0;
        }
        return;
    }
    int i, j; const pose_t* m; double w[6];
    if (k < n) {
        i = k - 1; j = k; m = &meas[k];
#pragma unroll
        for (int a = 0; a < 6; ++a) w[a] = W.odo[a];
    } else {
        const int e = k - n; i = ea[e]; j = eb[e]; m = &emeas[e];
#pragma unroll
        for (int a = 0; a < 6; ++a) w[a] = ew[(size_t)e * 6 + a];
    }
    pose_t h, er;
    pose_between(&X[i], &X[j], &h);
    pose_between(m, &h, &er);
    pose_log(&er, xi);
#pragma unroll
    for (int a = 0; a < 6; ++a) r[a] = xi[a] * w[a];
    if (WJ) {
        pose_t hi; double Ad[36];
        pose_inverse(&h, &hi);
        pose_adjoint(&hi, Ad);
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) Ji[a * 6 + b] = -Ad[a * 6 + b] * w[a];
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
template <bool WJ>        // WJ: residuals AND Jacobians (r, Ji both given); otherwise the error alone (r, Ji null)
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
        factor_eval<WJ>(k, n, X, meas, W, ea, eb, emeas, ew, rr, J);
#pragma unroll
        for (int a = 0; a < 6; ++a) { e2 += rr[a] * rr[a]; if (WJ) r[(size_t)k * 6 + a] = rr[a]; }
    }
    if (WJ) {
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

template __global__ void pg_linearize_kernel<false>(int, int, const pose_t* __restrict__, const pose_t* __restrict__, pg_weights, const int* __restrict__, const int* __restrict__, const int* __restrict__, const pose_t* __restrict__, const double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, int, int);
template __global__ void pg_linearize_kernel<true>(int, int, const pose_t* __restrict__, const pose_t* __restrict__, pg_weights, const int* __restrict__, const int* __restrict__, const int* __restrict__, const pose_t* __restrict__, const double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, int, int);

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
// `plist` (round 5): the blocks of the np listed poses only -- the ends of the level-1 chain, whose D, C, g the chain kernels read; the
// interior poses of the segments carry chain factors alone, and pg_segment_kernel<true> / pg_backsub_kernel<true> form their blocks
// from the Jacobians themselves (the same sums in the same order).  NULL: every pose.
__global__ __launch_bounds__(6 * PG_ASM_POSES) void pg_assemble_kernel(int n, pg_weights W, const double* __restrict__ r, const double* __restrict__ Ji,
                                                          const int* __restrict__ adj_ptr, const int* __restrict__ adj_edge,
                                                          const double* __restrict__ ew, const double* __restrict__ lambda_ptr,
                                                          double* __restrict__ D, double* __restrict__ C, double* __restrict__ g,
                                                          const int* __restrict__ eo, int mp0, int mp1, const int* __restrict__ plist, int np)
{
    // (a thread's six values of a block row are 48 contiguous bytes, the threads of a wavefront 48 bytes apart: stored directly, every
    // store instruction touched 24 cache lines for a sixth each.  The rows go through LDS and leave as 16-byte stores of whole lines.)
    __shared__ double s_dc[2][PG_ASM_POSES * 36];
    __shared__ double s_j[PG_ASM_POSES * 36];            // Jacobians of the chain factors i + 1 of the workgroup's poses: one contiguous 9 KB read
    const int slot_ = blockIdx.x * PG_ASM_POSES + threadIdx.x / 6, a = threadIdx.x % 6;
    const int i = plist ? (slot_ < np ? plist[slot_] : n) : slot_;
    const bool live = i < n;
    if (!plist) {
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
        const double* J = plist ? Ji + (size_t)(i + 1) * 36 : s_j + (threadIdx.x / 6) * 36; const double* rr = r + (size_t)(i + 1) * 6;
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
    if (plist) {                                               // scattered poses: the row straight out
#pragma unroll
        for (int b = 0; b < 6; ++b) { D[(size_t)i * 36 + a * 6 + b] = Dd[b]; C[(size_t)i * 36 + a * 6 + b] = Cc[b]; }
    }
    }       // live
    if (plist) return;
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
// back-substitution; outputs the end-point corrections.  Per pose the thirteen right-hand sides of D_i^-1 [E_i^T | C_i | g_i] are spread
// over the lanes of a group (each factorises the 6 x 6 pivot itself: cheaper than broadcasting the factor), and the products with E_i
// and C_i^T that follow are column-parallel as well.  One thread per segment took 24 us per pose (3 000 dependent f64 operations, 512
// registers and scratch).  Global memory is touched at ONE point of a step, its top: the blocks of the next step are requested there and
// consumed at the top of the next step, and the records of the back-substitution (E_i, g_i, and the factor of the PREVIOUS step, parked
// in LDS) are stored there, so the one wait on the memory counter per step finds everything a whole step old (dependent loads inside
// the step cost 8 000 of its 12 500 cycles).  E, D, G are double-buffered in LDS.  Rounds 2 - 4 ran SIXTEEN lanes per segment, one
// right-hand side each: ~540 vector instructions per step
struct pg_seg_lds { double E[2][36], D[2][36], G[2][6], C[36], L[36], Rv[6], Wo[6]; };      // 240 doubles (x 32 groups = 61 440 B)
// the lanes of a group sit in one wavefront, whose LDS operations execute in program order: waiting for the LDS queue
// (not for the global stores in flight -- a fence would) and keeping the compiler from moving memory operations across is enough
#define PG_COMPILER_FENCE() asm volatile("" ::: "memory")
#define PG_GROUP_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
// -- the 6 x 6 factor (150, the same on every lane), one pair of triangular solves (42) and two products (72) --
// for FOUR segments, thirteen of sixteen lanes busy and half of the second product thrown away.  EIGHT lanes per segment (round 5; eight
// segments per wavefront): lane c < 6 carries TWO right-hand
// sides -- row c of E_i (its E y accumulates column c of DL) and column c of C_i (its E y is column c of E_(i+1), its C^T y column c of the
// next pivot) -- lane 6 the gradient, lane 7 only helps to move blocks: the factor is computed once per EIGHT segments, the two solves of a
// lane are independent chains (the kernel is bound by dependent f64 latency at two wavefronts per SIMD), and a step is ~520 instructions
// for eight segments (124 -> 103 us per launch at C3).  Every right-hand side sees the arithmetic of the sixteen-lane kernel in the same
// order: the records are the same bits.
#define PG_SEG_LANES 8
// FROMJ (round 5, pass 1 on the pose chain): the blocks of the INTERIOR poses are not read but formed here -- an interior pose p carries
// the chain factors p and p + 1 and nothing else (every loop-closure pose is a separator), so with J = the Jacobian of factor p + 1
//     D_p = diag(w^2) + J^T J + lambda I,   C_p = J^T diag(w),   g_p = w r_p + J^T r_(p+1)
// in the order of operations of pg_assemble_kernel (which then runs on the 48 k ends of the level-1 chain instead of all 400 k poses: it
// wrote 230 MB per trial that this kernel read back).  The Jacobian of the NEXT pose is requested at the top of a step, parked in LDS at
// the step's end (in the slot of C_i, which is free by then) and turned into the lane's column of D, its entries of C and its component
// of g there; D_p is symmetric bit for bit (the products commute, the sums run over the same index), so column c is row c.
// D, C, g are read at the segment's LEFT END only (C_L couples it to the first interior pose).
template <bool FROMJ>
__device__ __forceinline__ void pg_seg_from_j(const double* __restrict__ Jl, const double* __restrict__ Rn, double rp, const double* __restrict__ Wl, double wa, double lambda,
                                              int c, double* __restrict__ nC, double* __restrict__ pre, double& gpre)
{
    if constexpr (FROMJ) {
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int a = c + PG_SEG_LANES * u; if (a < 36) { const int row = a / 6, col = a - 6 * row; nC[u] = Jl[col * 6 + row] * Wl[col]; } }
        if (c < 6) {
            double ja[6], sb[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < 6; ++q) ja[q] = Jl[q * 6 + c];
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int b = 0; b < 6; ++b) sb[b] += ja[q] * Jl[q * 6 + b];
#pragma unroll
            for (int b = 0; b < 6; ++b) { double d = b == c ? 0.0 + wa * wa : 0.0; d += sb[b]; if (b == c) d += lambda; pre[b] = d; }
            double gg = 0.0 + wa * rp, s2 = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) s2 += ja[q] * Rn[q];
            gpre = gg + s2;
        }
    }
}
template <bool FROMJ>
__global__ __launch_bounds__(256, 2) void pg_segment_kernel(int nseg, const int* __restrict__ seg_order, const int* __restrict__ sep_pose, const double* __restrict__ D,
                                                         const double* __restrict__ C, const double* __restrict__ g,
                                                         double* __restrict__ E, double* __restrict__ Dl, double* __restrict__ gi,
                                                         double* __restrict__ segDL, double* __restrict__ segDR, double* __restrict__ segGL,
                                                         double* __restrict__ segGR, double* __restrict__ segS, int* __restrict__ fail, int mp0, int mp1,
                                                         const double* __restrict__ rf, const double* __restrict__ Jf, pg_weights W, const double* __restrict__ lambda_ptr)
{
    __shared__ pg_seg_lds sh_all[256 / PG_SEG_LANES];
    const int grp = threadIdx.x / PG_SEG_LANES, c = threadIdx.x % PG_SEG_LANES;
    const int slot = blockIdx.x * (256 / PG_SEG_LANES) + grp;
    if (slot >= nseg) return;                                   // whole groups leave together
    const int s = seg_order[slot];                              // descending length: the eight segments of a wavefront run the same number of steps
    pg_seg_lds& sh = sh_all[grp];
    const int L = sep_pose[s], R = sep_pose[s + 1];
    if (L + 1 < mp0 || L + 1 >= mp1) return;
    if (R == L + 1) {
        for (int a = c; a < 36; a += PG_SEG_LANES) { segDL[(size_t)s * 36 + a] = 0; segDR[(size_t)s * 36 + a] = 0; segS[(size_t)s * 36 + a] = C[(size_t)L * 36 + a]; }
        if (c < 6) { segGL[(size_t)s * 6 + c] = 0; segGR[(size_t)s * 6 + c] = 0; }
        return;
    }
    double lambda = 0.0, rp = 0.0, rn = 0.0, jn[5] = { 0, 0, 0, 0, 0 }, gpre = 0.0, wa = 0.0;      // FROMJ: r_p[c], r_(p+1)[c], the lane's share of the next Jacobian, its component of g, its weight
    if constexpr (FROMJ) {
        lambda = *lambda_ptr;
        wa = c == 0 ? W.odo[0] : c == 1 ? W.odo[1] : c == 2 ? W.odo[2] : c == 3 ? W.odo[3] : c == 4 ? W.odo[4] : W.odo[5];      // (a lane-dependent index would put the argument into scratch)
        if (c < 6) sh.Wo[c] = wa;
        for (int a = c; a < 36; a += PG_SEG_LANES) { sh.E[0][a] = C[(size_t)L * 36 + a]; sh.C[a] = Jf[(size_t)(L + 2) * 36 + a]; }
        if (c < 6) { rp = rf[(size_t)(L + 1) * 6 + c]; rn = rf[(size_t)(L + 2) * 6 + c]; sh.Rv[c] = rn; }
    } else {
        for (int a = c; a < 36; a += PG_SEG_LANES) { sh.E[0][a] = C[(size_t)L * 36 + a]; sh.D[0][a] = D[(size_t)(L + 1) * 36 + a]; }
        if (c < 6) sh.G[0][c] = g[(size_t)(L + 1) * 6 + c];
    }
    // acc[]: column c of DL on lanes 0..5, GL on lane 6.  pre[]: the prefetched column c of D_(i+1) on lanes 0..5, g_(i+1) on lane 6 -- added
    // to the next step's pivot / gradient at its top (one load sequence for both kinds of lane: two divergent ones that write the same registers are serialised by a full wait).  nC: the lane's
    // share of C_(i+1).
    double nC[5], acc[6] = { 0, 0, 0, 0, 0, 0 }, pre[6] = { 0, 0, 0, 0, 0, 0 };
    const bool has_role = c < 7, is_col = c < 6;
    const int pre_stride = is_col ? 6 : 1;
    const double* pre_src = is_col ? D + c : g;
    const int pre_lds = is_col ? (int)(&sh.D[0][0] - &sh.E[0][0]) + c : (int)(&sh.G[0][0] - &sh.E[0][0]);     // offsets from sh.E[0] in doubles
    const int pre_flip = is_col ? 36 : 6;
    int cb = 0;
    if constexpr (FROMJ) {                                      // the first interior pose: D, g into LDS, its C into nC
#pragma unroll
        for (int u = 0; u < 5; ++u) nC[u] = 0.0;
        PG_GROUP_SYNC();
        double d0[6] = { 0, 0, 0, 0, 0, 0 };
        pg_seg_from_j<true>(sh.C, sh.Rv, rp, sh.Wo, wa, lambda, c, nC, d0, gpre);
        if (c < 6) {
#pragma unroll
            for (int a = 0; a < 6; ++a) sh.D[0][a * 6 + c] = d0[a];
            sh.G[0][c] = gpre;
            rp = rn;
        }
        gpre = 0.0;
        PG_GROUP_SYNC();
    } else {
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int a = c + PG_SEG_LANES * u; nC[u] = a < 36 ? C[(size_t)(L + 1) * 36 + a] : 0.0; }
    }
    for (int i = L + 1; i < R; ++i) {
        const bool last = (i + 1 == R);
        double* __restrict__ Ec = sh.E[cb]; double* __restrict__ Dc = sh.D[cb]; double* __restrict__ Gc = sh.G[cb];
        double* __restrict__ En = sh.E[cb ^ 1]; double* __restrict__ Dn = sh.D[cb ^ 1]; double* __restrict__ Gn = sh.G[cb ^ 1];
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int a = c + PG_SEG_LANES * u; if (a < 36) sh.C[a] = nC[u]; }
        if constexpr (FROMJ) {
            if (is_col) {                                        // completes the entries the last step left in D and G: column c of D_i, component c of g_i
#pragma unroll
                for (int a = 0; a < 6; ++a) Dc[a * 6 + c] += pre[a];
                Gc[c] += gpre;
            }
            if (!last) {                                         // the Jacobian of factor i + 2 and r_(i+2): pose i + 1's blocks, formed at the end of this step
#pragma unroll
                for (int u = 0; u < 5; ++u) { const int a = c + PG_SEG_LANES * u; if (a < 36) jn[u] = Jf[(size_t)(i + 2) * 36 + a]; }
                if (is_col) rn = rf[(size_t)(i + 2) * 6 + c];
            }
        } else {
        if (has_role) {                                          // completes the entries the last step left in D and G
            double* dst = &sh.E[0][0] + pre_lds + cb * pre_flip;
#pragma unroll
            for (int a = 0; a < 6; ++a) dst[a * pre_stride] += pre[a];
        }
        if (!last) {
#pragma unroll
            for (int u = 0; u < 5; ++u) { const int a = c + PG_SEG_LANES * u; if (a < 36) nC[u] = C[(size_t)(i + 1) * 36 + a]; }
            if (has_role) {
                const double* src = pre_src + (size_t)(i + 1) * (is_col ? 36 : 6);
#pragma unroll
                for (int a = 0; a < 6; ++a) pre[a] = src[a * pre_stride];
            }
        } else if (has_role) {
#pragma unroll
            for (int a = 0; a < 6; ++a) pre[a] = 0.0;
        }
        }
        if (i > L + 1) for (int a = c; a < 36; a += PG_SEG_LANES) Dl[(size_t)(i - 1) * 36 + a] = sh.L[a];
        PG_GROUP_SYNC();
        for (int a = c; a < 36; a += PG_SEG_LANES) E[(size_t)i * 36 + a] = Ec[a];
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
            for (int q = 0; q < 6; ++q) {
                double ya1 = yA[q], yb1 = yB[q];
                if constexpr (FROMJ) asm volatile("" : "+v"(ya1), "+v"(yb1));      // (register values: the optimiser otherwise selects between the two ARRAYS through scratch memory)
                yc[q] = is_col ? yb1 : ya1;
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double u = 0;
#pragma unroll
                for (int q = 0; q < 6; ++q) u += sh.C[q * 6 + a] * yc[q];
                if (is_col) Dn[a * 6 + c] = -u;
                else if (c == 6) Gn[a] = -u;
            }
        }
        if constexpr (FROMJ) {
            if (!last) {                                         // C_i has been used for the last time: its slot takes the next Jacobian
                PG_GROUP_SYNC();
#pragma unroll
                for (int u = 0; u < 5; ++u) { const int a = c + PG_SEG_LANES * u; if (a < 36) sh.C[a] = jn[u]; }
                if (is_col) sh.Rv[c] = rn;
                PG_GROUP_SYNC();
                pg_seg_from_j<true>(sh.C, sh.Rv, rp, sh.Wo, wa, lambda, c, nC, pre, gpre);
                rp = rn;
            } else {
#pragma unroll
                for (int a = 0; a < 6; ++a) pre[a] = 0.0;
                gpre = 0.0;
            }
        }
        cb ^= 1;
        PG_GROUP_SYNC();
    }
    for (int a = c; a < 36; a += PG_SEG_LANES) Dl[(size_t)(R - 1) * 36 + a] = sh.L[a];
    if (c < 6) for (int a = 0; a < 6; ++a) segDL[(size_t)s * 36 + a * 6 + c] = acc[a];
    if (c == 6) for (int a = 0; a < 6; ++a) segGL[(size_t)s * 6 + a] = acc[a];
    for (int a = c; a < 36; a += PG_SEG_LANES) { segDR[(size_t)s * 36 + a] = sh.D[cb][a]; segS[(size_t)s * 36 + a] = sh.E[cb][a]; }
    if (c < 6) segGR[(size_t)s * 6 + c] = sh.G[cb][c];
}
template __global__ void pg_segment_kernel<false>(int, const int* __restrict__, const int* __restrict__, const double* __restrict__, const double* __restrict__, const double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, int* __restrict__, int, int, const double* __restrict__, const double* __restrict__, pg_weights, const double* __restrict__);
template __global__ void pg_segment_kernel<true>(int, const int* __restrict__, const int* __restrict__, const double* __restrict__, const double* __restrict__, const double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, int* __restrict__, int, int, const double* __restrict__, const double* __restrict__, pg_weights, const double* __restrict__);

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

__global__ __launch_bounds__(256) void pg_sep_delta_kernel(int ns, const int* __restrict__ sep_pose, const int* __restrict__ perm,
                                                           const double* __restrict__ x, double* __restrict__ delta)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns) return;
    const int src = perm ? perm[k] : k;
    if (src < 0) return;                                     // (rank-local analysis: another rank's interior separator -- not solved here, not read here)
    for (int a = 0; a < 6; ++a) delta[(size_t)sep_pose[k] * 6 + a] = x[(size_t)src * 6 + a];
}

// interiors, right to left: delta_i = D_i^-1 (-g_i - E_i^T delta_L - C_i delta_{i+1}).  EIGHT LANES PER SEGMENT: lane a < 6 forms
// component a of the right-hand side (column a of E_i, row a of C_i: the group reads the 288-byte blocks together; one thread
// per segment read them alone, 8.7 us per pose), the six components are exchanged, and every lane solves the 6 x 6 system itself,
// which leaves delta_i in all of them for the next step.  The blocks of pose i - 1 are requested before pose i is computed.
#define PG_BS_LANES 8
struct pg_bs_blk { double Ec[6], Cr[6], gv, Lm[21]; };
// FROMJ (pass 1 on the pose chain): row aa of C_i = column aa of the Jacobian of factor i + 1, times the weights -- the product
// pg_assemble_kernel forms (it no longer writes the blocks of interior poses, see pg_segment_kernel<true>)
template <bool FROMJ>
__device__ __forceinline__ void pg_bs_load(pg_bs_blk& B, int i, int aa, const double* __restrict__ C, const double* __restrict__ E,
                                           const double* __restrict__ Dl, const double* __restrict__ gi, const double* __restrict__ Jf, const pg_weights& W)
{
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        B.Ec[q] = E[(size_t)i * 36 + q * 6 + aa];
        if constexpr (FROMJ) B.Cr[q] = Jf[(size_t)(i + 1) * 36 + q * 6 + aa] * W.odo[q];
        else B.Cr[q] = C[(size_t)i * 36 + aa * 6 + q];
    }
    B.gv = gi[(size_t)i * 6 + aa];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int k = 0; k <= r; ++k) B.Lm[r * (r + 1) / 2 + k] = Dl[(size_t)i * 36 + r * 6 + k];
}
template <bool FROMJ>
__global__ __launch_bounds__(256) void pg_backsub_kernel(int nseg, const int* __restrict__ seg_order, const int* __restrict__ sep_pose, const double* __restrict__ C,
                                                         const double* __restrict__ E, const double* __restrict__ Dl, const double* __restrict__ gi,
                                                         double* __restrict__ delta, int mp0, int mp1, const double* __restrict__ Jf, pg_weights W)
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
    pg_bs_load<FROMJ>(cur, R - 1, aa, C, E, Dl, gi, Jf, W);
    for (int i = R - 1; i > L; --i) {
        if (i - 1 > L) pg_bs_load<FROMJ>(nxt, i - 1, aa, C, E, Dl, gi, Jf, W);
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
template __global__ void pg_backsub_kernel<false>(int, const int* __restrict__, const int* __restrict__, const double* __restrict__, const double* __restrict__, const double* __restrict__, const double* __restrict__, double* __restrict__, int, int, const double* __restrict__, pg_weights);
template __global__ void pg_backsub_kernel<true>(int, const int* __restrict__, const int* __restrict__, const double* __restrict__, const double* __restrict__, const double* __restrict__, const double* __restrict__, double* __restrict__, int, int, const double* __restrict__, pg_weights);

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
// poses picked out of a trajectory by index (the frozen end points of a window update)
__global__ __launch_bounds__(256) void pg_gather_pose_kernel(int n, const int* __restrict__ idx, const pose_t* __restrict__ X, pose_t* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = X[idx[i]];
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

