// diasss_amd/csrc/dsss_lc.hip -- batched loop-closure measurements: one 15-DoF Levenberg-Marquardt problem per
// matched keypoint pair.  Restates Optimizer::LoopClosingTFs, graph_option = 0
// (/root/reference/src/core/optimizer.cpp:641-982): graph {Prior(X1, 1e-6), Between(X1, X2; DR odometry, adaptive
// sigmas :778), SssPoint(L1, X1), SssPoint(L1, X2)} (:773-786), GTSAM LM with default parameters (:815-822), marginal
// covariance of X2 (:956-959), score = ini/final - 2 (:853-896).  GTSAM semantics per SURVEY.md A.2/A.3 and
// oracle/orc_lc.c.  f64 VALU: 15 x 15 systems are far too small for MFMA; 16 lanes per problem.
#include "dsss_internal.h"
#include "dsss_pose.h"

#define MR 16
#define MD 15

struct mini_prob {
    pose_t prior, odo;
    double sig_prior[6], sig_odo[6], sig_s[2], sig_t[2];
    double slant_s, slant_t;
};
struct mini_val { double L[3]; pose_t X1, X2; };

// ---- 16 lanes per problem, four problems per wavefront.  The pose algebra (a few hundred dependent flops) is evaluated
// redundantly by every lane of the group; the 16 x 15 Jacobian and the 15 x 15 systems live in LDS with one row per lane,
// and every sum keeps the element-wise order of oracle/orc_lc.c (k ascending), so results equal the one-thread version
// bit for bit while the serial chains shrink from O(15^3) to O(15^2) and nothing spills to scratch.
#define LG 16                  // lanes per problem
#define LS 16                  // LDS row stride (doubles)
// The problem's constants and its current / trial values live in LDS too: every lane of the group holds the same 100 doubles,
// and kept in registers (200 VGPRs of the 512) they were what held the kernel at one wavefront per SIMD with 110 spills.
// H and its Cholesky factor share one 15 x 16 array: H keeps its lower triangle ([i][j], j <= i), the factor goes into the other half
// transposed and shifted by a column (L(i, k), k <= i, at [k][i + 1]) -- H survives the retries of a trial with a larger lambda, and
// 4.9 KB per problem instead of 6.8 let eight workgroups share a compute unit's LDS (two wavefronts per SIMD).
#define LIDX(i, k) ((k) * LS + (i) + 1)
struct lc_lds { double J[MR * LS]; double H[MD * LS]; double r[MR]; mini_prob m; mini_val v, nv; };

// The pose algebra of a problem is a chain of a few hundred dependent f64 operations (two Logmaps with acos / sin / tan, two sss
// factors with a square root and divisions) that every lane of the group used to run in full: 75 % of the kernel's cycles (in-kernel
// stamps, round 3).  The factors come in PAIRS of the same code on different data -- prior / between (Logmap of a relative pose),
// source / target sss factor, retraction of X1 / X2, the two DR poses of the set-up -- so lanes 0..7 of the group run the first of
// a pair and lanes 8..15 the second, at the same time, and the results cross by one shuffle each.  Every value is produced by the
// same operations in the same order as before: the output is bit-identical.
__device__ inline void lc_bcast_pose(const pose_t& mine, int src, pose_t* out)
{
#pragma unroll
    for (int k = 0; k < 9; ++k) out->R[k] = __shfl(mine.R[k], src, LG);
#pragma unroll
    for (int k = 0; k < 3; ++k) out->t[k] = __shfl(mine.t[k], src, LG);
}
__device__ inline void lc_select_pose(bool second, const pose_t& a, const pose_t& b, pose_t* out)
{
#pragma unroll
    for (int k = 0; k < 9; ++k) out->R[k] = second ? b.R[k] : a.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) out->t[k] = second ? b.t[k] : a.t[k];
}
// whitened residual r and Jacobian J.  With J: both go to LDS (lane `lane` clears row `lane` of J, lanes 0 and 8 write the entries of their
// halves and their halves of r).  Without J (error evaluation): r in registers on every lane.
__device__ static void mini_lin(const mini_prob& m, const mini_val& v, double* r, double* J, int lane)
{
    const bool hb = (lane & 8) != 0;                       // second half of the group: between factor, target's sss factor
    const bool wr = J && (lane & 7) == 0;
    if (J) {
#pragma unroll
        for (int c2 = 0; c2 < LS; ++c2) J[lane * LS + c2] = 0.0;
        __builtin_amdgcn_wave_barrier();
    }
    // first half:  PriorFactor   e = Logmap(prior^-1 X1), H = I
    // second half: BetweenFactor e = Logmap(meas^-1 h), h = X1^-1 X2, H1 = -Ad(h^-1), H2 = I
    pose_t t1, t2, E; double xi[6], rr[6];
    const pose_t* P = hb ? &v.X1 : &m.prior;               // (lane-dependent LDS addresses: one load per element)
    const pose_t* Q = hb ? &v.X2 : &v.X1;
    pose_between(P, Q, &t1);
    pose_between(&m.odo, &t1, &t2);                        // (unused by the first half)
    lc_select_pose(hb, t1, t2, &E);
    pose_log(&E, xi);
#pragma unroll
    for (int i = 0; i < 6; ++i) { const double sg = hb ? m.sig_odo[i] : m.sig_prior[i]; rr[i] = xi[i] / sg; }
    if (!J) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { r[i] = __shfl(rr[i], 0, LG); r[6 + i] = __shfl(rr[i], 8, LG); }
    } else if (wr) {                                       // the linearisation keeps its residual in LDS (r = the group's S.r): sixteen registers less across the solve
#pragma unroll
        for (int i = 0; i < 6; ++i) r[(hb ? 6 : 0) + i] = rr[i];
    }
    if (J) {
        pose_t hi; double Ad[36];
        pose_inverse(&t1, &hi);                            // t1 = h on the second half
        pose_adjoint(&hi, Ad);
        if (wr && !hb) {
#pragma unroll
            for (int i = 0; i < 6; ++i) J[i * LS + 3 + i] = 1.0 / m.sig_prior[i];
        }
        if (wr && hb) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int j = 0; j < 6; ++j) J[(6 + i) * LS + 3 + j] = -Ad[6 * i + j] / m.sig_odo[i];
                J[(6 + i) * LS + 9 + i] = 1.0 / m.sig_odo[i];
            }
        }
    }
    // first half: SssPoint(L1, X1), second half: SssPoint(L1, X2)
    double ee[2], H1[6], H2[12], r2[2];
    const pose_t* X = hb ? &v.X2 : &v.X1;
    const double slant = hb ? m.slant_t : m.slant_s;
    const double sg0 = hb ? m.sig_t[0] : m.sig_s[0], sg1 = hb ? m.sig_t[1] : m.sig_s[1];
    sss_factor(v.L, X, slant, 0.0, ee, J ? H1 : nullptr, H2);
    r2[0] = ee[0] / sg0; r2[1] = ee[1] / sg1;
    if (!J) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { r[12 + i] = __shfl(r2[i], 0, LG); r[14 + i] = __shfl(r2[i], 8, LG); }
    }
    if (wr) {
        const int row0 = hb ? 14 : 12, col0 = hb ? 9 : 3;
        r[row0] = r2[0]; r[row0 + 1] = r2[1];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const double sg = i ? sg1 : sg0;
#pragma unroll
            for (int j = 0; j < 3; ++j) J[(row0 + i) * LS + j] = H1[3 * i + j] / sg;
#pragma unroll
            for (int j = 0; j < 6; ++j) J[(row0 + i) * LS + col0 + j] = H2[6 * i + j] / sg;
        }
    }
    if (J) __builtin_amdgcn_wave_barrier();
}
__device__ static double mini_err(const mini_prob& m, const mini_val& v, int lane)
{
    double r[MR];
    mini_lin(m, v, r, nullptr, lane);
    double s = 0;
#pragma unroll
    for (int i = 0; i < MR; ++i) s += r[i] * r[i];
    return 0.5 * s;
}
// lane a < 15: row a of H = J^T J up to the diagonal, and g[a] = (J^T r)[a]
__device__ static double normal_eq(const double* J, const double* r, double* H, int lane)
{
    double g = 0;
    if (lane < MD) {
        double ca[MR];
#pragma unroll
        for (int k = 0; k < MR; ++k) ca[k] = J[k * LS + lane];
#pragma unroll
        for (int k = 0; k < MR; ++k) g += ca[k] * r[k];
        for (int b2 = 0; b2 < MD; ++b2) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < MR; ++k) t += ca[k] * J[k * LS + b2];
            if (b2 <= lane) H[lane * LS + b2] = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_wave_barrier();
    return g;
}
// lower Cholesky of A = H + lambda I into L (LDS, the LIDX half of the array H lives in): lane i owns row i in registers; returns 0 on success (uniform
// over the group).  Column j: lanes i >= j subtract sum_k L(i,k) L(j,k), k ascending, exactly as the sequential form.
__device__ static int chol15(const double* H, double lambda, double* L, int lane)
{
    double row[MD];
    const int li = lane < MD ? lane : MD - 1;               // lane 15 shadows row 14 and never writes
#pragma unroll
    for (int j = 0; j < MD; ++j) row[j] = j <= li ? H[li * LS + j] + (j == li ? lambda : 0.0) : 0.0;      // (the entries right of the diagonal are never used)
    int bad = 0;
#pragma unroll
    for (int j = 0; j < MD; ++j) {
        double sacc = row[j];
#pragma unroll
        for (int k = 0; k < MD; ++k) if (k < j) sacc -= row[k] * L[LIDX(j, k)];
        const double dj = __shfl(sacc, j, LG);              // pivot before the square root, from the diagonal lane
        if (!(dj > 0) || !isfinite(dj)) { bad = 1; break; }
        const double dq = sqrt(dj);
        row[j] = li == j ? dq : sacc / dq;
        if (lane < MD && li >= j) L[LIDX(li, j)] = row[j];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    return bad;
}
// b <- (L L^T)^-1 b, b in registers of every lane (same values on every lane of the group), L read from LDS
__device__ static void chol15_solve(const double* L, double* b)
{
#pragma unroll
    for (int i = 0; i < MD; ++i) {
        double sacc = b[i];
#pragma unroll
        for (int k = 0; k < MD; ++k) if (k < i) sacc -= L[LIDX(i, k)] * b[k];
        b[i] = sacc / L[LIDX(i, i)];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = MD - 1; i >= 0; --i) {
        double sacc = b[i];
#pragma unroll
        for (int k = 0; k < MD; ++k) if (k > i) sacc -= L[LIDX(k, i)] * b[k];
        b[i] = sacc / L[LIDX(i, i)];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// kp7: n x 7; per problem: frame pointers (pose6 / alt / gr of source and target), M of both, flip flags
__global__ __launch_bounds__(64, 2) void lc_kernel(const double* __restrict__ kp7, int n,
                                                const int* __restrict__ kp7_pair, const uint8_t* __restrict__ kp7_flip,
                                                const int* __restrict__ act_s, const int* __restrict__ act_t,
                                                int single_s, int single_t, int single_flip,
                                                const double* const* __restrict__ alt_ptr, const double* const* __restrict__ gr_ptr,
                                                const double* const* __restrict__ pose_ptr, const int* __restrict__ fcols,
                                                dsss_lc* __restrict__ out)
{
    __shared__ lc_lds s_all[64 / LG];
    const int grp = threadIdx.x / LG, lane = threadIdx.x % LG;
    const int i = blockIdx.x * (64 / LG) + grp;
    if (i >= n) return;                                     // whole groups leave together
    lc_lds& S = s_all[grp];
    const double PI = DSSS_PI_REF;
    const double* kp = kp7 + (size_t)i * 7;
    const int fs = kp7_pair ? act_s[kp7_pair[i]] : single_s, ft = kp7_pair ? act_t[kp7_pair[i]] : single_t;
    int flip = kp7_flip ? kp7_flip[i] : 0;
    const double* pose_s = pose_ptr[fs]; const double* pose_t_ = pose_ptr[ft];
    const int id_s = (int)kp[0], id_t = (int)kp[3];
    if (!kp7_flip) {
        // stand-alone call: the sticky compensation (optimizer.cpp:650,700-703) is resolved over the caller's list
        const double thr = 2 * PI / 3;
        for (int k = 0; k <= i; ++k) {
            const double* q = kp7 + (size_t)k * 7;
            if (fabs(pose_s[(size_t)(int)q[0] * 6 + 2]) > thr) flip |= 1;
            if (fabs(pose_t_[(size_t)(int)q[3] * 6 + 2]) > thr) flip |= 2;
        }
        (void)single_flip;
    }
    pose_t cps_s, cps_t;
    pose_identity(&cps_s); pose_identity(&cps_t);
    const double flipv[3] = { 0, 0, PI };
    if (flip & 1) so3_exp(flipv, cps_s.R);
    if (flip & 2) so3_exp(flipv, cps_t.R);
    const double sigma_r = 0.1, alpha_bw = 0.1 * PI / 180;     // :685
    mini_prob m;                                            // built in registers, parked in LDS below (S.m)
    m.slant_s = kp[2]; m.slant_t = kp[5];
    m.sig_s[0] = sigma_r; m.sig_s[1] = kp[2] * alpha_bw;
    m.sig_t[0] = sigma_r; m.sig_t[1] = kp[5] * alpha_bw;
    pose_t Tp_s, Tp_t, Tp_st;
    const bool hb0 = (lane & 8) != 0;                       // lanes 0..7: the source ping's pose and geo sample, lanes 8..15: the target's
    {
        pose_t Pm, Cm, Tm;
        pose_from_rodrigues(hb0 ? pose_t_ + (size_t)id_t * 6 : pose_s + (size_t)id_s * 6, &Pm);
        lc_select_pose(hb0, cps_s, cps_t, &Cm);
        pose_compose(&Pm, &Cm, &Tm);
        lc_bcast_pose(Tm, 0, &Tp_s); lc_bcast_pose(Tm, 8, &Tp_t);
    }
    pose_between(&Tp_s, &Tp_t, &Tp_st);
    for (int k = 0; k < 6; ++k) m.sig_prior[k] = 0.000001;
    m.sig_odo[0] = 0.1 * PI / 180; m.sig_odo[1] = 0.1 * PI / 180; m.sig_odo[2] = 0.5 * PI / 180;          // :778
    m.sig_odo[3] = fabs(Tp_st.t[0] * 2); m.sig_odo[4] = fabs(Tp_st.t[1] / 10); m.sig_odo[5] = 0.1;
    for (int k = 3; k < 5; ++k) if (m.sig_odo[k] < 1e-9) m.sig_odo[k] = 1e-9;
    m.prior = Tp_s; m.odo = Tp_st;
    const int Ms = fcols[fs], Mt = fcols[ft];
    const int id_ss = (int)kp[1], id_tt = (int)kp[4];
    double gsx, gsy, gtx, gty;
    {
        double gx, gy;
        dsss_geo_at(hb0 ? pose_t_ : pose_s, hb0 ? gr_ptr[ft] : gr_ptr[fs], hb0 ? Mt : Ms, hb0 ? id_t : id_s, hb0 ? id_tt : id_ss, &gx, &gy);
        gsx = __shfl(gx, 0, LG); gsy = __shfl(gy, 0, LG); gtx = __shfl(gx, 8, LG); gty = __shfl(gy, 8, LG);
    }
    mini_val v;
    v.L[0] = (gsx + gtx) / 2; v.L[1] = (gsy + gty) / 2;                                                    // :792-795
    v.L[2] = ((pose_s[(size_t)id_s * 6 + 5] - alt_ptr[fs][id_s]) + (pose_t_[(size_t)id_t * 6 + 5] - alt_ptr[ft][id_t])) / 2;
    v.X1 = Tp_s; v.X2 = Tp_t;
    if (lane == 0) { S.m = m; S.v = v; }
    __builtin_amdgcn_wave_barrier();                        // the group's lanes sit in one wavefront: its LDS operations execute in program order
    const mini_prob& M_ = S.m; const mini_val& V_ = S.v;
    // ---- LevenbergMarquardtOptimizer::optimize, default params
    const double relTol = 1e-5, absTol = 1e-5, lamMax = 1e5, minFid = 1e-3;
    double lambda = 1e-5;
    int iters = 0;
    double err = mini_err(M_, V_, lane);
    const double err0 = err;
    double d[MD];
    const double* r = S.r;                                  // (LDS; written by mini_lin)
    if (err > 0) {
        double cur;
        do {
            cur = err;
            mini_lin(M_, V_, S.r, S.J, lane);
            const double g_mine = normal_eq(S.J, r, S.H, lane);
            double oldLin = 0;
#pragma unroll
            for (int k = 0; k < MR; ++k) oldLin += r[k] * r[k];
            oldLin *= 0.5;
            for (;;) {
                const bool ok = chol15(S.H, lambda, S.H, lane) == 0;
                bool success = false, stop = false;
                double newErr = 0;
                if (ok) {
#pragma unroll
                    for (int a = 0; a < MD; ++a) d[a] = -__shfl(g_mine, a, LG);
                    chol15_solve(S.H, d);
                    double sk = r[lane];                                // lane k: row k of J d + r
#pragma unroll
                    for (int a = 0; a < MD; ++a) sk += S.J[lane * LS + a] * d[a];
                    double newLin = 0;
#pragma unroll
                    for (int k = 0; k < MR; ++k) { const double t = __shfl(sk, k, LG); newLin += t * t; }
                    newLin *= 0.5;
                    const double linChange = oldLin - newLin;
                    if (linChange >= 0) {
                        {   // the trial values go to LDS (S.nv); the registers that held them are free again afterwards
                            const bool hb = (lane & 8) != 0;                  // lanes 0..7 retract X1, lanes 8..15 X2
                            pose_t Xn; double dx[6];
                            const pose_t* X = hb ? &V_.X2 : &V_.X1;
#pragma unroll
                            for (int a = 0; a < 6; ++a) dx[a] = hb ? d[9 + a] : d[3 + a];
                            pose_retract(X, dx, &Xn);
                            if (lane == 0) { for (int a = 0; a < 3; ++a) S.nv.L[a] = V_.L[a] + d[a]; S.nv.X1 = Xn; }
                            if (lane == 8) S.nv.X2 = Xn;
                            __builtin_amdgcn_wave_barrier();
                        }
                        newErr = mini_err(M_, S.nv, lane);
                        const double costChange = err - newErr;
                        if (linChange > 2.220446049250313e-16 * oldLin) success = (costChange / linChange) > minFid;
                        if (fabs(costChange) < relTol * err) stop = true;
                    }
                }
                if (success) {
                    {   // v = nv, element-wise by the lanes of the group (27 doubles)
                        const double* src = reinterpret_cast<const double*>(&S.nv); double* dst = reinterpret_cast<double*>(&S.v);
                        const double e0 = src[lane], e1 = lane + LG < (int)(sizeof(mini_val) / sizeof(double)) ? src[lane + LG] : 0.0;
                        __builtin_amdgcn_wave_barrier();
                        dst[lane] = e0; if (lane + LG < (int)(sizeof(mini_val) / sizeof(double))) dst[lane + LG] = e1;
                        __builtin_amdgcn_wave_barrier();
                    }
                    err = newErr; lambda /= 10; ++iters; break;
                }
                else if (!stop) { lambda *= 10; if (lambda >= lamMax) break; }
                else break;
            }
        } while (iters < 100 && !((err <= 0) || ((cur - err) / cur <= relTol) || ((cur - err) <= absTol)) && isfinite(cur));
    }
    dsss_lc o;
    o.iters = iters; o.pad_ = 0; o.err0 = err0; o.err1 = err;
    // eval_1 (:853-896)
    pose_t cti, new_pose;
    {   // the yaw compensations are rebuilt from the flags here rather than kept in registers across the LM loop
        pose_t cq; pose_identity(&cq);
        if (flip & 2) so3_exp(flipv, cq.R);
        pose_inverse(&cq, &cti);
    }
    pose_compose(&V_.X2, &cti, &new_pose);
    const double x_o = gsx - gtx, y_o = gsy - gty;
    const double ini = sqrt(x_o * x_o + y_o * y_o);
    double rpy[3];
    pose_rpy(&new_pose, rpy);
    const double* gr_t = gr_ptr[ft];
    double lx, ly;
    if (kp[4] < Mt / 2) {
        const int gi = Mt / 2 - (int)kp[4];
        lx = new_pose.t[0] + gr_t[gi] * cos(rpy[2] + PI / 2 - PI);
        ly = new_pose.t[1] + gr_t[gi] * sin(rpy[2] + PI / 2 - PI);
    } else {
        const int gi = (int)kp[4] - Mt / 2;
        lx = new_pose.t[0] + gr_t[gi] * cos(rpy[2] - PI / 2 - PI);
        ly = new_pose.t[1] + gr_t[gi] * sin(rpy[2] - PI / 2 - PI);
    }
    const double x_n = gsx - lx, y_n = gsy - ly;
    const double fin = sqrt(x_n * x_n + y_n * y_n);
    o.score = ini / fin - 2;
    // Marginals(graph, result).marginalCovariance(X2).diagonal() (:956-959): lane c < 6 solves for unit vector 9 + c
    mini_lin(M_, V_, S.r, S.J, lane);
    (void)normal_eq(S.J, r, S.H, lane);
    double var_mine = NAN;
    if (chol15(S.H, 0.0, S.H, lane) == 0) {
#pragma unroll
        for (int a = 0; a < MD; ++a) d[a] = (a == 9 + lane) ? 1.0 : 0.0;
        chol15_solve(S.H, d);
        var_mine = 0;
#pragma unroll
        for (int a = 9; a < MD; ++a) if (a == 9 + lane) var_mine = d[a];
    }
#pragma unroll
    for (int c2 = 0; c2 < 6; ++c2) o.var[c2] = __shfl(var_mine, c2, LG);
    pose_t csi, src, rel;
    {
        pose_t cq; pose_identity(&cq);
        if (flip & 1) so3_exp(flipv, cq.R);
        pose_inverse(&cq, &csi);
    }
    pose_compose(&M_.prior, &csi, &src);                         // m.prior = Tp_s
    pose_between(&src, &new_pose, &rel);                         // :958
    for (int a = 0; a < 9; ++a) o.rel[a] = rel.R[a];
    for (int a = 0; a < 3; ++a) o.rel[9 + a] = rel.t[a];
    if (lane == 0) out[i] = o;
}

// ------------------------------------------------------------------ a21 / a23: landmark triangulation
// LMTriaFactor (LMtriangulatefactor.cpp:10-27: residual and 2x3 Jacobian = those of SssPointFactor with the pose held
// fixed) inside Optimizer::TriangulateOneLandmark (optimizer.cpp:984-1021): 3-DoF GTSAM LM on the landmark, point prior
// (10, 10, |xy baseline| / 100).  Call site :907-921 (eval_2): DR poses with the sticky yaw compensation, landmark
// initialised as in :789-795; the four consistency figures printed there are returned next to the point.
// One thread per problem: 7 x 3 Jacobian and a 3 x 3 system in registers; same operation order as oracle/orc_lc.c.
__device__ static void tri_lin(const pose_t& Tp_s, const pose_t& Tp_t, double slant_s, double slant_t, const double* sig_s, const double* sig_t,
                               const double* sig_p, const double* ini, const double* p, double* r, double* J)
{
    double ee[2], H1[6], H2[12];
    sss_factor(p, &Tp_s, slant_s, 0.0, ee, J ? H1 : nullptr, H2);
    for (int i = 0; i < 2; ++i) { r[i] = ee[i] / sig_s[i]; if (J) for (int j = 0; j < 3; ++j) J[i * 3 + j] = H1[3 * i + j] / sig_s[i]; }
    sss_factor(p, &Tp_t, slant_t, 0.0, ee, J ? H1 : nullptr, H2);
    for (int i = 0; i < 2; ++i) { r[2 + i] = ee[i] / sig_t[i]; if (J) for (int j = 0; j < 3; ++j) J[(2 + i) * 3 + j] = H1[3 * i + j] / sig_t[i]; }
    for (int i = 0; i < 3; ++i) { r[4 + i] = (p[i] - ini[i]) / sig_p[i]; if (J) for (int j = 0; j < 3; ++j) J[(4 + i) * 3 + j] = (i == j) ? 1.0 / sig_p[i] : 0.0; }
}
__device__ static int tri_chol3(double* A)
{
    for (int j = 0; j < 3; ++j) {
        double d = A[j * 3 + j];
        for (int k = 0; k < j; ++k) d -= A[j * 3 + k] * A[j * 3 + k];
        if (!(d > 0) || !isfinite(d)) return -1;
        d = sqrt(d); A[j * 3 + j] = d;
        for (int i = j + 1; i < 3; ++i) { double t = A[i * 3 + j]; for (int k = 0; k < j; ++k) t -= A[i * 3 + k] * A[j * 3 + k]; A[i * 3 + j] = t / d; }
    }
    return 0;
}
__global__ __launch_bounds__(64) void tri_kernel(const double* __restrict__ kp7, int n, const double* __restrict__ pose_s, const double* __restrict__ alt_s,
                                                 const double* __restrict__ gr_s, int Ms, const double* __restrict__ pose_t_, const double* __restrict__ alt_t,
                                                 const double* __restrict__ gr_t, int Mt, const double* __restrict__ explicit27, double* __restrict__ out7)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const double PI = DSSS_PI_REF, thr = 2 * PI / 3;
    int flip = 0;                                                   // sticky over the caller's list (optimizer.cpp:650,700-703)
    if (!explicit27) for (int k = 0; k <= i; ++k) {
        const double* q = kp7 + (size_t)k * 7;
        if (fabs(pose_s[(size_t)(int)q[0] * 6 + 2]) > thr) flip |= 1;
        if (fabs(pose_t_[(size_t)(int)q[3] * 6 + 2]) > thr) flip |= 2;
    }
    const double* kp = kp7 + (size_t)i * 7;
    const int id_s = (int)kp[0], id_t = (int)kp[3];
    pose_t cps_s, cps_t, Ps, Pt, Tp_s, Tp_t;
    pose_identity(&cps_s); pose_identity(&cps_t);
    const double flipv[3] = { 0, 0, PI };
    if (flip & 1) so3_exp(flipv, cps_s.R);
    if (flip & 2) so3_exp(flipv, cps_t.R);
    double ini[3];
    if (explicit27) {                                               // TriangulateOneLandmark with the caller's poses and start point
        const double* q = explicit27 + (size_t)i * 27;
        for (int k = 0; k < 9; ++k) { Tp_s.R[k] = q[k]; Tp_t.R[k] = q[12 + k]; }
        for (int k = 0; k < 3; ++k) { Tp_s.t[k] = q[9 + k]; Tp_t.t[k] = q[21 + k]; ini[k] = q[24 + k]; }
    } else {
        pose_from_rodrigues(pose_s + (size_t)id_s * 6, &Ps);
        pose_from_rodrigues(pose_t_ + (size_t)id_t * 6, &Pt);
        pose_compose(&Ps, &cps_s, &Tp_s);
        pose_compose(&Pt, &cps_t, &Tp_t);
        double gsx, gsy, gtx, gty;
        dsss_geo_at(pose_s, gr_s, Ms, id_s, (int)kp[1], &gsx, &gsy);
        dsss_geo_at(pose_t_, gr_t, Mt, id_t, (int)kp[4], &gtx, &gty);
        ini[0] = (gsx + gtx) / 2; ini[1] = (gsy + gty) / 2;
        ini[2] = ((pose_s[(size_t)id_s * 6 + 5] - alt_s[id_s]) + (pose_t_[(size_t)id_t * 6 + 5] - alt_t[id_t])) / 2;
    }
    const double sigma_r = 0.1, alpha_bw = 0.1 * PI / 180;
    const double sig_s[2] = { sigma_r, kp[2] * alpha_bw }, sig_t[2] = { sigma_r, kp[5] * alpha_bw };
    const double dx = Tp_s.t[0] - Tp_t.t[0], dy = Tp_s.t[1] - Tp_t.t[1];
    double sig_p[3] = { 10.0, 10.0, sqrt(dx * dx + dy * dy) / 100 };
    if (sig_p[2] < 1e-9) sig_p[2] = 1e-9;
    double p[3] = { ini[0], ini[1], ini[2] };
    const double relTol = 1e-5, absTol = 1e-5, lamMax = 1e5, minFid = 1e-3;
    double r[7], J[21];
    auto err_of = [&](const double* q) { double rr[7]; tri_lin(Tp_s, Tp_t, kp[2], kp[5], sig_s, sig_t, sig_p, ini, q, rr, nullptr);
                                         double t = 0; for (int k = 0; k < 7; ++k) t += rr[k] * rr[k]; return 0.5 * t; };
    double lambda = 1e-5, err = err_of(p), cur;
    int iters = 0;
    if (err > 0) do {
        cur = err;
        double H[9], g[3];
        tri_lin(Tp_s, Tp_t, kp[2], kp[5], sig_s, sig_t, sig_p, ini, p, r, J);
        for (int a = 0; a < 3; ++a) {
            double t = 0; for (int k = 0; k < 7; ++k) t += J[k * 3 + a] * r[k];
            g[a] = t;
            for (int b = 0; b < 3; ++b) { double u = 0; for (int k = 0; k < 7; ++k) u += J[k * 3 + a] * J[k * 3 + b]; H[a * 3 + b] = u; }
        }
        double oldLin = 0; for (int k = 0; k < 7; ++k) oldLin += r[k] * r[k];
        oldLin *= 0.5;
        for (;;) {
            double A[9], d[3], np_[3] = { 0, 0, 0 };
            for (int a = 0; a < 9; ++a) A[a] = H[a];
            for (int a = 0; a < 3; ++a) { A[a * 3 + a] += lambda; d[a] = -g[a]; }
            const bool ok = tri_chol3(A) == 0;
            bool success = false, stop = false;
            double newErr = 0;
            if (ok) {
                for (int a = 0; a < 3; ++a) { double t = d[a]; for (int k = 0; k < a; ++k) t -= A[a * 3 + k] * d[k]; d[a] = t / A[a * 3 + a]; }
                for (int a = 2; a >= 0; --a) { double t = d[a]; for (int k = a + 1; k < 3; ++k) t -= A[k * 3 + a] * d[k]; d[a] = t / A[a * 3 + a]; }
                double newLin = 0;
                for (int k = 0; k < 7; ++k) { double t = r[k]; for (int a = 0; a < 3; ++a) t += J[k * 3 + a] * d[a]; newLin += t * t; }
                newLin *= 0.5;
                const double linChange = oldLin - newLin;
                if (linChange >= 0) {
                    for (int a = 0; a < 3; ++a) np_[a] = p[a] + d[a];
                    newErr = err_of(np_);
                    const double costChange = err - newErr;
                    if (linChange > 2.220446049250313e-16 * oldLin) success = (costChange / linChange) > minFid;
                    if (fabs(costChange) < relTol * err) stop = true;
                }
            }
            if (success) { for (int a = 0; a < 3; ++a) p[a] = np_[a]; err = newErr; lambda /= 10; ++iters; break; }
            else if (!stop) { lambda *= 10; if (lambda >= lamMax) break; }
            else break;
        }
    } while (iters < 100 && !((err <= 0) || ((cur - err) / cur <= relTol) || ((cur - err) <= absTol)) && isfinite(cur));
    double* o = out7 + (size_t)i * 7;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    double e[2];
    sss_factor(p, &Tp_s, kp[2], 0.0, e, nullptr, nullptr); o[3] = fabs(e[0]); o[4] = fabs(e[1]);
    sss_factor(p, &Tp_t, kp[5], 0.0, e, nullptr, nullptr); o[5] = fabs(e[0]); o[6] = fabs(e[1]);
}

static int ensure_ptr_tables(dsss_ctx* c)
{
    const int F = c->max_frames;
    std::vector<const double*> hp(3 * (size_t)F, nullptr);
    for (int f = 0; f < F; ++f) { hp[f] = c->frames[f].alt; hp[F + f] = c->frames[f].gr; hp[2 * F + f] = c->frames[f].pose6; }
    const size_t need = hp.size() * sizeof(double*) + 2 * (size_t)c->mt.scc_iters * sizeof(uint32_t);
    if (c->mt_aux_bytes < need) {
        HIPCHK(c, hipStreamSynchronize(c->stream)); hipFree(c->mt_aux); c->mt_aux = nullptr; c->mt_aux_bytes = 0;
        HIPCHK(c, hipMalloc(&c->mt_aux, need)); c->mt_aux_bytes = need;
    }
    c->d_ptrs = (const double**)c->mt_aux;
    HIPCHK(c, hipMemcpyAsync((void*)c->d_ptrs, hp.data(), hp.size() * sizeof(double*), hipMemcpyHostToDevice, c->stream));
    return DSSS_OK;
}

extern "C" {

int dsss_lc_solve_all(dsss_ctx* c)
{
    if (!c) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->total_kp7;
    c->has_lc = true; ++c->lc_gen;
    if (n == 0) return DSSS_OK;
    if ((size_t)n > c->lcs_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream)); hipFree(c->lcs); c->lcs = nullptr;
        c->lcs_cap = (size_t)n + 1024;
        HIPCHK(c, hipMalloc(&c->lcs, c->lcs_cap * sizeof(dsss_lc)));
    }
    const int F = c->max_frames;
    dsss_scope sc(c, DSSS_K_LC);
    hipLaunchKernelGGL(lc_kernel, dim3((n + 3) / 4), dim3(64), 0, c->stream, c->kp7, n, c->kp7_pair, c->kp7_flip, c->act_s, c->act_t,
                       0, 0, 0, c->d_ptrs, c->d_ptrs + F, c->d_ptrs + 2 * F, c->cols_dev, c->lcs);
    HIPCHK(c, hipGetLastError());
    return DSSS_OK;
}

int dsss_lc_get(dsss_ctx* c, int pair, dsss_lc* out, int cap, int* nout)
{
    if (!c) return DSSS_E_ARG;
    if (pair < 0 || pair >= c->npairs) DSSS_FAIL(c, DSSS_E_ARG, "pair %d out of range", pair);
    if (!c->has_lc) DSSS_FAIL(c, DSSS_E_STATE, "dsss_lc_solve_all has not run");
    const int a = c->pair_active[pair];
    const int n = a < 0 ? 0 : c->h_kp7_off[a + 1] - c->h_kp7_off[a];
    if (nout) *nout = n;
    if (n == 0 || !out) return DSSS_OK;
    if (cap < n) DSSS_FAIL(c, DSSS_E_CAPACITY, "caller capacity %d < %d", cap, n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->lcs + c->h_kp7_off[a], (size_t)n * sizeof(dsss_lc), hipMemcpyDeviceToHost));
    return DSSS_OK;
}

// LoopClosingTFs for the kp7 lists of MANY pairs given by the caller (what the host mirror's TrajOptimizationAll builds
// with GetKpsPairs from corres_kps or, with USE_ANNO = 1, from anno_kps: optimizer.cpp:35-97): one upload, one launch.
// The results stay on the device exactly as after dsss_match_pairs + dsss_lc_solve_all, so dsss_lc_get,
// dsss_posegraph_select and dsss_posegraph_solve work on them (pair order = the caller's order = the reference's loop order).
int dsss_lc_solve_pairs(dsss_ctx* c, const int* src_ids, const int* tgt_ids, int npairs, const double* kp7, const int* pair_off)
{
    if (!c || npairs < 0 || (npairs > 0 && (!src_ids || !tgt_ids || !pair_off))) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = npairs > 0 ? pair_off[npairs] : 0;
    if (n < 0 || (n > 0 && !kp7)) return DSSS_E_ARG;
    for (int p = 0; p < npairs; ++p) {
        const int s = src_ids[p], t = tgt_ids[p];
        if (s < 0 || s >= c->max_frames || t < 0 || t >= c->max_frames || s == t) DSSS_FAIL(c, DSSS_E_ARG, "pair %d: bad frame ids (%d,%d)", p, s, t);
        if (!c->frames[s].has_geom || !c->frames[t].has_geom) DSSS_FAIL(c, DSSS_E_STATE, "pair %d: frames need dsss_frame_set first", p);
        if (pair_off[p + 1] < pair_off[p]) DSSS_FAIL(c, DSSS_E_ARG, "pair_off is not ascending at pair %d", p);
    }
    int rc = dsss_sync_bboxes(c); if (rc) return rc;
    rc = ensure_ptr_tables(c); if (rc) return rc;
    // every listed pair is "active" here: the bookkeeping below is what dsss_match_pairs leaves behind
    c->npairs = npairs; c->nactive = npairs; c->has_lc = false;
    c->pair_s.assign(src_ids, src_ids + npairs); c->pair_t.assign(tgt_ids, tgt_ids + npairs);
    c->pair_active.resize(npairs);
    for (int p = 0; p < npairs; ++p) c->pair_active[p] = p;
    c->h_kp7_off.assign(pair_off, pair_off + npairs + 1);
    c->h_row_off.assign(npairs + 1, 0);
    c->total_rows = 0; c->total_kp7 = n;
    if (npairs == 0 || n == 0) { c->has_lc = true; ++c->lc_gen; return DSSS_OK; }
    std::vector<double> h((size_t)n * 7);
    HIPCHK(c, hipMemcpy(h.data(), kp7, h.size() * sizeof(double), hipMemcpyDefault));
    std::vector<int> h_pair(n); std::vector<uint8_t> h_flip(n);
    const double thr = 2 * DSSS_PI_REF / 3;                         // optimizer.cpp:697-703: sticky within one LoopClosingTFs call
    for (int p = 0; p < npairs; ++p) {
        const dsss_frame &fs = c->frames[src_ids[p]], &ft = c->frames[tgt_ids[p]];
        if (!fs.h_geo || !ft.h_geo) DSSS_FAIL(c, DSSS_E_STATE, "pair %d: host copy of the DR poses missing", p);
        uint8_t flip = 0;
        for (int i = pair_off[p]; i < pair_off[p + 1]; ++i) {
            const double* k = h.data() + (size_t)i * 7;
            const bool ok = k[0] >= 0 && k[0] < fs.N && k[3] >= 0 && k[3] < ft.N && k[1] >= 1 && k[1] < fs.M && k[4] >= 1 && k[4] < ft.M;
            if (!ok) DSSS_FAIL(c, DSSS_E_ARG, "pair %d kp7 row %d: ping/bin outside the frames", p, i - pair_off[p]);
            if (std::fabs(fs.h_geo[(size_t)(int)k[0] * 6 + 2]) > thr) flip |= 1;
            if (std::fabs(ft.h_geo[(size_t)(int)k[3] * 6 + 2]) > thr) flip |= 2;
            h_pair[i] = p; h_flip[i] = flip;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((size_t)npairs > c->match_cap_pairs) {
        c->match_cap_pairs = 0;
        hipFree(c->act_s); c->act_s = nullptr; hipFree(c->act_t); c->act_t = nullptr; hipFree(c->kp7_off); c->kp7_off = nullptr;
        hipFree(c->corres_nn); c->corres_nn = nullptr; hipFree(c->corres); c->corres = nullptr;
        hipFree(c->scc_hist); c->scc_hist = nullptr; hipFree(c->scc_count); c->scc_count = nullptr; hipFree(c->scc_model); c->scc_model = nullptr;
        hipFree(c->row_cnt); c->row_cnt = nullptr; hipFree(c->kp7_cnt); c->kp7_cnt = nullptr; hipFree(c->row_off); c->row_off = nullptr;
        HIPCHK(c, hipMalloc(&c->act_s, npairs * sizeof(int))); HIPCHK(c, hipMalloc(&c->act_t, npairs * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->kp7_off, (npairs + 1) * sizeof(int)));
        // (the matcher's own per-pair buffers are re-created by the next dsss_match_pairs: capacity stays 0)
    }
    if ((size_t)n > c->rows_cap) {
        c->rows_cap = 0;
        hipFree(c->rows6); c->rows6 = nullptr; hipFree(c->kp7); c->kp7 = nullptr; hipFree(c->kp7_pair); c->kp7_pair = nullptr; hipFree(c->kp7_flip); c->kp7_flip = nullptr;
        const size_t want = (size_t)n + 1024;
        HIPCHK(c, hipMalloc(&c->rows6, want * 6 * sizeof(double))); HIPCHK(c, hipMalloc(&c->kp7, want * 7 * sizeof(double)));
        HIPCHK(c, hipMalloc(&c->kp7_pair, want * sizeof(int))); HIPCHK(c, hipMalloc(&c->kp7_flip, want));
        c->rows_cap = want;
    }
    HIPCHK(c, hipMemcpy(c->act_s, src_ids, npairs * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->act_t, tgt_ids, npairs * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->kp7_off, pair_off, (npairs + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->kp7, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->kp7_pair, h_pair.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->kp7_flip, h_flip.data(), (size_t)n, hipMemcpyHostToDevice));
    return dsss_lc_solve_all(c);
}

int dsss_lc_solve(dsss_ctx* c, int id_s, int id_t, const double* kp7, int n, dsss_lc* out)
{
    if (!c || n < 0 || (n > 0 && (!kp7 || !out))) return DSSS_E_ARG;
    if (id_s < 0 || id_s >= c->max_frames || id_t < 0 || id_t >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id out of range");
    if (!c->frames[id_s].has_geom || !c->frames[id_t].has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frames need dsss_frame_set first");
    if (n == 0) return DSSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    {   // caller-supplied kp7: pings and bins index altitude / ground-range tables on the device, so range-check them here
        // (GetKpsPairs never emits |bin - M/2| < 20, optimizer.cpp:602-609; bin - M/2 == M/2 is the one-past-the-end read)
        std::vector<double> h((size_t)n * 7);
        HIPCHK(c, hipMemcpy(h.data(), kp7, h.size() * sizeof(double), hipMemcpyDefault));
        const dsss_frame &fs = c->frames[id_s], &ft = c->frames[id_t];
        for (int i = 0; i < n; ++i) {
            const double* k = h.data() + (size_t)i * 7;
            const bool ok = k[0] >= 0 && k[0] < fs.N && k[3] >= 0 && k[3] < ft.N && k[1] >= 1 && k[1] < fs.M && k[4] >= 1 && k[4] < ft.M;
            if (!ok) DSSS_FAIL(c, DSSS_E_ARG, "kp7 row %d: ping/bin outside the frames (%g,%g | %g,%g)", i, k[0], k[1], k[3], k[4]);
        }
    }
    int rc = dsss_sync_bboxes(c); if (rc) return rc;          // also publishes the frames' N and M to the device tables
    rc = ensure_ptr_tables(c); if (rc) return rc;
    double* d_kp7 = nullptr; dsss_lc* d_out = nullptr;
    HIPCHK(c, hipMalloc(&d_kp7, (size_t)n * 7 * sizeof(double)));
    HIPCHK(c, hipMalloc(&d_out, (size_t)n * sizeof(dsss_lc)));
    HIPCHK(c, hipMemcpyAsync(d_kp7, kp7, (size_t)n * 7 * sizeof(double), hipMemcpyDefault, c->stream));
    const int F = c->max_frames;
    {
        dsss_scope sc(c, DSSS_K_LC);
        hipLaunchKernelGGL(lc_kernel, dim3((n + 3) / 4), dim3(64), 0, c->stream, d_kp7, n, (const int*)nullptr, (const uint8_t*)nullptr,
                           (const int*)nullptr, (const int*)nullptr, id_s, id_t, 0, c->d_ptrs, c->d_ptrs + F, c->d_ptrs + 2 * F, c->cols_dev, d_out);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n * sizeof(dsss_lc), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_kp7); hipFree(d_out);
    HIPCHK(c, e);
    return DSSS_OK;
}

// Optimizer::TriangulateOneLandmark for every row of the caller's kp7 list of one pair (optimizer.h:56-59, call site
// optimizer.cpp:907-921).  out7_host: n x 7 = [x y z | |range_s err| |plane_s| |range_t err| |plane_t|].
int dsss_triangulate(dsss_ctx* c, int id_s, int id_t, const double* kp7, int n, double* out7)
{
    if (!c || n < 0 || (n > 0 && (!kp7 || !out7))) return DSSS_E_ARG;
    if (id_s < 0 || id_s >= c->max_frames || id_t < 0 || id_t >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id out of range");
    const dsss_frame &fs = c->frames[id_s], &ft = c->frames[id_t];
    if (!fs.has_geom || !ft.has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frames need dsss_frame_set first");
    if (n == 0) return DSSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<double> h((size_t)n * 7);
    HIPCHK(c, hipMemcpy(h.data(), kp7, h.size() * sizeof(double), hipMemcpyDefault));
    for (int i = 0; i < n; ++i) {
        const double* k = h.data() + (size_t)i * 7;
        const bool ok = k[0] >= 0 && k[0] < fs.N && k[3] >= 0 && k[3] < ft.N && k[1] >= 1 && k[1] < fs.M && k[4] >= 1 && k[4] < ft.M;
        if (!ok) DSSS_FAIL(c, DSSS_E_ARG, "kp7 row %d: ping/bin outside the frames", i);
    }
    double* d_kp7 = nullptr; double* d_out = nullptr;
    HIPCHK(c, hipMalloc(&d_kp7, h.size() * sizeof(double)));
    hipError_t e = hipMalloc(&d_out, h.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_kp7, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(tri_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, d_kp7, n, fs.pose6, fs.alt, fs.gr, fs.M, ft.pose6, ft.alt, ft.gr, ft.M, (const double*)nullptr, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out7, d_out, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_kp7); hipFree(d_out);
    HIPCHK(c, e);
    return DSSS_OK;
}

// the same with the caller's own poses: Optimizer::TriangulateOneLandmark(kps_pair, Ts_s, Ts_t, Tp_s, Tp_t, lm_ini) with
// Ts = identity (frame.cpp:38-39).  in27: n x [Tp_s R(9) t(3) | Tp_t R(9) t(3) | lm_ini(3)]; only kp7[2], kp7[5] (slant ranges) are read.
int dsss_triangulate_poses(dsss_ctx* c, const double* kp7, const double* in27, int n, double* out7)
{
    if (!c || n < 0 || (n > 0 && (!kp7 || !in27 || !out7))) return DSSS_E_ARG;
    if (n == 0) return DSSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    double* d = nullptr;
    HIPCHK(c, hipMalloc(&d, (size_t)n * (7 + 27 + 7) * sizeof(double)));
    double* d_kp7 = d; double* d_in = d + (size_t)n * 7; double* d_out = d_in + (size_t)n * 27;
    hipError_t e = hipMemcpyAsync(d_kp7, kp7, (size_t)n * 7 * sizeof(double), hipMemcpyDefault, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in27, (size_t)n * 27 * sizeof(double), hipMemcpyDefault, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(tri_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, d_kp7, n, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, 0,
                           (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, 0, (const double*)d_in, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out7, d_out, (size_t)n * 7 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    HIPCHK(c, e);
    return DSSS_OK;
}

} // extern "C"
