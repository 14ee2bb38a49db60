// diasss_amd/csrc/dsss_lc.hip -- batched loop-closure measurements: one 15-DoF Levenberg-Marquardt problem per
// matched keypoint pair.  Restates Optimizer::LoopClosingTFs, graph_option = 0
// (/root/reference/src/core/optimizer.cpp:641-982): graph {Prior(X1, 1e-6), Between(X1, X2; DR odometry, adaptive
// sigmas :778), SssPoint(L1, X1), SssPoint(L1, X2)} (:773-786), GTSAM LM with default parameters (:815-822), marginal
// covariance of X2 (:956-959), score = ini/final - 2 (:853-896).  GTSAM semantics per SURVEY.md A.2/A.3 and
// oracle/orc_lc.c.  f64 VALU: 15 x 15 systems are far too small for MFMA; one thread per problem.
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

__device__ static void mini_lin(const mini_prob& m, const mini_val& v, double* r, double* J)
{
    if (J) for (int i = 0; i < MR * MD; ++i) J[i] = 0.0;
    pose_t d; double xi[6];
    pose_between(&m.prior, &v.X1, &d);                     // PriorFactor: e = Logmap(prior^-1 x), H = I
    pose_log(&d, xi);
    for (int i = 0; i < 6; ++i) { r[i] = xi[i] / m.sig_prior[i]; if (J) J[i * MD + 3 + i] = 1.0 / m.sig_prior[i]; }
    pose_t h, e;
    pose_between(&v.X1, &v.X2, &h);                        // BetweenFactor: e = Logmap(meas^-1 h), H1 = -Ad(h^-1), H2 = I
    pose_between(&m.odo, &h, &e);
    pose_log(&e, xi);
    for (int i = 0; i < 6; ++i) r[6 + i] = xi[i] / m.sig_odo[i];
    if (J) {
        pose_t hi; double Ad[36];
        pose_inverse(&h, &hi);
        pose_adjoint(&hi, Ad);
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) J[(6 + i) * MD + 3 + j] = -Ad[6 * i + j] / m.sig_odo[i];
            J[(6 + i) * MD + 9 + i] = 1.0 / m.sig_odo[i];
        }
    }
    double ee[2], H1[6], H2[12];
    sss_factor(v.L, &v.X1, m.slant_s, 0.0, ee, J ? H1 : nullptr, H2);
    for (int i = 0; i < 2; ++i) {
        r[12 + i] = ee[i] / m.sig_s[i];
        if (J) {
            for (int j = 0; j < 3; ++j) J[(12 + i) * MD + j] = H1[3 * i + j] / m.sig_s[i];
            for (int j = 0; j < 6; ++j) J[(12 + i) * MD + 3 + j] = H2[6 * i + j] / m.sig_s[i];
        }
    }
    sss_factor(v.L, &v.X2, m.slant_t, 0.0, ee, J ? H1 : nullptr, H2);
    for (int i = 0; i < 2; ++i) {
        r[14 + i] = ee[i] / m.sig_t[i];
        if (J) {
            for (int j = 0; j < 3; ++j) J[(14 + i) * MD + j] = H1[3 * i + j] / m.sig_t[i];
            for (int j = 0; j < 6; ++j) J[(14 + i) * MD + 9 + j] = H2[6 * i + j] / m.sig_t[i];
        }
    }
}
__device__ static double mini_err(const mini_prob& m, const mini_val& v)
{
    double r[MR];
    mini_lin(m, v, r, nullptr);
    double s = 0;
    for (int i = 0; i < MR; ++i) s += r[i] * r[i];
    return 0.5 * s;
}
__device__ static int chol15(double* A)
{
    for (int j = 0; j < MD; ++j) {
        double d = A[j * MD + j];
        for (int k = 0; k < j; ++k) d -= A[j * MD + k] * A[j * MD + k];
        if (!(d > 0) || !isfinite(d)) return -1;
        d = sqrt(d); A[j * MD + j] = d;
        for (int i = j + 1; i < MD; ++i) {
            double s = A[i * MD + j];
            for (int k = 0; k < j; ++k) s -= A[i * MD + k] * A[j * MD + k];
            A[i * MD + j] = s / d;
        }
    }
    return 0;
}
__device__ static void chol15_solve(const double* L, double* b)
{
    for (int i = 0; i < MD; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * MD + k] * b[k]; b[i] = s / L[i * MD + i]; }
    for (int i = MD - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < MD; ++k) s -= L[k * MD + i] * b[k]; b[i] = s / L[i * MD + i]; }
}
__device__ static void normal_eq(const double* J, const double* r, double* H, double* g)
{
    for (int a = 0; a < MD; ++a) {
        double s = 0;
        for (int k = 0; k < MR; ++k) s += J[k * MD + a] * r[k];
        if (g) g[a] = s;
        for (int b = 0; b <= a; ++b) {
            double t = 0;
            for (int k = 0; k < MR; ++k) t += J[k * MD + a] * J[k * MD + b];
            H[a * MD + b] = t; H[b * MD + a] = t;
        }
    }
}

// kp7: n x 7; per problem: frame pointers (pose6 / alt / gr of source and target), M of both, flip flags
__global__ __launch_bounds__(64) void lc_kernel(const double* __restrict__ kp7, int n,
                                                const int* __restrict__ kp7_pair, const uint8_t* __restrict__ kp7_flip,
                                                const int* __restrict__ act_s, const int* __restrict__ act_t,
                                                int single_s, int single_t, int single_flip,
                                                const double* const* __restrict__ alt_ptr, const double* const* __restrict__ gr_ptr,
                                                const double* const* __restrict__ pose_ptr, const int* __restrict__ fcols,
                                                dsss_lc* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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
    mini_prob m;
    m.slant_s = kp[2]; m.slant_t = kp[5];
    m.sig_s[0] = sigma_r; m.sig_s[1] = kp[2] * alpha_bw;
    m.sig_t[0] = sigma_r; m.sig_t[1] = kp[5] * alpha_bw;
    pose_t Ps, Pt, Tp_s, Tp_t, Tp_st;
    pose_from_rodrigues(pose_s + (size_t)id_s * 6, &Ps);
    pose_from_rodrigues(pose_t_ + (size_t)id_t * 6, &Pt);
    pose_compose(&Ps, &cps_s, &Tp_s);
    pose_compose(&Pt, &cps_t, &Tp_t);
    pose_between(&Tp_s, &Tp_t, &Tp_st);
    for (int k = 0; k < 6; ++k) m.sig_prior[k] = 0.000001;
    m.sig_odo[0] = 0.1 * PI / 180; m.sig_odo[1] = 0.1 * PI / 180; m.sig_odo[2] = 0.5 * PI / 180;          // :778
    m.sig_odo[3] = fabs(Tp_st.t[0] * 2); m.sig_odo[4] = fabs(Tp_st.t[1] / 10); m.sig_odo[5] = 0.1;
    for (int k = 3; k < 5; ++k) if (m.sig_odo[k] < 1e-9) m.sig_odo[k] = 1e-9;
    m.prior = Tp_s; m.odo = Tp_st;
    const int Ms = fcols[fs], Mt = fcols[ft];
    const int id_ss = (int)kp[1], id_tt = (int)kp[4];
    double gsx, gsy, gtx, gty;
    dsss_geo_at(pose_s, gr_ptr[fs], Ms, id_s, id_ss, &gsx, &gsy);
    dsss_geo_at(pose_t_, gr_ptr[ft], Mt, id_t, id_tt, &gtx, &gty);
    mini_val v;
    v.L[0] = (gsx + gtx) / 2; v.L[1] = (gsy + gty) / 2;                                                    // :792-795
    v.L[2] = ((pose_s[(size_t)id_s * 6 + 5] - alt_ptr[fs][id_s]) + (pose_t_[(size_t)id_t * 6 + 5] - alt_ptr[ft][id_t])) / 2;
    v.X1 = Tp_s; v.X2 = Tp_t;
    // ---- LevenbergMarquardtOptimizer::optimize, default params
    const double relTol = 1e-5, absTol = 1e-5, lamMax = 1e5, minFid = 1e-3;
    double lambda = 1e-5;
    int iters = 0;
    double err = mini_err(m, v);
    const double err0 = err;
    double r[MR], J[MR * MD], H[MD * MD], g[MD], A[MD * MD], d[MD];
    if (err > 0) {
        double cur;
        do {
            cur = err;
            mini_lin(m, v, r, J);
            normal_eq(J, r, H, g);
            double oldLin = 0;
            for (int k = 0; k < MR; ++k) oldLin += r[k] * r[k];
            oldLin *= 0.5;
            for (;;) {
                for (int a = 0; a < MD * MD; ++a) A[a] = H[a];
                for (int a = 0; a < MD; ++a) { A[a * MD + a] += lambda; d[a] = -g[a]; }
                const bool ok = chol15(A) == 0;
                bool success = false, stop = false;
                double newErr = 0; mini_val nv;
                if (ok) {
                    chol15_solve(A, d);
                    double newLin = 0;
                    for (int k = 0; k < MR; ++k) {
                        double s = r[k];
                        for (int a = 0; a < MD; ++a) s += J[k * MD + a] * d[a];
                        newLin += s * s;
                    }
                    newLin *= 0.5;
                    const double linChange = oldLin - newLin;
                    if (linChange >= 0) {
                        for (int a = 0; a < 3; ++a) nv.L[a] = v.L[a] + d[a];
                        pose_retract(&v.X1, d + 3, &nv.X1);
                        pose_retract(&v.X2, d + 9, &nv.X2);
                        newErr = mini_err(m, nv);
                        const double costChange = err - newErr;
                        if (linChange > 2.220446049250313e-16 * oldLin) success = (costChange / linChange) > minFid;
                        if (fabs(costChange) < relTol * err) stop = true;
                    }
                }
                if (success) { v = nv; err = newErr; lambda /= 10; ++iters; break; }
                else if (!stop) { lambda *= 10; if (lambda >= lamMax) break; }
                else break;
            }
        } while (iters < 100 && !((err <= 0) || ((cur - err) / cur <= relTol) || ((cur - err) <= absTol)) && isfinite(cur));
    }
    dsss_lc o;
    o.iters = iters; o.pad_ = 0; o.err0 = err0; o.err1 = err;
    // eval_1 (:853-896)
    pose_t cti, new_pose;
    pose_inverse(&cps_t, &cti);
    pose_compose(&v.X2, &cti, &new_pose);
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
    // Marginals(graph, result).marginalCovariance(X2).diagonal() (:956-959)
    mini_lin(m, v, r, J);
    normal_eq(J, r, H, nullptr);
    if (chol15(H) == 0) {
        for (int c = 0; c < 6; ++c) {
            for (int a = 0; a < MD; ++a) d[a] = 0;
            d[9 + c] = 1;
            chol15_solve(H, d);
            o.var[c] = d[9 + c];
        }
    } else for (int c = 0; c < 6; ++c) o.var[c] = NAN;
    pose_t csi, src, rel;
    pose_inverse(&cps_s, &csi);
    pose_compose(&Tp_s, &csi, &src);
    pose_between(&src, &new_pose, &rel);                         // :958
    for (int a = 0; a < 9; ++a) o.rel[a] = rel.R[a];
    for (int a = 0; a < 3; ++a) o.rel[9 + a] = rel.t[a];
    out[i] = o;
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
    c->has_lc = true;
    if (n == 0) return DSSS_OK;
    if ((size_t)n > c->lcs_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream)); hipFree(c->lcs); c->lcs = nullptr;
        c->lcs_cap = (size_t)n + 1024;
        HIPCHK(c, hipMalloc(&c->lcs, c->lcs_cap * sizeof(dsss_lc)));
    }
    const int F = c->max_frames;
    dsss_scope sc(c, DSSS_K_LC);
    hipLaunchKernelGGL(lc_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->kp7, n, c->kp7_pair, c->kp7_flip, c->act_s, c->act_t,
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

int dsss_lc_solve(dsss_ctx* c, int id_s, int id_t, const double* kp7, int n, dsss_lc* out)
{
    if (!c || n < 0 || (n > 0 && (!kp7 || !out))) return DSSS_E_ARG;
    if (id_s < 0 || id_s >= c->max_frames || id_t < 0 || id_t >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id out of range");
    if (!c->frames[id_s].has_geom || !c->frames[id_t].has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frames need dsss_frame_set first");
    if (n == 0) return DSSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_ptr_tables(c); if (rc) return rc;
    double* d_kp7 = nullptr; dsss_lc* d_out = nullptr;
    HIPCHK(c, hipMalloc(&d_kp7, (size_t)n * 7 * sizeof(double)));
    HIPCHK(c, hipMalloc(&d_out, (size_t)n * sizeof(dsss_lc)));
    HIPCHK(c, hipMemcpyAsync(d_kp7, kp7, (size_t)n * 7 * sizeof(double), hipMemcpyDefault, c->stream));
    const int F = c->max_frames;
    {
        dsss_scope sc(c, DSSS_K_LC);
        hipLaunchKernelGGL(lc_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, d_kp7, n, (const int*)nullptr, (const uint8_t*)nullptr,
                           (const int*)nullptr, (const int*)nullptr, id_s, id_t, 0, c->d_ptrs, c->d_ptrs + F, c->d_ptrs + 2 * F, c->cols_dev, d_out);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n * sizeof(dsss_lc), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_kp7); hipFree(d_out);
    HIPCHK(c, e);
    return DSSS_OK;
}

} // extern "C"
