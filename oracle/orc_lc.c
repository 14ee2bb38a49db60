/* oracle/orc_lc.c -- sonar reprojection (GetKpsPairs) and the per-match 15-DoF mini-LM that manufactures
 * loop-closure measurements, restated from /root/reference/src/core/optimizer.cpp:575-982.
 * GTSAM 4.2 (not in tree) semantics follow SURVEY.md Appendix A.2/A.3:
 *   PriorFactor<Pose3>: e = -Local(x, prior), H = I
 *   BetweenFactor<Pose3> (default build, GTSAM_SLOW_BUT_CORRECT_BETWEENFACTOR off):
 *       h = x1^-1 x2, e = Logmap(measured^-1 h), H1 = -Ad(h^-1), H2 = I
 *   LevenbergMarquardtParams(): lambda 1e-5, factor 10, upper 1e5, lower 0, minModelFidelity 1e-3,
 *       diagonalDamping off, maxIterations 100, relTol 1e-5, absTol 1e-5, errorTol 0.
 * Test infrastructure, see orc.h. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* GetKpsPairs, USE_ANNO = 0 branch (optimizer.cpp:575-639) */
int orc_get_kps_pairs(const double* rows6, int nrows, int id_t, const double* alt_s, const double* gr_s, int ngr_s,
                      const double* alt_t, const double* gr_t, int ngr_t, double* kp7, int cap)
{
    int n = 0;
    for (int i = 0; i < nrows; ++i) {
        const double* r = rows6 + (size_t)i * 6;
        int id_check = (int)r[1];
        int kp_s[2] = { (int)r[2], (int)r[3] }, kp_t[2] = { (int)r[4], (int)r[5] };
        const int nd_thres = 20;
        int ds = kp_s[1] - ngr_s, dt = kp_t[1] - ngr_t;
        if (abs(ds) < nd_thres || abs(dt) < nd_thres) continue;
        if (id_check != id_t) continue;
        double a_s = alt_s[kp_s[0]], g_s = gr_s[abs(ds)];
        double a_t = alt_t[kp_t[0]], g_t = gr_t[abs(dt)];
        if (n >= cap) break;
        double* o = kp7 + (size_t)n * 7;
        o[0] = kp_s[0]; o[1] = kp_s[1]; o[2] = sqrt(a_s * a_s + g_s * g_s);
        o[3] = kp_t[0]; o[4] = kp_t[1]; o[5] = sqrt(a_t * a_t + g_t * g_t);
        o[6] = 0;
        ++n;
    }
    return n;
}

/* in-place lower Cholesky of a dense n x n SPD matrix; returns 0 on success */
static int chol(double* A, int n)
{
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0) || !isfinite(d)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(const double* L, int n, double* b)
{
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k]; b[i] = s / L[i * n + i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k]; b[i] = s / L[i * n + i]; }
}

#define MROWS 16
#define MDIM 15
typedef struct {
    orc_pose prior, odo, Ts_s, Ts_t;
    double sig_prior[6], sig_odo[6], sig_s[2], sig_t[2];
    double slant_s, slant_t;
} mini_t;
typedef struct { double L[3]; orc_pose X1, X2; } mini_vals;

/* whitened residual (16) and Jacobian (16 x 15, row-major) of the 4-factor graph (optimizer.cpp:773-786) */
static void mini_lin(const mini_t* m, const mini_vals* v, double* r, double* J)
{
    if (J) memset(J, 0, sizeof(double) * MROWS * MDIM);
    orc_pose d;
    double xi[6];
    /* PriorFactor(X1) */
    orc_pose_between(&m->prior, &v->X1, &d);
    orc_pose_log(&d, xi);
    for (int i = 0; i < 6; ++i) { r[i] = xi[i] / m->sig_prior[i]; if (J) J[i * MDIM + 3 + i] = 1.0 / m->sig_prior[i]; }
    /* BetweenFactor(X1, X2) */
    orc_pose h, hi, e;
    orc_pose_between(&v->X1, &v->X2, &h);
    orc_pose_between(&m->odo, &h, &e);
    orc_pose_log(&e, xi);
    for (int i = 0; i < 6; ++i) r[6 + i] = xi[i] / m->sig_odo[i];
    if (J) {
        double Ad[36];
        orc_pose_inverse(&h, &hi);
        orc_pose_adjoint(&hi, Ad);
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) J[(6 + i) * MDIM + 3 + j] = -Ad[6 * i + j] / m->sig_odo[i];
            J[(6 + i) * MDIM + 9 + i] = 1.0 / m->sig_odo[i];
        }
    }
    /* SssPointFactor(L, X1), SssPointFactor(L, X2) */
    double ee[2], H1[6], H2[12];
    orc_sss_factor(v->L, &v->X1, &m->Ts_s, m->slant_s, 0.0, ee, H1, H2);
    for (int i = 0; i < 2; ++i) {
        r[12 + i] = ee[i] / m->sig_s[i];
        if (J) {
            for (int j = 0; j < 3; ++j) J[(12 + i) * MDIM + j] = H1[3 * i + j] / m->sig_s[i];
            for (int j = 0; j < 6; ++j) J[(12 + i) * MDIM + 3 + j] = H2[6 * i + j] / m->sig_s[i];
        }
    }
    orc_sss_factor(v->L, &v->X2, &m->Ts_t, m->slant_t, 0.0, ee, H1, H2);
    for (int i = 0; i < 2; ++i) {
        r[14 + i] = ee[i] / m->sig_t[i];
        if (J) {
            for (int j = 0; j < 3; ++j) J[(14 + i) * MDIM + j] = H1[3 * i + j] / m->sig_t[i];
            for (int j = 0; j < 6; ++j) J[(14 + i) * MDIM + 9 + j] = H2[6 * i + j] / m->sig_t[i];
        }
    }
}
static double mini_err(const mini_t* m, const mini_vals* v)
{
    double r[MROWS];
    mini_lin(m, v, r, NULL);
    double s = 0;
    for (int i = 0; i < MROWS; ++i) s += r[i] * r[i];
    return 0.5 * s;
}
static void mini_retract(const mini_vals* v, const double* d, mini_vals* o)
{
    for (int i = 0; i < 3; ++i) o->L[i] = v->L[i] + d[i];
    orc_pose_retract(&v->X1, d + 3, &o->X1);
    orc_pose_retract(&v->X2, d + 9, &o->X2);
}

/* GTSAM LevenbergMarquardtOptimizer::optimize with default params */
static int mini_lm(const mini_t* m, mini_vals* v, double* err0, double* err1)
{
    const double relTol = 1e-5, absTol = 1e-5, lamMax = 1e5, minFid = 1e-3;
    const int maxIter = 100;
    double lambda = 1e-5;
    int iters = 0;
    double err = mini_err(m, v);
    *err0 = err;
    if (err <= 0) { *err1 = err; return 0; }
    double cur;
    do {
        cur = err;
        double r[MROWS], J[MROWS * MDIM], H[MDIM * MDIM], g[MDIM];
        mini_lin(m, v, r, J);
        for (int a = 0; a < MDIM; ++a) {
            double s = 0;
            for (int k = 0; k < MROWS; ++k) s += J[k * MDIM + a] * r[k];
            g[a] = s;
            for (int b = 0; b < MDIM; ++b) {
                double t = 0;
                for (int k = 0; k < MROWS; ++k) t += J[k * MDIM + a] * J[k * MDIM + b];
                H[a * MDIM + b] = t;
            }
        }
        double oldLin = 0;
        for (int k = 0; k < MROWS; ++k) oldLin += r[k] * r[k];
        oldLin *= 0.5;
        for (;;) {
            double A[MDIM * MDIM], d[MDIM];
            memcpy(A, H, sizeof A);
            for (int a = 0; a < MDIM; ++a) { A[a * MDIM + a] += lambda; d[a] = -g[a]; }
            int ok = chol(A, MDIM) == 0;
            int success = 0, stop = 0;
            double newErr = 0; mini_vals nv;
            if (ok) {
                chol_solve(A, MDIM, d);
                double newLin = 0;
                for (int k = 0; k < MROWS; ++k) {
                    double s = r[k];
                    for (int a = 0; a < MDIM; ++a) s += J[k * MDIM + a] * d[a];
                    newLin += s * s;
                }
                newLin *= 0.5;
                double linChange = oldLin - newLin;
                if (linChange >= 0) {
                    mini_retract(v, d, &nv);
                    newErr = mini_err(m, &nv);
                    double costChange = err - newErr;
                    if (linChange > 2.220446049250313e-16 * oldLin) {
                        double fid = costChange / linChange;
                        success = fid > minFid;
                    }
                    if (fabs(costChange) < relTol * err) stop = 1;
                }
            }
            if (success) { *v = nv; err = newErr; lambda /= 10; ++iters; break; }
            else if (!stop) { lambda *= 10; if (lambda >= lamMax) break; }
            else break;
        }
    } while (iters < maxIter && !((err <= 0) || ((cur - err) / cur <= relTol) || ((cur - err) <= absTol)) && isfinite(cur));
    *err1 = err;
    return iters;
}

/* LoopClosingTFs (optimizer.cpp:641-982), graph_option = 0 */
int orc_lc_solve(const double* kp7, int n, const double* pose6_s, const double* alt_s, const double* gr_s, int Ns, int Ms,
                 const double* pose6_t, const double* alt_t, const double* gr_t, int Nt, int Mt, orc_lc* out)
{
    const double PI = ORC_PI_REF;
    const double sigma_r = 0.1, alpha_bw = 0.1 * PI / 180;
    orc_pose cps_s, cps_t, ident;
    memset(&ident, 0, sizeof ident); ident.R[0] = ident.R[4] = ident.R[8] = 1;
    cps_s = ident; cps_t = ident;                       /* declared outside the loop: sticky (:650,:700-703) */
    const double flipv[3] = { 0, 0, PI };
    for (int i = 0; i < n; ++i) {
        const double* kp = kp7 + (size_t)i * 7;
        int id_s = (int)kp[0], id_t = (int)kp[3];
        double yaw_s = pose6_s[(size_t)id_s * 6 + 2], yaw_t = pose6_t[(size_t)id_t * 6 + 2];
        if (fabs(yaw_s) > 2 * PI / 3) { orc_so3_exp(flipv, cps_s.R); cps_s.t[0] = cps_s.t[1] = cps_s.t[2] = 0; }
        if (fabs(yaw_t) > 2 * PI / 3) { orc_so3_exp(flipv, cps_t.R); cps_t.t[0] = cps_t.t[1] = cps_t.t[2] = 0; }
        mini_t m;
        m.slant_s = kp[2]; m.slant_t = kp[5];
        m.sig_s[0] = sigma_r; m.sig_s[1] = kp[2] * alpha_bw;
        m.sig_t[0] = sigma_r; m.sig_t[1] = kp[5] * alpha_bw;
        m.Ts_s = ident; m.Ts_t = ident;               /* tf_stb = tf_port = 0 (frame.cpp:38-39) */
        orc_pose Ps, Pt, Tp_s, Tp_t, Tp_st;
        orc_pose_from_rodrigues(pose6_s + (size_t)id_s * 6, &Ps);
        orc_pose_from_rodrigues(pose6_t + (size_t)id_t * 6, &Pt);
        orc_pose_compose(&Ps, &cps_s, &Tp_s);
        orc_pose_compose(&Pt, &cps_t, &Tp_t);
        orc_pose_between(&Tp_s, &Tp_t, &Tp_st);
        for (int k = 0; k < 6; ++k) m.sig_prior[k] = 0.000001;
        m.sig_odo[0] = 0.1 * PI / 180; m.sig_odo[1] = 0.1 * PI / 180; m.sig_odo[2] = 0.5 * PI / 180;
        m.sig_odo[3] = fabs(Tp_st.t[0] * 2); m.sig_odo[4] = fabs(Tp_st.t[1] / 10); m.sig_odo[5] = 0.1;
        for (int k = 3; k < 5; ++k) if (m.sig_odo[k] < 1e-9) m.sig_odo[k] = 1e-9;   /* sigma 0 would be a constrained model */
        m.prior = Tp_s; m.odo = Tp_st;
        int id_ss = (int)kp[1], id_tt = (int)kp[4];
        double gsx, gsy, gtx, gty;
        orc_geo_at(pose6_s, gr_s, Ns, Ms, id_s, id_ss, &gsx, &gsy);
        orc_geo_at(pose6_t, gr_t, Nt, Mt, id_t, id_tt, &gtx, &gty);
        mini_vals v;
        v.L[0] = (gsx + gtx) / 2; v.L[1] = (gsy + gty) / 2;
        v.L[2] = ((pose6_s[(size_t)id_s * 6 + 5] - alt_s[id_s]) + (pose6_t[(size_t)id_t * 6 + 5] - alt_t[id_t])) / 2;
        v.X1 = Tp_s; v.X2 = Tp_t;
        orc_lc* o = &out[i];
        o->iters = mini_lm(&m, &v, &o->err0, &o->err1);
        /* eval_1 (:853-896) */
        orc_pose cti, new_pose;
        orc_pose_inverse(&cps_t, &cti);
        orc_pose_compose(&v.X2, &cti, &new_pose);
        double x_o = gsx - gtx, y_o = gsy - gty;
        double ini = sqrt(x_o * x_o + y_o * y_o);
        double rpy[3];
        orc_pose_rpy(&new_pose, rpy);
        double lx, ly;
        if (kp[4] < Mt / 2) {
            int gi = Mt / 2 - (int)kp[4];
            lx = new_pose.t[0] + gr_t[gi] * cos(rpy[2] + PI / 2 - PI);
            ly = new_pose.t[1] + gr_t[gi] * sin(rpy[2] + PI / 2 - PI);
        } else {
            int gi = (int)kp[4] - Mt / 2;
            lx = new_pose.t[0] + gr_t[gi] * cos(rpy[2] - PI / 2 - PI);
            ly = new_pose.t[1] + gr_t[gi] * sin(rpy[2] - PI / 2 - PI);
        }
        double x_n = gsx - lx, y_n = gsy - ly;
        double fin = sqrt(x_n * x_n + y_n * y_n);
        o->score = ini / fin - 2;
        /* Marginals(graph, result).marginalCovariance(X2).diagonal() (:956-959) */
        double r[MROWS], J[MROWS * MDIM], H[MDIM * MDIM];
        mini_lin(&m, &v, r, J);
        for (int a = 0; a < MDIM; ++a)
            for (int b = 0; b < MDIM; ++b) {
                double t = 0;
                for (int k = 0; k < MROWS; ++k) t += J[k * MDIM + a] * J[k * MDIM + b];
                H[a * MDIM + b] = t;
            }
        if (chol(H, MDIM) == 0) {
            for (int c = 0; c < 6; ++c) {
                double e[MDIM];
                memset(e, 0, sizeof e);
                e[9 + c] = 1;
                chol_solve(H, MDIM, e);
                o->var[c] = e[9 + c];
            }
        } else for (int c = 0; c < 6; ++c) o->var[c] = NAN;
        /* (Tp_s * cps_s^-1).between(X2 * cps_t^-1) (:958) */
        orc_pose csi, src, rel;
        orc_pose_inverse(&cps_s, &csi);
        orc_pose_compose(&Tp_s, &csi, &src);
        orc_pose_between(&src, &new_pose, &rel);
        memcpy(o->rel, rel.R, sizeof(double) * 9);
        memcpy(o->rel + 9, rel.t, sizeof(double) * 3);
    }
    return n;
}

/* ---- LMTriaFactor (LMtriangulatefactor.cpp:10-27) + Optimizer::TriangulateOneLandmark (optimizer.cpp:984-1021):
 * 3-DoF LM on one landmark with both ping poses fixed.  The factor's residual and 2x3 Jacobian are exactly the
 * residual and H1 of SssPointFactor (same expressions, pose held constant).  Graph: LMTriaFactor(s), LMTriaFactor(t),
 * point prior at lm_ini with sigmas (10, 10, |xy baseline| / 100); GTSAM LM defaults.
 * A zero baseline would make the prior a constrained model in GTSAM: clamped at 1e-9 like sig_odo above. */
#define TROWS 7
typedef struct { orc_pose Ts_s, Ts_t, Tp_s, Tp_t; double slant_s, slant_t, sig_s[2], sig_t[2], sig_p[3], ini[3]; } tri_t;
static void tri_lin(const tri_t* m, const double* p, double* r, double* J /* 7 x 3 or NULL */)
{
    double ee[2], H1[6];
    orc_sss_factor(p, &m->Tp_s, &m->Ts_s, m->slant_s, 0.0, ee, J ? H1 : NULL, NULL);
    for (int i = 0; i < 2; ++i) { r[i] = ee[i] / m->sig_s[i]; if (J) for (int j = 0; j < 3; ++j) J[i * 3 + j] = H1[3 * i + j] / m->sig_s[i]; }
    orc_sss_factor(p, &m->Tp_t, &m->Ts_t, m->slant_t, 0.0, ee, J ? H1 : NULL, NULL);
    for (int i = 0; i < 2; ++i) { r[2 + i] = ee[i] / m->sig_t[i]; if (J) for (int j = 0; j < 3; ++j) J[(2 + i) * 3 + j] = H1[3 * i + j] / m->sig_t[i]; }
    for (int i = 0; i < 3; ++i) { r[4 + i] = (p[i] - m->ini[i]) / m->sig_p[i]; if (J) for (int j = 0; j < 3; ++j) J[(4 + i) * 3 + j] = (i == j) ? 1.0 / m->sig_p[i] : 0.0; }
}
static double tri_err(const tri_t* m, const double* p)
{
    double r[TROWS]; tri_lin(m, p, r, NULL);
    double s = 0; for (int i = 0; i < TROWS; ++i) s += r[i] * r[i];
    return 0.5 * s;
}
int orc_triangulate_one(const double kp7[7], const orc_pose* Ts_s, const orc_pose* Ts_t, const orc_pose* Tp_s, const orc_pose* Tp_t,
                        const double lm_ini[3], double out[3])
{
    const double PI = ORC_PI_REF, sigma_r = 0.1, alpha_bw = 0.1 * PI / 180;
    tri_t m;
    m.Ts_s = *Ts_s; m.Ts_t = *Ts_t; m.Tp_s = *Tp_s; m.Tp_t = *Tp_t;
    m.slant_s = kp7[2]; m.slant_t = kp7[5];
    m.sig_s[0] = sigma_r; m.sig_s[1] = kp7[2] * alpha_bw; m.sig_t[0] = sigma_r; m.sig_t[1] = kp7[5] * alpha_bw;
    const double dx = Tp_s->t[0] - Tp_t->t[0], dy = Tp_s->t[1] - Tp_t->t[1];
    m.sig_p[0] = 10.0; m.sig_p[1] = 10.0; m.sig_p[2] = sqrt(dx * dx + dy * dy) / 100;
    if (m.sig_p[2] < 1e-9) m.sig_p[2] = 1e-9;
    for (int i = 0; i < 3; ++i) m.ini[i] = lm_ini[i];
    double p[3] = { lm_ini[0], lm_ini[1], lm_ini[2] };
    const double relTol = 1e-5, absTol = 1e-5, lamMax = 1e5, minFid = 1e-3;
    double lambda = 1e-5, err = tri_err(&m, p), cur;
    int iters = 0;
    if (err > 0) do {
        cur = err;
        double r[TROWS], J[TROWS * 3], H[9], g[3];
        tri_lin(&m, p, r, J);
        for (int a = 0; a < 3; ++a) {
            double s = 0; for (int k = 0; k < TROWS; ++k) s += J[k * 3 + a] * r[k];
            g[a] = s;
            for (int b = 0; b < 3; ++b) { double t = 0; for (int k = 0; k < TROWS; ++k) t += J[k * 3 + a] * J[k * 3 + b]; H[a * 3 + b] = t; }
        }
        double oldLin = 0; for (int k = 0; k < TROWS; ++k) oldLin += r[k] * r[k];
        oldLin *= 0.5;
        for (;;) {
            double A[9], d[3];
            memcpy(A, H, sizeof A);
            for (int a = 0; a < 3; ++a) { A[a * 3 + a] += lambda; d[a] = -g[a]; }
            int ok = chol(A, 3) == 0, success = 0, stop = 0;
            double newErr = 0, np_[3];
            if (ok) {
                chol_solve(A, 3, d);
                double newLin = 0;
                for (int k = 0; k < TROWS; ++k) { double s = r[k]; for (int a = 0; a < 3; ++a) s += J[k * 3 + a] * d[a]; newLin += s * s; }
                newLin *= 0.5;
                const double linChange = oldLin - newLin;
                if (linChange >= 0) {
                    for (int a = 0; a < 3; ++a) np_[a] = p[a] + d[a];
                    newErr = tri_err(&m, np_);
                    const double costChange = err - newErr;
                    if (linChange > 2.220446049250313e-16 * oldLin) success = (costChange / linChange) > minFid;
                    if (fabs(costChange) < relTol * err) stop = 1;
                }
            }
            if (success) { for (int a = 0; a < 3; ++a) p[a] = np_[a]; err = newErr; lambda /= 10; ++iters; break; }
            else if (!stop) { lambda *= 10; if (lambda >= lamMax) break; }
            else break;
        }
    } while (iters < 100 && !((err <= 0) || ((cur - err) / cur <= relTol) || ((cur - err) <= absTol)) && isfinite(cur));
    for (int a = 0; a < 3; ++a) out[a] = p[a];
    return iters;
}

/* the call site in LoopClosingTFs (optimizer.cpp:907-921, eval_2): landmark triangulated from the (yaw-compensated,
 * sticky) DR poses of both pings, initialised as in :789-795; out7 = [x y z | |range_s err| |plane_s| |range_t err| |plane_t|] */
int orc_triangulate(const double* kp7, int n, const double* pose6_s, const double* alt_s, const double* gr_s, int Ns, int Ms,
                    const double* pose6_t, const double* alt_t, const double* gr_t, int Nt, int Mt, double* out7)
{
    const double PI = ORC_PI_REF;
    orc_pose cps_s, cps_t, ident;
    memset(&ident, 0, sizeof ident); ident.R[0] = ident.R[4] = ident.R[8] = 1;
    cps_s = ident; cps_t = ident;
    const double flipv[3] = { 0, 0, PI };
    for (int i = 0; i < n; ++i) {
        const double* kp = kp7 + (size_t)i * 7;
        const int id_s = (int)kp[0], id_t = (int)kp[3];
        if (fabs(pose6_s[(size_t)id_s * 6 + 2]) > 2 * PI / 3) orc_so3_exp(flipv, cps_s.R);
        if (fabs(pose6_t[(size_t)id_t * 6 + 2]) > 2 * PI / 3) orc_so3_exp(flipv, cps_t.R);
        orc_pose Ps, Pt, Tp_s, Tp_t;
        orc_pose_from_rodrigues(pose6_s + (size_t)id_s * 6, &Ps);
        orc_pose_from_rodrigues(pose6_t + (size_t)id_t * 6, &Pt);
        orc_pose_compose(&Ps, &cps_s, &Tp_s);
        orc_pose_compose(&Pt, &cps_t, &Tp_t);
        double gsx, gsy, gtx, gty, L0[3];
        orc_geo_at(pose6_s, gr_s, Ns, Ms, id_s, (int)kp[1], &gsx, &gsy);
        orc_geo_at(pose6_t, gr_t, Nt, Mt, id_t, (int)kp[4], &gtx, &gty);
        L0[0] = (gsx + gtx) / 2; L0[1] = (gsy + gty) / 2;
        L0[2] = ((pose6_s[(size_t)id_s * 6 + 5] - alt_s[id_s]) + (pose6_t[(size_t)id_t * 6 + 5] - alt_t[id_t])) / 2;
        double* o = out7 + (size_t)i * 7;
        orc_triangulate_one(kp, &ident, &ident, &Tp_s, &Tp_t, L0, o);
        double e[2];
        orc_sss_factor(o, &Tp_s, &ident, kp[2], 0.0, e, NULL, NULL); o[3] = fabs(e[0]); o[4] = fabs(e[1]);
        orc_sss_factor(o, &Tp_t, &ident, kp[5], 0.0, e, NULL, NULL); o[5] = fabs(e[0]); o[6] = fabs(e[1]);
    }
    return n;
}
