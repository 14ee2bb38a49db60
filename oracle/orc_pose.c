/* oracle/orc_pose.c -- GTSAM 4.2 Rot3/Pose3 semantics restated (GTSAM is not in the
 * reference tree: pinned version "4.2", README.md:35).  Default build flags: GTSAM_POSE3_EXPMAP=ON,
 * GTSAM_ROT3_EXPMAP=ON, rotation-matrix Rot3, tangent order [omega(3), v(3)], retract(T,xi) = T * Expmap(xi).
 * Also SssPointFactor (src/core/SSSpointfactor.cpp:11-80).  Test infrastructure, see orc.h. */
#include "orc.h"
#include <math.h>
#include <string.h>

static void mat3_mul(const double* A, const double* B, double* C)
{
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, T, sizeof T);
}
static void mat3_vec(const double* A, const double* v, double* o)
{
    double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
static void mat3_tvec(const double* A, const double* v, double* o)
{
    double t0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    double t1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    double t2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
static void cross3(const double* a, const double* b, double* o)
{
    double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    o[0] = t0; o[1] = t1; o[2] = t2;
}

/* SO3::Expmap / Rot3::Rodrigues: Rodrigues' formula, first order below theta^2 <= eps */
void orc_so3_exp(const double w[3], double R[9])
{
    double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double W[9] = { 0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0 };
    if (t2 <= 2.220446049250313e-16) {
        for (int i = 0; i < 9; ++i) R[i] = W[i];
        R[0] += 1; R[4] += 1; R[8] += 1;
        return;
    }
    double th = sqrt(t2), s = sin(th), s2 = sin(th / 2), omc = 2 * s2 * s2;
    double K[9], KK[9];
    for (int i = 0; i < 9; ++i) K[i] = W[i] / th;
    mat3_mul(K, K, KK);
    for (int i = 0; i < 9; ++i) R[i] = s * K[i] + omc * KK[i];
    R[0] += 1; R[4] += 1; R[8] += 1;
}

/* SO3::Logmap */
void orc_so3_log(const double R[9], double w[3])
{
    const double R11 = R[0], R12 = R[1], R13 = R[2], R21 = R[3], R22 = R[4], R23 = R[5], R31 = R[6], R32 = R[7], R33 = R[8];
    const double tr = R11 + R22 + R33;
    if (tr + 1.0 < 1e-3) {
        /* theta close to pi */
        if (R33 > R22 && R33 > R11) {
            const double Wv = R21 - R12, Q1 = 2.0 + 2.0 * R33, Q2 = R31 + R13, Q3 = R23 + R32;
            const double r = sqrt(Q1), one_over_r = 1 / r, nrm = sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + Wv * Wv);
            const double sgn = Wv < 0 ? -1.0 : 1.0, mag = M_PI - (2 * sgn * Wv) / nrm, sc = 0.5 * one_over_r * mag;
            w[0] = sgn * sc * Q2; w[1] = sgn * sc * Q3; w[2] = sgn * sc * Q1;
        } else if (R22 > R11) {
            const double Wv = R13 - R31, Q1 = 2.0 + 2.0 * R22, Q2 = R23 + R32, Q3 = R12 + R21;
            const double r = sqrt(Q1), one_over_r = 1 / r, nrm = sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + Wv * Wv);
            const double sgn = Wv < 0 ? -1.0 : 1.0, mag = M_PI - (2 * sgn * Wv) / nrm, sc = 0.5 * one_over_r * mag;
            w[0] = sgn * sc * Q3; w[1] = sgn * sc * Q1; w[2] = sgn * sc * Q2;
        } else {
            const double Wv = R32 - R23, Q1 = 2.0 + 2.0 * R11, Q2 = R12 + R21, Q3 = R31 + R13;
            const double r = sqrt(Q1), one_over_r = 1 / r, nrm = sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + Wv * Wv);
            const double sgn = Wv < 0 ? -1.0 : 1.0, mag = M_PI - (2 * sgn * Wv) / nrm, sc = 0.5 * one_over_r * mag;
            w[0] = sgn * sc * Q1; w[1] = sgn * sc * Q2; w[2] = sgn * sc * Q3;
        }
        return;
    }
    double magnitude;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-6) {
        double theta = acos((tr - 1.0) / 2.0);
        magnitude = theta / (2.0 * sin(theta));
    } else {
        magnitude = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
    }
    w[0] = magnitude * (R32 - R23); w[1] = magnitude * (R13 - R31); w[2] = magnitude * (R21 - R12);
}

/* Pose3(Rot3::Rodrigues(roll,pitch,yaw), Point3(x,y,z)): the three angles are used as a ROTATION VECTOR
 * (optimizer.cpp:150-152,735-738) */
void orc_pose_from_rodrigues(const double p6[6], orc_pose* T)
{
    orc_so3_exp(p6, T->R);
    T->t[0] = p6[3]; T->t[1] = p6[4]; T->t[2] = p6[5];
}
void orc_pose_compose(const orc_pose* A, const orc_pose* B, orc_pose* C)
{
    orc_pose O;
    mat3_mul(A->R, B->R, O.R);
    mat3_vec(A->R, B->t, O.t);
    O.t[0] += A->t[0]; O.t[1] += A->t[1]; O.t[2] += A->t[2];
    *C = O;
}
void orc_pose_inverse(const orc_pose* A, orc_pose* B)
{
    orc_pose O;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) O.R[3 * i + j] = A->R[3 * j + i];
    mat3_vec(O.R, A->t, O.t);
    O.t[0] = -O.t[0]; O.t[1] = -O.t[1]; O.t[2] = -O.t[2];
    *B = O;
}
void orc_pose_between(const orc_pose* A, const orc_pose* B, orc_pose* C)
{
    orc_pose Ai;
    orc_pose_inverse(A, &Ai);
    orc_pose_compose(&Ai, B, C);
}
/* Pose3::Expmap */
void orc_pose_exp(const double xi[6], orc_pose* T)
{
    const double* w = xi; const double* v = xi + 3;
    orc_so3_exp(w, T->R);
    double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (t2 > 2.220446049250313e-16) {
        double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
        double tpar[3] = { w[0] * wv, w[1] * wv, w[2] * wv };
        double wxv[3], Rwxv[3];
        cross3(w, v, wxv);
        mat3_vec(T->R, wxv, Rwxv);
        for (int i = 0; i < 3; ++i) T->t[i] = (wxv[i] - Rwxv[i] + tpar[i]) / t2;
    } else {
        T->t[0] = v[0]; T->t[1] = v[1]; T->t[2] = v[2];
    }
}
/* Pose3::Logmap */
void orc_pose_log(const orc_pose* T, double xi[6])
{
    double w[3];
    orc_so3_log(T->R, w);
    double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
    if (t < 1e-10) { xi[3] = T->t[0]; xi[4] = T->t[1]; xi[5] = T->t[2]; return; }
    double k[3] = { w[0] / t, w[1] / t, w[2] / t };
    double Tan = tan(0.5 * t);
    double WT[3], WWT[3];
    cross3(k, T->t, WT);
    cross3(k, WT, WWT);
    for (int i = 0; i < 3; ++i) xi[3 + i] = T->t[i] - (0.5 * t) * WT[i] + (1 - t / (2. * Tan)) * WWT[i];
}
/* Pose3::AdjointMap: [R 0; [t]x R  R] */
void orc_pose_adjoint(const orc_pose* T, double Ad[36])
{
    const double* R = T->R; const double* t = T->t;
    double tx[9] = { 0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0 }, A[9];
    mat3_mul(tx, R, A);
    memset(Ad, 0, sizeof(double) * 36);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Ad[6 * i + j] = R[3 * i + j];
            Ad[6 * (i + 3) + j] = A[3 * i + j];
            Ad[6 * (i + 3) + j + 3] = R[3 * i + j];
        }
}
void orc_pose_retract(const orc_pose* T, const double xi[6], orc_pose* out)
{
    orc_pose E;
    orc_pose_exp(xi, &E);
    orc_pose_compose(T, &E, out);
}
/* Rot3::rpy(): R = Rz(y) Ry(p) Rx(r) (optimizer.cpp:871,1181) */
void orc_pose_rpy(const orc_pose* T, double rpy[3])
{
    const double* R = T->R;
    rpy[0] = atan2(R[7], R[8]);
    rpy[1] = atan2(-R[6], sqrt(R[7] * R[7] + R[8] * R[8]));
    rpy[2] = atan2(R[3], R[0]);
}

/* SssPointFactor::evaluateError, plan_a (SSSpointfactor.cpp:11-80). H1 2x3, H2 2x6 row-major.
 * The translation block of H2 is -(Rs^T R^T), as written in the reference. */
void orc_sss_factor(const double p[3], const orc_pose* T, const orc_pose* Ts, double mx, double my,
                    double e[2], double H1[6], double H2[12])
{
    double d[3] = { p[0] - T->t[0], p[1] - T->t[1], p[2] - T->t[2] }, pm[3], d2[3], ps[3];
    mat3_tvec(T->R, d, pm);
    d2[0] = pm[0] - Ts->t[0]; d2[1] = pm[1] - Ts->t[1]; d2[2] = pm[2] - Ts->t[2];
    mat3_tvec(Ts->R, d2, ps);
    double nrm = sqrt(ps[0] * ps[0] + ps[1] * ps[1] + ps[2] * ps[2]);
    e[0] = nrm - mx; e[1] = ps[0] - my;
    double RsT[9], RT[9], J[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { RsT[3 * i + j] = Ts->R[3 * j + i]; RT[3 * i + j] = T->R[3 * j + i]; }
    mat3_mul(RsT, RT, J);
    if (H1) {
        for (int j = 0; j < 3; ++j) {
            H1[j] = (ps[0] * J[j] + ps[1] * J[3 + j] + ps[2] * J[6 + j]) / nrm;
            H1[3 + j] = J[j];
        }
    }
    if (H2) {
        double mxm[9] = { 0, -pm[2], pm[1], pm[2], 0, -pm[0], -pm[1], pm[0], 0 }, Br[9];
        mat3_mul(RsT, mxm, Br);
        double Jp[18];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { Jp[6 * i + j] = Br[3 * i + j]; Jp[6 * i + 3 + j] = -J[3 * i + j]; }
        for (int j = 0; j < 6; ++j) {
            H2[j] = (ps[0] * Jp[j] + ps[1] * Jp[6 + j] + ps[2] * Jp[12 + j]) / nrm;
            H2[6 + j] = Jp[j];
        }
    }
}
