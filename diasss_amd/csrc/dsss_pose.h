// diasss_amd/csrc/dsss_pose.h -- SO(3)/SE(3) algebra with GTSAM 4.2 Rot3/Pose3 conventions, usable from host
// and device code.  GTSAM is not in the reference tree (README.md:35 pins "4.2"); conventions per SURVEY.md A.2:
// rotation-matrix Rot3, tangent [omega, v], retract(T, xi) = T * Expmap(xi), full exponential maps.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

#define PDEV __host__ __device__ inline

struct pose_t { double R[9]; double t[3]; };   // R row-major

PDEV void m3_mul(const double* A, const double* B, double* C)
{
    double T[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) C[i] = T[i];
}
PDEV void m3_vec(const double* A, const double* v, double* o)
{
    const double a = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    const double b = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    const double c = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
PDEV void m3_tvec(const double* A, const double* v, double* o)
{
    const double a = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    const double b = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    const double c = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
PDEV void v3_cross(const double* a, const double* b, double* o)
{
    const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

PDEV void so3_exp(const double* w, double* R)          // Rot3::Rodrigues / SO3::Expmap
{
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double W[9] = { 0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0 };
    if (t2 <= 2.220446049250313e-16) {
        for (int i = 0; i < 9; ++i) R[i] = W[i];
        R[0] += 1; R[4] += 1; R[8] += 1;
        return;
    }
    const double th = sqrt(t2), s = sin(th), s2 = sin(th / 2), omc = 2 * s2 * s2;
    double K[9], KK[9];
    for (int i = 0; i < 9; ++i) K[i] = W[i] / th;
    m3_mul(K, K, KK);
    for (int i = 0; i < 9; ++i) R[i] = s * K[i] + omc * KK[i];
    R[0] += 1; R[4] += 1; R[8] += 1;
}

PDEV void so3_log(const double* R, double* w)          // SO3::Logmap incl. the near-pi branch
{
    const double R11 = R[0], R12 = R[1], R13 = R[2], R21 = R[3], R22 = R[4], R23 = R[5], R31 = R[6], R32 = R[7], R33 = R[8];
    const double tr = R11 + R22 + R33;
    const double PI_ = 3.14159265358979323846;
    if (tr + 1.0 < 1e-3) {
        double Wv, Q1, Q2, Q3; int perm;
        if (R33 > R22 && R33 > R11) { Wv = R21 - R12; Q1 = 2.0 + 2.0 * R33; Q2 = R31 + R13; Q3 = R23 + R32; perm = 0; }
        else if (R22 > R11) { Wv = R13 - R31; Q1 = 2.0 + 2.0 * R22; Q2 = R23 + R32; Q3 = R12 + R21; perm = 1; }
        else { Wv = R32 - R23; Q1 = 2.0 + 2.0 * R11; Q2 = R12 + R21; Q3 = R31 + R13; perm = 2; }
        const double r = sqrt(Q1), nrm = sqrt(Q1 * Q1 + Q2 * Q2 + Q3 * Q3 + Wv * Wv);
        const double sgn = Wv < 0 ? -1.0 : 1.0, mag = PI_ - (2 * sgn * Wv) / nrm, sc = 0.5 * (1 / r) * mag;
        if (perm == 0) { w[0] = sgn * sc * Q2; w[1] = sgn * sc * Q3; w[2] = sgn * sc * Q1; }
        else if (perm == 1) { w[0] = sgn * sc * Q3; w[1] = sgn * sc * Q1; w[2] = sgn * sc * Q2; }
        else { w[0] = sgn * sc * Q1; w[1] = sgn * sc * Q2; w[2] = sgn * sc * Q3; }
        return;
    }
    double mag;
    const double tr_3 = tr - 3.0;
    if (tr_3 < -1e-6) { const double th = acos((tr - 1.0) / 2.0); mag = th / (2.0 * sin(th)); }
    else mag = 0.5 - tr_3 / 12.0 + tr_3 * tr_3 / 60.0;
    w[0] = mag * (R32 - R23); w[1] = mag * (R13 - R31); w[2] = mag * (R21 - R12);
}

PDEV void pose_identity(pose_t* T) { for (int i = 0; i < 9; ++i) T->R[i] = 0; T->R[0] = T->R[4] = T->R[8] = 1; T->t[0] = T->t[1] = T->t[2] = 0; }
PDEV void pose_from_rodrigues(const double* p6, pose_t* T) { so3_exp(p6, T->R); T->t[0] = p6[3]; T->t[1] = p6[4]; T->t[2] = p6[5]; }
PDEV void pose_compose(const pose_t* A, const pose_t* B, pose_t* C)
{
    pose_t O;
    m3_mul(A->R, B->R, O.R);
    m3_vec(A->R, B->t, O.t);
    O.t[0] += A->t[0]; O.t[1] += A->t[1]; O.t[2] += A->t[2];
    *C = O;
}
PDEV void pose_inverse(const pose_t* A, pose_t* B)
{
    pose_t O;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) O.R[3 * i + j] = A->R[3 * j + i];
    m3_vec(O.R, A->t, O.t);
    O.t[0] = -O.t[0]; O.t[1] = -O.t[1]; O.t[2] = -O.t[2];
    *B = O;
}
PDEV void pose_between(const pose_t* A, const pose_t* B, pose_t* C) { pose_t Ai; pose_inverse(A, &Ai); pose_compose(&Ai, B, C); }

PDEV void pose_exp(const double* xi, pose_t* T)        // Pose3::Expmap
{
    const double* w = xi; const double* v = xi + 3;
    so3_exp(w, T->R);
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (t2 > 2.220446049250313e-16) {
        const double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
        double wxv[3], Rwxv[3];
        v3_cross(w, v, wxv);
        m3_vec(T->R, wxv, Rwxv);
        for (int i = 0; i < 3; ++i) T->t[i] = (wxv[i] - Rwxv[i] + w[i] * wv) / t2;
    } else { T->t[0] = v[0]; T->t[1] = v[1]; T->t[2] = v[2]; }
}
PDEV void pose_log(const pose_t* T, double* xi)        // Pose3::Logmap
{
    double w[3];
    so3_log(T->R, w);
    const double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
    if (t < 1e-10) { xi[3] = T->t[0]; xi[4] = T->t[1]; xi[5] = T->t[2]; return; }
    const double k[3] = { w[0] / t, w[1] / t, w[2] / t };
    const double Tan = tan(0.5 * t);
    double WT[3], WWT[3];
    v3_cross(k, T->t, WT);
    v3_cross(k, WT, WWT);
    for (int i = 0; i < 3; ++i) xi[3 + i] = T->t[i] - (0.5 * t) * WT[i] + (1 - t / (2. * Tan)) * WWT[i];
}
PDEV void pose_adjoint(const pose_t* T, double* Ad)    // Pose3::AdjointMap = [R 0; [t]x R, R], row-major 6x6
{
    const double* R = T->R; const double* t = T->t;
    const double tx[9] = { 0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0 };
    double A[9];
    m3_mul(tx, R, A);
    for (int i = 0; i < 36; ++i) Ad[i] = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { Ad[6 * i + j] = R[3 * i + j]; Ad[6 * (i + 3) + j] = A[3 * i + j]; Ad[6 * (i + 3) + j + 3] = R[3 * i + j]; }
}
PDEV void pose_retract(const pose_t* T, const double* xi, pose_t* out) { pose_t E; pose_exp(xi, &E); pose_compose(T, &E, out); }
PDEV void pose_rpy(const pose_t* T, double* rpy)       // Rot3::rpy(): R = Rz(y) Ry(p) Rx(r)
{
    const double* R = T->R;
    rpy[0] = atan2(R[7], R[8]);
    rpy[1] = atan2(-R[6], sqrt(R[7] * R[7] + R[8] * R[8]));
    rpy[2] = atan2(R[3], R[0]);
}

// SssPointFactor::evaluateError, plan_a (/root/reference/src/core/SSSpointfactor.cpp:11-80); sensor offset Ts = identity
// rotation + translation ts (frame.cpp:38-39 sets it to zero).  H1 2x3, H2 2x6 row-major.  The translation block of
// H2 is -(Rs^T R^T) exactly as the reference writes it, not the chart-consistent -Rs^T.
PDEV void sss_factor(const double* p, const pose_t* T, double mx, double my, double* e, double* H1, double* H2)
{
    const double d[3] = { p[0] - T->t[0], p[1] - T->t[1], p[2] - T->t[2] };
    double pm[3];
    m3_tvec(T->R, d, pm);
    const double nrm = sqrt(pm[0] * pm[0] + pm[1] * pm[1] + pm[2] * pm[2]);
    e[0] = nrm - mx; e[1] = pm[0] - my;
    if (!H1) return;
    double J[9];                                       // Rs^T R^T with Rs = I
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[3 * i + j] = T->R[3 * j + i];
    for (int j = 0; j < 3; ++j) { H1[j] = (pm[0] * J[j] + pm[1] * J[3 + j] + pm[2] * J[6 + j]) / nrm; H1[3 + j] = J[j]; }
    const double Br[9] = { 0, -pm[2], pm[1], pm[2], 0, -pm[0], -pm[1], pm[0], 0 };
    double Jp[18];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Jp[6 * i + j] = Br[3 * i + j]; Jp[6 * i + 3 + j] = -J[3 * i + j]; }
    for (int j = 0; j < 6; ++j) { H2[j] = (pm[0] * Jp[j] + pm[1] * Jp[6 + j] + pm[2] * Jp[12 + j]) / nrm; H2[6 + j] = Jp[j]; }
}
