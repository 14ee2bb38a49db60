/* oracle/orc_posegraph.c -- pose-graph assembly of Optimizer::TrajOptimizationAll
 * (/root/reference/src/core/optimizer.cpp:101-279) solved by BATCH Levenberg-Marquardt (GTSAM default LM
 * schedule) in place of the reference's incremental iSAM2 (north_star: "the GTSAM
 * NonlinearFactorGraph replaced outright").  Factors: prior on X0 (sigma 1e-6, :164-170), odometry
 * BetweenFactor chain over every ping of every frame (:173-200, sigmas :28), at most one loop-closure
 * BetweenFactor per target ping (:203-258, Diagonal::Variances).
 * Linear solve: block-tridiagonal chain eliminated onto the LC-touched poses, then an envelope Cholesky of
 * the reduced system (exact).  Test infrastructure, see orc.h. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_pg_params_default(orc_pg_params* p)
{
    p->max_iters = 100; p->rel_tol = 1e-5; p->abs_tol = 1e-5; p->lambda0 = 1e-5; p->lambda_factor = 10;
    p->lambda_max = 1e5; p->min_fidelity = 1e-3; p->add_noise = 1;
}

/* LC selection loop of TrajOptimizationAll (:203-258): for target frame i, ping j the LAST pair (k,i) holding a
 * kp whose target ping is j wins, and within it the FIRST such kp; the factor is added only if its score > 0.
 * Global pose id = frame offset + ping.  pair_off[k]..pair_off[k+1] index kp7/lcs of pair k. */
int orc_pg_select_lc(int F, const int* frame_rows, int npairs, const int* pair_s, const int* pair_t,
                     const int* pair_off, const double* kp7, const orc_lc* lcs, orc_lc_edge* edges, int cap)
{
    int* off = (int*)malloc(sizeof(int) * (F + 1));
    off[0] = 0;
    for (int f = 0; f < F; ++f) off[f + 1] = off[f] + frame_rows[f];
    int ne = 0;
    for (int i = 1; i < F; ++i) {
        for (int j = 0; j < frame_rows[i]; ++j) {
            int kps_id = -1, pair_id = -1;
            for (int k = 0; k < npairs; ++k) {
                if (pair_t[k] != i) continue;
                for (int q = pair_off[k]; q < pair_off[k + 1]; ++q) {
                    if ((int)kp7[(size_t)q * 7 + 3] == j) { kps_id = q; pair_id = k; break; }
                }
            }
            /* usable = score > 0 (:234) AND a finite score and finite positive variances: the reference's Marginals would
             * throw on a singular 15x15 system (uncaught -> abort); both sides of the parity drop such a measurement */
            int usable = kps_id != -1 && lcs[kps_id].score > 0 && isfinite(lcs[kps_id].score);
            for (int c = 0; usable && c < 6; ++c) usable = lcs[kps_id].var[c] > 0 && isfinite(lcs[kps_id].var[c]);
            if (usable) {
                if (ne >= cap) { free(off); return ne; }
                int id_1 = (int)kp7[(size_t)kps_id * 7 + 0], id_2 = (int)kp7[(size_t)kps_id * 7 + 3];
                edges[ne].a = off[pair_s[pair_id]] + id_1;
                edges[ne].b = off[i] + id_2;
                memcpy(edges[ne].rel, lcs[kps_id].rel, sizeof(double) * 12);
                memcpy(edges[ne].var, lcs[kps_id].var, sizeof(double) * 6);
                ++ne;
            }
        }
    }
    free(off);
    return ne;
}

/* ---- small dense helpers on 6x6 blocks (row-major) */
static int chol6(double* A)
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
static void chol6_solve(const double* L, double* b, int nrhs) /* b is 6 x nrhs row-major */
{
    for (int c = 0; c < nrhs; ++c) {
        for (int i = 0; i < 6; ++i) { double s = b[i * nrhs + c]; for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * b[k * nrhs + c]; b[i * nrhs + c] = s / L[i * 6 + i]; }
        for (int i = 5; i >= 0; --i) { double s = b[i * nrhs + c]; for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * b[k * nrhs + c]; b[i * nrhs + c] = s / L[i * 6 + i]; }
    }
}

typedef struct { int i, j; orc_pose meas; double w[6]; } pg_factor;  /* between(i,j); i<0: prior on j */

/* Second linear solver for the reduced system (round 6).  The envelope Cholesky below is exact but as wide as a leg's
 * loop closures: at the size of BASELINE config 3 (23 k separators) it needs hours.  When a callback is installed the
 * reduced matrix leaves as lower-triangular 6 x 6 blocks (bi >= bj, row-major, duplicates are to be SUMMED) and the
 * callback overwrites rhs with the solution (oracle/binding.py: scipy.sparse.linalg.splu) -- the LM loop, the chain
 * condensation and the back-substitution stay the ones of this file.  Returns 0 on success. */
static orc_pg_reduced_solver_fn g_reduced_solver = NULL;
void orc_pg_set_reduced_solver(orc_pg_reduced_solver_fn fn) { g_reduced_solver = fn; }
/* Steps of iterative refinement of every LM trial's linear solve against the FULL system (residual of the normal
 * equations accumulated in long double from the factors' Jacobians, correction through the same condensation +
 * reduced solve): with it the trial step is the exact solution of the damped normal equations to ~1e-19 relative --
 * the yardstick for "how far does a solver's rounding move the LM iterate" (tests/test_gpu_configs.py). */
static int g_full_refine = 0;
void orc_pg_set_full_refine(int steps) { g_full_refine = steps; }

typedef struct {
    int n, nf;
    pg_factor* f;
    /* linearisation */
    double* r;      /* nf x 6 whitened residual */
    double* Ji;     /* nf x 36 whitened Jacobian wrt i (6x6) */
    double* Jj;     /* nf x 36 whitened Jacobian wrt j */
} pg_t;

static void pg_factor_eval(const pg_factor* f, const orc_pose* X, double* r, double* Ji, double* Jj)
{
    double xi[6];
    if (f->i < 0) {
        orc_pose d;
        orc_pose_between(&f->meas, &X[f->j], &d);
        orc_pose_log(&d, xi);
        for (int k = 0; k < 6; ++k) r[k] = xi[k] * f->w[k];
        if (Jj) { memset(Jj, 0, sizeof(double) * 36); for (int k = 0; k < 6; ++k) Jj[k * 6 + k] = f->w[k]; }
        if (Ji) memset(Ji, 0, sizeof(double) * 36);
        return;
    }
    orc_pose h, e;
    orc_pose_between(&X[f->i], &X[f->j], &h);
    orc_pose_between(&f->meas, &h, &e);
    orc_pose_log(&e, xi);
    for (int k = 0; k < 6; ++k) r[k] = xi[k] * f->w[k];
    if (Ji) {
        orc_pose hi; double Ad[36];
        orc_pose_inverse(&h, &hi);
        orc_pose_adjoint(&hi, Ad);
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) Ji[a * 6 + b] = -Ad[a * 6 + b] * f->w[a];
        memset(Jj, 0, sizeof(double) * 36);
        for (int k = 0; k < 6; ++k) Jj[k * 6 + k] = f->w[k];
    }
}
static double pg_error(const pg_t* g, const orc_pose* X)
{
    double s = 0;
    for (int k = 0; k < g->nf; ++k) {
        double r[6];
        pg_factor_eval(&g->f[k], X, r, NULL, NULL);
        for (int a = 0; a < 6; ++a) s += r[a] * r[a];
    }
    return 0.5 * s;
}

/* Solve (H + lambda I) delta = -g for H = sum J^T J.  Chain factors are f[0] (prior) and f[1..n-1]
 * (odometry i-1 -> i); LC factors follow.  Returns 0 on success. */
static int pg_solve(const pg_t* g, double lambda, double* delta, const double* neg_rhs)
{
    int n = g->n;
    double* D = (double*)calloc((size_t)n * 36, sizeof(double));      /* diagonal blocks */
    double* C = (double*)calloc((size_t)n * 36, sizeof(double));      /* C[i] = H(i, i+1) */
    double* gr = (double*)calloc((size_t)n * 6, sizeof(double));      /* gradient J^T r */
    char* sep = (char*)calloc(n, 1);
    int nlc = g->nf - n;
    for (int k = 0; k < g->nf; ++k) {
        const pg_factor* f = &g->f[k];
        const double* r = g->r + (size_t)k * 6;
        const double* Ji = g->Ji + (size_t)k * 36;
        const double* Jj = g->Jj + (size_t)k * 36;
        for (int a = 0; a < 6; ++a) {
            for (int b = 0; b < 6; ++b) {
                double sjj = 0, sii = 0, sij = 0;
                for (int q = 0; q < 6; ++q) {
                    sjj += Jj[q * 6 + a] * Jj[q * 6 + b];
                    if (f->i >= 0) { sii += Ji[q * 6 + a] * Ji[q * 6 + b]; sij += Ji[q * 6 + a] * Jj[q * 6 + b]; }
                }
                D[(size_t)f->j * 36 + a * 6 + b] += sjj;
                if (f->i >= 0) {
                    D[(size_t)f->i * 36 + a * 6 + b] += sii;
                    if (k < n) C[(size_t)f->i * 36 + a * 6 + b] += sij;   /* chain: j = i+1 */
                }
            }
            double gj = 0, gi = 0;
            for (int q = 0; q < 6; ++q) { gj += Jj[q * 6 + a] * r[q]; if (f->i >= 0) gi += Ji[q * 6 + a] * r[q]; }
            gr[(size_t)f->j * 6 + a] += gj;
            if (f->i >= 0) gr[(size_t)f->i * 6 + a] += gi;
        }
        if (k >= n) { sep[f->i] = 1; sep[f->j] = 1; }
    }
    for (int i = 0; i < n; ++i) for (int a = 0; a < 6; ++a) D[(size_t)i * 36 + a * 6 + a] += lambda;
    if (neg_rhs) memcpy(gr, neg_rhs, (size_t)n * 6 * sizeof(double));   /* refinement: solve for another right-hand side */
    sep[0] = 1; sep[n - 1] = 1;
    int ns = 0;
    int* sidx = (int*)malloc(sizeof(int) * n);     /* pose -> separator index or -1 */
    int* spose = (int*)malloc(sizeof(int) * n);
    for (int i = 0; i < n; ++i) { if (sep[i]) { sidx[i] = ns; spose[ns++] = i; } else sidx[i] = -1; }
    int N = ns * 6;
    /* reduced matrix in envelope (skyline) storage: row i holds columns first[i]..i */
    int* first = (int*)malloc(sizeof(int) * N);
    for (int s = 0; s < ns; ++s) for (int a = 0; a < 6; ++a) first[s * 6 + a] = (s > 0 ? (s - 1) * 6 : 0);
    for (int k = n; k < g->nf; ++k) {
        int sa = sidx[g->f[k].i], sb = sidx[g->f[k].j];
        int lo = sa < sb ? sa : sb, hi = sa < sb ? sb : sa;
        for (int a = 0; a < 6; ++a) if (first[hi * 6 + a] > lo * 6) first[hi * 6 + a] = lo * 6;
    }
    const int sparse = g_reduced_solver != NULL;
    size_t* rowp = (size_t*)malloc(sizeof(size_t) * (N + 1));
    rowp[0] = 0;
    for (int i = 0; i < N; ++i) rowp[i + 1] = rowp[i] + (sparse ? 0 : (size_t)(i - first[i] + 1));
    double* S = (double*)calloc(rowp[N] + 1, sizeof(double));
    double* rhs = (double*)calloc(N, sizeof(double));
#define SREF(i, j) S[rowp[i] + (size_t)((j) - first[i])]
    /* eliminate interior chain nodes of every segment (left separator L, right separator R) */
    double* E = (double*)calloc((size_t)n * 36, sizeof(double));   /* E[i] = H(L, i) after fill */
    double* Dl = (double*)malloc((size_t)n * 36 * sizeof(double)); /* Cholesky factors of eliminated D_i */
    double* gi_ = (double*)malloc((size_t)n * 6 * sizeof(double));
    memcpy(gi_, gr, (size_t)n * 6 * sizeof(double));
    double* Dw = (double*)malloc((size_t)n * 36 * sizeof(double));
    memcpy(Dw, D, (size_t)n * 36 * sizeof(double));
    int fail = 0;
    double* SLR = (double*)calloc((size_t)ns * 36, sizeof(double)); /* reduced coupling S(L, next separator) */
    for (int s = 0; s + 1 < ns && !fail; ++s) {
        int L = spose[s], R = spose[s + 1];
        if (R == L + 1) { memcpy(SLR + (size_t)s * 36, C + (size_t)L * 36, 36 * sizeof(double)); continue; }
        memcpy(E + (size_t)(L + 1) * 36, C + (size_t)L * 36, 36 * sizeof(double));
        for (int i = L + 1; i < R; ++i) {
            double* Li = Dl + (size_t)i * 36;
            memcpy(Li, Dw + (size_t)i * 36, 36 * sizeof(double));
            if (chol6(Li)) { fail = 1; break; }
            /* X = D_i^-1 [E_i^T | C_i | g_i] */
            double ET[36], Ci[36], gv[6];
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) ET[a * 6 + b] = E[(size_t)i * 36 + b * 6 + a];
            memcpy(Ci, C + (size_t)i * 36, sizeof Ci);
            memcpy(gv, gi_ + (size_t)i * 6, sizeof gv);
            double XE[36], XC[36], Xg[6];
            memcpy(XE, ET, sizeof XE); chol6_solve(Li, XE, 6);
            memcpy(XC, Ci, sizeof XC); chol6_solve(Li, XC, 6);
            memcpy(Xg, gv, sizeof Xg); chol6_solve(Li, Xg, 1);
            const double* Ei = E + (size_t)i * 36;
            double* DL = Dw + (size_t)L * 36;
            double* Dn = Dw + (size_t)(i + 1) * 36;
            double* En = (i + 1 < R) ? E + (size_t)(i + 1) * 36 : SLR + (size_t)s * 36;
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b < 6; ++b) {
                    double sLL = 0, sLn = 0, snn = 0;
                    for (int q = 0; q < 6; ++q) {
                        sLL += Ei[a * 6 + q] * XE[q * 6 + b];
                        sLn += Ei[a * 6 + q] * XC[q * 6 + b];
                        snn += Ci[q * 6 + a] * XC[q * 6 + b];
                    }
                    DL[a * 6 + b] -= sLL;
                    En[a * 6 + b] -= sLn;
                    Dn[a * 6 + b] -= snn;
                }
                double tL = 0, tn = 0;
                for (int q = 0; q < 6; ++q) { tL += Ei[a * 6 + q] * Xg[q]; tn += Ci[q * 6 + a] * Xg[q]; }
                gi_[(size_t)L * 6 + a] -= tL;
                gi_[(size_t)(i + 1) * 6 + a] -= tn;
            }
        }
    }
    if (!fail && sparse) {
        /* block list: ns diagonal blocks, ns - 1 chain couplings, one block per loop closure */
        int nblk = ns + (ns - 1) + (g->nf - n), q = 0;
        int* bi = (int*)malloc(sizeof(int) * nblk);
        int* bj = (int*)malloc(sizeof(int) * nblk);
        double* blk = (double*)calloc((size_t)nblk * 36, sizeof(double));
        for (int s = 0; s < ns; ++s) {
            int p = spose[s];
            bi[q] = s; bj[q] = s;
            memcpy(blk + (size_t)q * 36, Dw + (size_t)p * 36, 36 * sizeof(double));
            ++q;
            for (int a = 0; a < 6; ++a) rhs[s * 6 + a] = -gi_[(size_t)p * 6 + a];
        }
        for (int s = 0; s + 1 < ns; ++s) {      /* S(s, s+1) = SLR[s]: lower block (s+1, s) is its transpose */
            bi[q] = s + 1; bj[q] = s;
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) blk[(size_t)q * 36 + b * 6 + a] = SLR[(size_t)s * 36 + a * 6 + b];
            ++q;
        }
        for (int k = n; k < g->nf; ++k) {
            const pg_factor* f = &g->f[k];
            const double* Ji = g->Ji + (size_t)k * 36;
            const double* Jj = g->Jj + (size_t)k * 36;
            int sa = sidx[f->i], sb = sidx[f->j];
            if (sa == sb) continue;
            bi[q] = sa > sb ? sa : sb; bj[q] = sa > sb ? sb : sa;
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) {
                double sij = 0;
                for (int t = 0; t < 6; ++t) sij += Ji[t * 6 + a] * Jj[t * 6 + b];
                if (sa < sb) blk[(size_t)q * 36 + b * 6 + a] = sij; else blk[(size_t)q * 36 + a * 6 + b] = sij;
            }
            ++q;
        }
        if (g_reduced_solver(ns, q, bi, bj, blk, rhs) != 0) fail = 1;
        free(bi); free(bj); free(blk);
    }
    if (!fail && !sparse) {
        for (int s = 0; s < ns; ++s) {
            int p = spose[s];
            for (int a = 0; a < 6; ++a) {
                for (int b = 0; b <= a; ++b) SREF(s * 6 + a, s * 6 + b) += Dw[(size_t)p * 36 + a * 6 + b];
                rhs[s * 6 + a] = -gi_[(size_t)p * 6 + a];
            }
            if (s + 1 < ns)   /* S(s, s+1) stored in row block s+1, column block s */
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b)
                    SREF((s + 1) * 6 + b, s * 6 + a) += SLR[(size_t)s * 36 + a * 6 + b];
        }
        for (int k = n; k < g->nf; ++k) {
            const pg_factor* f = &g->f[k];
            const double* Ji = g->Ji + (size_t)k * 36;
            const double* Jj = g->Jj + (size_t)k * 36;
            int sa = sidx[f->i], sb = sidx[f->j];
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) {
                double sij = 0;
                for (int q = 0; q < 6; ++q) sij += Ji[q * 6 + a] * Jj[q * 6 + b];
                if (sa < sb) SREF(sb * 6 + b, sa * 6 + a) += sij;
                else if (sa > sb) SREF(sa * 6 + a, sb * 6 + b) += sij;
            }
        }
        /* envelope Cholesky */
        for (int i = 0; i < N && !fail; ++i) {
            for (int j = first[i]; j <= i; ++j) {
                int k0 = first[i] > first[j] ? first[i] : first[j];
                double s = SREF(i, j);
                for (int k = k0; k < j; ++k) s -= SREF(i, k) * SREF(j, k);
                if (j < i) SREF(i, j) = s / SREF(j, j);
                else { if (!(s > 0) || !isfinite(s)) { fail = 1; break; } SREF(i, i) = sqrt(s); }
            }
        }
    }
    if (!fail) {
        if (!sparse) {
            for (int i = 0; i < N; ++i) { double s = rhs[i]; for (int k = first[i]; k < i; ++k) s -= SREF(i, k) * rhs[k]; rhs[i] = s / SREF(i, i); }
            for (int i = N - 1; i >= 0; --i) {
                rhs[i] /= SREF(i, i);
                for (int k = first[i]; k < i; ++k) rhs[k] -= SREF(i, k) * rhs[i];
            }
        }
        for (int s = 0; s < ns; ++s) memcpy(delta + (size_t)spose[s] * 6, rhs + s * 6, 6 * sizeof(double));
        /* back-substitute the interiors, right to left */
        for (int s = ns - 2; s >= 0; --s) {
            int L = spose[s], R = spose[s + 1];
            for (int i = R - 1; i > L; --i) {
                double b[6];
                for (int a = 0; a < 6; ++a) {
                    double t = -gi_[(size_t)i * 6 + a];
                    for (int q = 0; q < 6; ++q) {
                        t -= E[(size_t)i * 36 + q * 6 + a] * delta[(size_t)L * 6 + q];
                        t -= C[(size_t)i * 36 + a * 6 + q] * delta[(size_t)(i + 1) * 6 + q];
                    }
                    b[a] = t;
                }
                chol6_solve(Dl + (size_t)i * 36, b, 1);
                memcpy(delta + (size_t)i * 6, b, sizeof b);
            }
        }
    }
#undef SREF
    free(D); free(C); free(gr); free(sep); free(sidx); free(spose); free(first); free(rowp); free(S); free(rhs);
    free(E); free(Dl); free(gi_); free(Dw); free(SLR);
    (void)nlc;
    return fail ? -1 : 0;
}

int orc_pg_solve(const double* dr, int total, const orc_lc_edge* edges, int ne, const orc_pg_params* p,
                 double* out12, double* stats)
{
    const double PI = ORC_PI_REF;
    /* odometry sigmas (optimizer.cpp:24,28) */
    const double wgt1 = 0.001, wgt2 = 10;
    const double sig_odo[6] = { wgt1 * PI / 180, wgt1 * PI / 180, 0.1 * wgt1 * wgt2 * PI / 180, wgt1 * wgt2, wgt1 * wgt2, wgt1 };
    int n = total;
    pg_t g;
    g.n = n; g.nf = n + ne;
    g.f = (pg_factor*)malloc(sizeof(pg_factor) * g.nf);
    g.r = (double*)malloc(sizeof(double) * 6 * g.nf);
    g.Ji = (double*)malloc(sizeof(double) * 36 * g.nf);
    g.Jj = (double*)malloc(sizeof(double) * 36 * g.nf);
    orc_pose* DR = (orc_pose*)malloc(sizeof(orc_pose) * n);
    orc_pose* X = (orc_pose*)malloc(sizeof(orc_pose) * n);
    orc_pose* Xn = (orc_pose*)malloc(sizeof(orc_pose) * n);
    for (int i = 0; i < n; ++i) orc_pose_from_rodrigues(dr + (size_t)i * 6, &DR[i]);
    /* initial estimate = DR o noise, six normal draws per pose in ping order (:154-160) */
    if (p->add_noise) {
        double* z = (double*)malloc(sizeof(double) * 6 * (size_t)n);
        orc_normal_fill(z, 6 * n);
        const double noise_xyz = 0.5, noise_rpy = 0.5 * PI / 180;
        for (int i = 0; i < n; ++i) {
            double w[3] = { z[6 * i] * noise_rpy, z[6 * i + 1] * noise_rpy, z[6 * i + 2] * noise_rpy };
            orc_pose N;
            orc_so3_exp(w, N.R);
            N.t[0] = z[6 * i + 3] * noise_xyz; N.t[1] = z[6 * i + 4] * noise_xyz; N.t[2] = z[6 * i + 5] * noise_xyz;
            orc_pose_compose(&DR[i], &N, &X[i]);
        }
        free(z);
    } else memcpy(X, DR, sizeof(orc_pose) * n);
    /* factors */
    g.f[0].i = -1; g.f[0].j = 0; g.f[0].meas = DR[0];
    for (int k = 0; k < 6; ++k) g.f[0].w[k] = 1.0 / 0.000001;
    for (int i = 1; i < n; ++i) {
        g.f[i].i = i - 1; g.f[i].j = i;
        orc_pose_between(&DR[i - 1], &DR[i], &g.f[i].meas);
        for (int k = 0; k < 6; ++k) g.f[i].w[k] = 1.0 / sig_odo[k];
    }
    for (int e = 0; e < ne; ++e) {
        pg_factor* f = &g.f[n + e];
        f->i = edges[e].a; f->j = edges[e].b;
        memcpy(f->meas.R, edges[e].rel, sizeof(double) * 9);
        memcpy(f->meas.t, edges[e].rel + 9, sizeof(double) * 3);
        for (int k = 0; k < 6; ++k) f->w[k] = 1.0 / sqrt(edges[e].var[k]);
    }
    /* LM (same loop as the mini problems) */
    double lambda = p->lambda0;
    int iters = 0;
    double err = pg_error(&g, X);
    double err0 = err, cur;
    double* delta = (double*)malloc(sizeof(double) * 6 * (size_t)n);
    if (err > 0 && p->max_iters > 0) do {
        cur = err;
        for (int k = 0; k < g.nf; ++k)
            pg_factor_eval(&g.f[k], X, g.r + (size_t)k * 6, g.Ji + (size_t)k * 36, g.Jj + (size_t)k * 36);
        double oldLin = 0;
        for (size_t k = 0; k < (size_t)g.nf * 6; ++k) oldLin += g.r[k] * g.r[k];
        oldLin *= 0.5;
        for (;;) {
            int ok = pg_solve(&g, lambda, delta, NULL) == 0;
            for (int it = 0; ok && it < g_full_refine; ++it) {
                /* rho = -J^T (r + J delta) - lambda delta, in long double; correction c: (H + lambda) c = rho */
                long double* rho = (long double*)calloc((size_t)n * 6, sizeof(long double));
                for (int k = 0; k < g.nf; ++k) {
                    const pg_factor* f = &g.f[k];
                    long double v[6];
                    for (int a = 0; a < 6; ++a) {
                        long double t = g.r[(size_t)k * 6 + a];
                        for (int q = 0; q < 6; ++q) {
                            t += (long double)g.Jj[(size_t)k * 36 + a * 6 + q] * delta[(size_t)f->j * 6 + q];
                            if (f->i >= 0) t += (long double)g.Ji[(size_t)k * 36 + a * 6 + q] * delta[(size_t)f->i * 6 + q];
                        }
                        v[a] = t;
                    }
                    for (int q = 0; q < 6; ++q) for (int a = 0; a < 6; ++a) {
                        rho[(size_t)f->j * 6 + q] -= (long double)g.Jj[(size_t)k * 36 + a * 6 + q] * v[a];
                        if (f->i >= 0) rho[(size_t)f->i * 6 + q] -= (long double)g.Ji[(size_t)k * 36 + a * 6 + q] * v[a];
                    }
                }
                double* nr = (double*)malloc(sizeof(double) * 6 * (size_t)n);
                double* c = (double*)malloc(sizeof(double) * 6 * (size_t)n);
                for (size_t q = 0; q < (size_t)n * 6; ++q) nr[q] = -(double)(rho[q] - (long double)lambda * delta[q]);
                ok = pg_solve(&g, lambda, c, nr) == 0;
                if (ok) for (size_t q = 0; q < (size_t)n * 6; ++q) delta[q] += c[q];
                free(rho); free(nr); free(c);
            }
            int success = 0, stop = 0;
            double newErr = 0;
            if (ok) {
                double newLin = 0;
                for (int k = 0; k < g.nf; ++k) {
                    const pg_factor* f = &g.f[k];
                    for (int a = 0; a < 6; ++a) {
                        double s = g.r[(size_t)k * 6 + a];
                        for (int q = 0; q < 6; ++q) {
                            s += g.Jj[(size_t)k * 36 + a * 6 + q] * delta[(size_t)f->j * 6 + q];
                            if (f->i >= 0) s += g.Ji[(size_t)k * 36 + a * 6 + q] * delta[(size_t)f->i * 6 + q];
                        }
                        newLin += s * s;
                    }
                }
                newLin *= 0.5;
                double linChange = oldLin - newLin;
                if (linChange >= 0) {
                    for (int i = 0; i < n; ++i) orc_pose_retract(&X[i], delta + (size_t)i * 6, &Xn[i]);
                    newErr = pg_error(&g, Xn);
                    double costChange = err - newErr;
                    if (linChange > 2.220446049250313e-16 * oldLin) success = (costChange / linChange) > p->min_fidelity;
                    if (fabs(costChange) < p->rel_tol * err) stop = 1;
                }
            }
            if (success) { memcpy(X, Xn, sizeof(orc_pose) * n); err = newErr; lambda /= p->lambda_factor; ++iters; break; }
            else if (!stop) { lambda *= p->lambda_factor; if (lambda >= p->lambda_max) break; }
            else break;
        }
    } while (iters < p->max_iters && !((err <= 0) || ((cur - err) / cur <= p->rel_tol) || ((cur - err) <= p->abs_tol)) && isfinite(cur));
    for (int i = 0; i < n; ++i) {
        memcpy(out12 + (size_t)i * 12, X[i].R, sizeof(double) * 9);
        memcpy(out12 + (size_t)i * 12 + 9, X[i].t, sizeof(double) * 3);
    }
    if (stats) { stats[0] = iters; stats[1] = err0; stats[2] = err; stats[3] = lambda; }
    free(delta); free(g.f); free(g.r); free(g.Ji); free(g.Jj); free(DR); free(X); free(Xn);
    return iters;
}

/* The LM objective 0.5 sum |r|^2 of the graph orc_pg_solve builds from (dr, edges), evaluated at the poses `x12` (total x 12,
 * R row-major then t) instead of being minimised: the full-size check of a solver's answer where running the LM here would
 * take hours (the error function is the one above: optimizer.cpp:134-199 factors, GTSAM Pose3 local coordinates). */
double orc_pg_error_at(const double* dr, int total, const orc_lc_edge* edges, int ne, const double* x12)
{
    const double PI = ORC_PI_REF;
    const double wgt1 = 0.001, wgt2 = 10;
    const double sig_odo[6] = { wgt1 * PI / 180, wgt1 * PI / 180, 0.1 * wgt1 * wgt2 * PI / 180, wgt1 * wgt2, wgt1 * wgt2, wgt1 };
    int n = total;
    pg_t g;
    g.n = n; g.nf = n + ne; g.r = NULL; g.Ji = NULL; g.Jj = NULL;
    g.f = (pg_factor*)malloc(sizeof(pg_factor) * g.nf);
    orc_pose* DR = (orc_pose*)malloc(sizeof(orc_pose) * n);
    orc_pose* X = (orc_pose*)malloc(sizeof(orc_pose) * n);
    for (int i = 0; i < n; ++i) {
        orc_pose_from_rodrigues(dr + (size_t)i * 6, &DR[i]);
        memcpy(X[i].R, x12 + (size_t)i * 12, sizeof(double) * 9);
        memcpy(X[i].t, x12 + (size_t)i * 12 + 9, sizeof(double) * 3);
    }
    g.f[0].i = -1; g.f[0].j = 0; g.f[0].meas = DR[0];
    for (int k = 0; k < 6; ++k) g.f[0].w[k] = 1.0 / 0.000001;
    for (int i = 1; i < n; ++i) {
        g.f[i].i = i - 1; g.f[i].j = i;
        orc_pose_between(&DR[i - 1], &DR[i], &g.f[i].meas);
        for (int k = 0; k < 6; ++k) g.f[i].w[k] = 1.0 / sig_odo[k];
    }
    for (int e = 0; e < ne; ++e) {
        pg_factor* f = &g.f[n + e];
        f->i = edges[e].a; f->j = edges[e].b;
        memcpy(f->meas.R, edges[e].rel, sizeof(double) * 9);
        memcpy(f->meas.t, edges[e].rel + 9, sizeof(double) * 3);
        for (int k = 0; k < 6; ++k) f->w[k] = 1.0 / sqrt(edges[e].var[k]);
    }
    double err = pg_error(&g, X);
    free(g.f); free(DR); free(X);
    return err;
}
