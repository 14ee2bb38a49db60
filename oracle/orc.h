/*
 * oracle/orc.h -- CPU ORACLE for the diasss hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C, single-threaded restatement of the reference algorithms
 * (halajun/diasss: ORB extraction -> geo-gated brute-force matching -> sonar
 * reprojection -> mini-LM loop-closure measurements -> pose-graph LM).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (diasss_amd/) never links, imports or calls anything in here.
 *
 * PARITY UNPINNED at the OpenCV / GTSAM boundary: the reference ships no tests,
 * golden vectors or data, and OpenCV/GTSAM/Eigen/Boost are absent from this
 * image, so the reference cannot be built (oracle/_ref is therefore empty).
 * The out-of-tree primitives (cv::FAST, cv::resize, fastAtan2, cv::RNG,
 * GTSAM Pose3/LM ...) are restated from their published algorithms and pinned
 * by first-principles known-answer tests (tests/test_oracle_*.py).
 *
 * Documented deviations where the reference is undefined:
 *   - descriptors are the ORB/rBRIEF configuration (the SIFT call site writes
 *     nothing; ORBextractor.cpp:1097 vs :1098);
 *   - matcher uses the Hamming branch (FEAmatcher.cpp:141-176) by default, the
 *     L2-on-bytes branch (:106-139) is selectable;
 *   - DistributeOctTree: nIni = max(1, round(w/h)) (ORBextractor.cpp:543 is a
 *     division by zero for tall frames); sort ties broken by node creation
 *     order instead of heap address (:684);
 *   - port-side column 0 reads g_range[M/2] one past the end (frame.cpp:148);
 *     here the index is clamped to M/2-1;
 *   - mask eraser (frame.cpp:100-102) is clipped to the image;
 *   - empty first-stage match list: SCC is skipped (FEAmatcher.cpp:201 is UB);
 *     ConsistentCheck with an empty scc history takes the "keep the larger
 *     direction" branch (:344 is UB);
 *   - cos/sin use a deterministic fdlibm-style polynomial (no FMA) so that the
 *     oracle and the HIP kernels agree bit for bit; mean uses a fixed
 *     summation order (OpenCV leaves it unspecified);
 *   - GaussianBlur 13x13 sigma 2: 8.8 fixed-point taps by OpenCV's error-diffusion
 *     rule (getGaussianKernelFixedPoint_ED, restated; not checkable against an
 *     OpenCV build here).
 */
#ifndef ORC_H
#define ORC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_PI_REF 3.14159265359 /* the reference's PI macro: frame.cpp:16, FEAmatcher.cpp:11, optimizer.cpp:19 */

/* ---------------------------------------------------------------- math */
int      orc_cvround(double v);                 /* cvRound: round half to even */
int      orc_cvroundf(float v);
int      orc_cvfloorf(float v);
void     orc_sincos(double a, double* s, double* c); /* deterministic, no FMA */
float    orc_fast_atan2(float y, float x);      /* cv::fastAtan2, degrees */
uint32_t orc_cvrng_next(uint64_t* state);       /* cv::RNG MWC */
int      orc_cvrng_uniform(uint64_t* state, int a, int b);
void     orc_normal_fill(double* out, int n);   /* default_random_engine + normal_distribution<double>(0,1) */
int      orc_hamming256(const uint8_t* a, const uint8_t* b); /* FEAmatcher.cpp:442-458 (SWAR form) */
int      orc_l2sq32(const uint8_t* a, const uint8_t* b);
int      orc_l2sq(const uint8_t* a, const uint8_t* b, int n);

/* ---------------------------------------------------------------- frame preprocessing (frame.cpp) */
typedef struct {
    double factor;   /* 2.5  frame.cpp:59,85 */
    int width;       /* 10   frame.cpp:86 centre line half width */
    int r;           /* 6    eraser radius */
    int side;        /* 150  first/last pings; columns use side*0.6 */
} orc_mask_params;
void  orc_mask_params_default(orc_mask_params* p);

double orc_mean(const double* raw, int N, int M);                 /* fixed summation order */
void  orc_normalize(const double* raw, int N, int M, uint8_t* out);/* frame.cpp:57-81 */
void  orc_mask(const double* raw, int N, int M, const orc_mask_params* p, uint8_t* mask); /* frame.cpp:83-124 */
void  orc_geo_at(const double* pose6, const double* gr, int N, int M, int row, int col,
                 double* x, double* y);                            /* frame.cpp:126-165, one bin */
void  orc_geo_img(const double* pose6, const double* gr, int N, int M, double* gx, double* gy);
void  orc_geo_bbox(const double* pose6, const double* gr, int N, int M, double bbox[4]); /* xmin,xmax,ymin,ymax */
float orc_overlap(const double bs[4], const double bt[4]);         /* util.cpp:13-43 */

/* ---------------------------------------------------------------- ORB (thirdparty/ORBextractor.cpp) */
typedef struct { float x, y, size, angle, response; int octave; } orc_kp;

typedef struct {
    int nfeatures;     /* 2000 frame.cpp:180 */
    float scale;       /* 1.2f */
    int nlevels;       /* 6 */
    int ini_th;        /* 12 */
    int min_th;        /* 7 */
} orc_orb_params;
void orc_orb_params_default(orc_orb_params* p);

#define ORC_MAX_LEVELS 8
void orc_orb_level_sizes(int rows, int cols, const orc_orb_params* p, int* lrows, int* lcols);
void orc_orb_level_quota(const orc_orb_params* p, int* quota);
void orc_orb_umax(int* umax16);
void orc_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw); /* cv::resize INTER_LINEAR */
/* FAST-9/16 "arc score" A of every pixel of a window (0 where not evaluable) */
void orc_fast_arc_map(const uint8_t* img, int stride, int h, int w, int* A);
/* cv::FAST(window, thr, nonmax=true): returns count, (x,y,score) row-major */
int  orc_fast_window(const uint8_t* img, int stride, int h, int w, int thr, int* xs, int* ys, int* sc, int cap);
/* per-level candidates in reference order (ORBextractor.cpp:789-829), coords relative to minBorder */
int  orc_fast_level(const uint8_t* img, int rows, int cols, int ini_th, int min_th,
                    float* xs, float* ys, float* resp, int cap);
/* DistributeOctTree (ORBextractor.cpp:539-763); returns kept count and indices into the candidate list */
int  orc_quadtree(const float* xs, const float* ys, const float* resp, int n,
                  int minX, int maxX, int minY, int maxY, int N, int* keep_idx);
float orc_ic_angle(const uint8_t* img, int stride, int x, int y);   /* ORBextractor.cpp:77-104 */
void orc_gauss13_taps(int taps[13]);
void orc_blur13(const uint8_t* src, int rows, int cols, uint8_t* dst);
void orc_brief(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t desc[32]); /* :108-147 */
/* whole extractor: operator() with the ORB descriptor call (ORBextractor.cpp:1049-1113) */
int  orc_orb_extract(const uint8_t* img, int rows, int cols, const orc_orb_params* p,
                     orc_kp* kps, uint8_t* desc, int cap);
/* Frame::DetectFeature mask filter (frame.cpp:184-195) applied in place; returns new count */
int  orc_mask_filter(orc_kp* kps, uint8_t* desc, int n, const uint8_t* mask, int cols);
int  orc_mask_filter2(orc_kp* kps, uint8_t* desc, uint8_t* desc128 /* may be NULL */, int n, const uint8_t* mask, int cols);
/* N4: the descriptor of the SIFT call site (ORBextractor.cpp:1043-1047,1098) as intended -- oracle/orc_sift.c */
void orc_sift_weights(float* w, int n);                                /* exp(-k / 512), k < n */
void orc_sift_hist(const uint8_t* blurred, int rows, int cols, int x, int y, float angle_deg, int32_t h[128]);  /* 2^-12 fixed point */
void orc_sift_finalize(const int32_t h[128], uint8_t out[128]);        /* normalise, clip 0.2, renormalise, x 512, saturate */
void orc_sift128(const uint8_t* blurred, int rows, int cols, int x, int y, float angle_deg, uint8_t out[128]);
int  orc_orb_extract_sift(const uint8_t* img, int rows, int cols, const orc_orb_params* p,
                          orc_kp* kps, uint8_t* desc, uint8_t* desc128, int cap);

/* ---------------------------------------------------------------- matcher (FEAmatcher.cpp) */
typedef struct {
    int use_l2;        /* 0: Hamming branch (:141-176); 1: L2 branch (:106-139, USE_SIFT) on the 32 ORB bytes the shipped
                          reference feeds it; 2: the same branch on 128-element SIFT rows (desc = n x 128, oracle/orc_sift.c) */
    double radius;     /* 8  :66 */
    int bound_same;    /* 88 :143 */
    int bound_diff;    /* 80 :145 (frame-id parities differ) */
    double l2_bound;   /* 350 :108 */
    double ratio;      /* 0.35 :110,:147 */
    int scc_iters;     /* 1000 :189 */
    double pix_err;    /* 2.5 :190 */
    double merge_thr;  /* 2.5 :329 */
} orc_match_params;
void orc_match_params_default(orc_match_params* p);

/* first stage (:79-183): CorresID (size na), returns number of accepted (== |ID_loc|) */
int orc_match_nn(int id, int id_ref, const orc_kp* kps, const uint8_t* desc, const double* geo /*na x 2*/, int na,
                 const orc_kp* kps_ref, const uint8_t* desc_ref, const double* geo_ref, int nb,
                 const double bbox_ref[4], const orc_match_params* p, int32_t* corres,
                 int32_t* best_d, int32_t* second_d, int32_t* ncand);
/* SCC_x (:186-248): filters corres in place; best (count, model) -> scc_best; returns history length */
int orc_match_scc(int id, int id_ref, int rows_ref, const orc_kp* kps, int na, const orc_kp* kps_ref,
                  const orc_match_params* p, int32_t* corres, int* scc_count, double* scc_model);
/* GeoNearNeighSearch = nn + scc */
int orc_match_dir(int id, int id_ref, int rows_ref, const orc_kp* kps, const uint8_t* desc, const double* geo, int na,
                  const orc_kp* kps_ref, const uint8_t* desc_ref, const double* geo_ref, int nb,
                  const double bbox_ref[4], const orc_match_params* p, int32_t* corres,
                  int* scc_count, double* scc_model);
/* ConsistentCheck (:323-405): outputs index pairs; returns count */
int orc_consistent_check(int id_s, int id_t, int rows_s, int rows_t,
                         const int32_t* c1, int n1, const int32_t* c2, int n2,
                         int hist1, int cnt1, double model1, int hist2, int cnt2, double model2,
                         const orc_match_params* p, int32_t* src_idx, int32_t* tgt_idx);
/* RobustMatching (:13-50): rows [id_s,id_t,y_s,x_s,y_t,x_t] for the source frame; returns count */
int orc_robust_matching(int id_s, int id_t, int rows_s, int rows_t,
                        const orc_kp* kps_s, const uint8_t* desc_s, const double* geo_s, int ns, const double bbox_s[4],
                        const orc_kp* kps_t, const uint8_t* desc_t, const double* geo_t, int nt, const double bbox_t[4],
                        const orc_match_params* p, double* rows6, int cap);

/* ---------------------------------------------------------------- geometry (GTSAM Pose3 semantics) */
typedef struct { double R[9]; double t[3]; } orc_pose;   /* R row-major */
void orc_so3_exp(const double w[3], double R[9]);
void orc_so3_log(const double R[9], double w[3]);
void orc_pose_from_rodrigues(const double p6[6], orc_pose* T); /* Pose3(Rot3::Rodrigues(r,p,y), Point3(x,y,z)) */
void orc_pose_compose(const orc_pose* A, const orc_pose* B, orc_pose* C);
void orc_pose_inverse(const orc_pose* A, orc_pose* B);
void orc_pose_between(const orc_pose* A, const orc_pose* B, orc_pose* C);
void orc_pose_exp(const double xi[6], orc_pose* T);
void orc_pose_log(const orc_pose* T, double xi[6]);
void orc_pose_adjoint(const orc_pose* T, double Ad[36]);
void orc_pose_retract(const orc_pose* T, const double xi[6], orc_pose* out);
void orc_pose_rpy(const orc_pose* T, double rpy[3]);
/* SssPointFactor::evaluateError (SSSpointfactor.cpp:11-80), unwhitened */
void orc_sss_factor(const double p[3], const orc_pose* T, const orc_pose* Ts, double mx, double my,
                    double e[2], double H1[6], double H2[12]);

/* ---------------------------------------------------------------- reprojection + loop-closure measurements (optimizer.cpp) */
/* GetKpsPairs non-anno branch (:575-639): rows6 -> Vector7; returns count */
int orc_get_kps_pairs(const double* rows6, int nrows, int id_t, const double* alt_s, const double* gr_s, int ngr_s,
                      const double* alt_t, const double* gr_t, int ngr_t, double* kp7, int cap);
typedef struct { double rel[12]; double var[6]; double score; int iters; double err0, err1; } orc_lc;
/* LoopClosingTFs (:641-982), graph_option=0 branch. pose6_* are N x 6 DR poses. */
int orc_lc_solve(const double* kp7, int n, const double* pose6_s, const double* alt_s, const double* gr_s, int Ns, int Ms,
                 const double* pose6_t, const double* alt_t, const double* gr_t, int Nt, int Mt, orc_lc* out);

/* ---------------------------------------------------------------- pose graph (optimizer.cpp:101-279, batch LM replaces iSAM2) */
/* LMTriaFactor + TriangulateOneLandmark (LMtriangulatefactor.cpp:10-27, optimizer.cpp:984-1021, call site :907-921) */
int orc_triangulate_one(const double kp7[7], const orc_pose* Ts_s, const orc_pose* Ts_t, const orc_pose* Tp_s, const orc_pose* Tp_t,
                        const double lm_ini[3], double out[3]);
int orc_triangulate(const double* kp7, int n, const double* pose6_s, const double* alt_s, const double* gr_s, int Ns, int Ms,
                    const double* pose6_t, const double* alt_t, const double* gr_t, int Nt, int Mt, double* out7);
typedef struct { int a, b; double rel[12]; double var[6]; } orc_lc_edge; /* BetweenFactor(X_a, X_b, rel, Variances(var)) */
typedef struct {
    int max_iters; double rel_tol, abs_tol; double lambda0, lambda_factor, lambda_max; double min_fidelity;
    int add_noise;   /* 1: initial = DR o noise (optimizer.cpp:154-160) */
} orc_pg_params;
void orc_pg_params_default(orc_pg_params* p);
/* LC selection (optimizer.cpp:203-258): "last pair wins, first kp in it", score>0 */
int orc_pg_select_lc(int F, const int* frame_rows, int npairs, const int* pair_s, const int* pair_t,
                     const int* pair_off /*npairs+1*/, const double* kp7, const orc_lc* lcs,
                     orc_lc_edge* edges, int cap);
/* batch LM over all pings; dr = total x 6 (roll,pitch,yaw,x,y,z); out poses total x 12 (R row-major, t) */
int orc_pg_solve(const double* dr, int total, const orc_lc_edge* edges, int ne, const orc_pg_params* p,
                 double* out12, double* stats /* [iters, err0, err1, lambda] */);
/* the same graph's LM objective evaluated at given poses (total x 12), for full-size checks of a solver's answer */
/* optional second linear solver of the reduced (separator) system of orc_pg_solve: see orc_posegraph.c */
typedef int (*orc_pg_reduced_solver_fn)(int ns, int nblk, const int* bi, const int* bj, const double* blk36, double* rhs);
void orc_pg_set_reduced_solver(orc_pg_reduced_solver_fn fn);
void orc_pg_set_full_refine(int steps);   /* iterative refinement of every trial's solve against the full system (long double residual) */
double orc_pg_error_at(const double* dr, int total, const orc_lc_edge* edges, int ne, const double* x12);

#ifdef __cplusplus
}
#endif
#endif
