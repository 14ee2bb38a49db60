/*
 * include/dsss.h -- C ABI of the MI355X-native diasss hot path (libdsss.so).
 *
 * The reference (halajun/diasss) has no plugin/FFI layer: its boundary is the C++ API that
 * src/diasss2.cpp compiles against (Frame, FEAmatcher, Optimizer, Util).  The host-side mirror of those
 * classes lives in diasss_amd/host/ and calls ONLY the entry points declared here; every entry point cites
 * the reference interface it replaces (path:line under /root/reference).
 *
 * Conventions
 *   - plain C: opaque context, plain pointers and sizes, no C++/torch types;
 *   - return 0 (DSSS_OK) or a negative DSSS_E_* code, never throws; dsss_last_error() has the detail;
 *   - the context owns all device memory and HIP streams; one context per (host thread, device);
 *   - every input pointer may be a HOST or a DEVICE pointer (copied with hipMemcpyDefault);
 *     output pointers named *_host are host memory;
 *   - frames are identified by their img_id (0..F-1), which is also the index diasss2.cpp:84 uses;
 *   - there is NO CPU fallback: without a HIP device dsss_create fails with DSSS_E_NODEVICE.
 */
#ifndef DSSS_H
#define DSSS_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSSS_OK            0
#define DSSS_E_NODEVICE   -1   /* no HIP device / hipSetDevice failed */
#define DSSS_E_ARG        -2   /* bad argument (null, range, size) */
#define DSSS_E_HIP        -3   /* a HIP runtime call failed */
#define DSSS_E_STATE      -4   /* call order violated (e.g. match before features exist) */
#define DSSS_E_CAPACITY   -5   /* a caller buffer or an internal fixed-capacity buffer is too small */
#define DSSS_E_NUMERIC    -6   /* linear system not positive definite / non-finite value */
#define DSSS_E_COMM       -7   /* RCCL not available / a collective failed */

typedef struct dsss_ctx dsss_ctx;

/* cv::KeyPoint subset that the path touches (ORBextractor.cpp:840-847,1103-1109) */
typedef struct { float x, y, size, angle, response; int32_t octave; } dsss_kp;

/* hard-coded locals of the reference gathered per stage ("Config / flags") */
typedef struct {                 /* frame.cpp:59,85-86 */
    double factor;  int32_t width, r, side;
} dsss_mask_params;
#define DSSS_DESC_ORB      0      /* rotated BRIEF, 256 bits (ORBextractor.cpp:1034-1041,1097: the configuration north_star names) */
#define DSSS_DESC_SIFT128  1      /* ... AND the 128-float descriptor of the live SIFT call site (ORBextractor.cpp:1043-1047,1098) as its
                                     author evidently meant it: 4 x 4 x 8 gradient-orientation histogram on the blurred level image the
                                     keypoint was found on, at its IC angle (definition: oracle/orc_sift.c)            */
typedef struct {                 /* frame.cpp:180: ORBextractor(2000, 1.2, 6, 12, 7) */
    int32_t nfeatures; float scale; int32_t nlevels, ini_th, min_th;
    int32_t descriptor;          /* DSSS_DESC_* (default DSSS_DESC_ORB) */
} dsss_orb_params;
typedef struct {                 /* FEAmatcher.cpp:63-66,108-110,143-147,189-190,329 */
    /* use_l2: 0 = the Hamming branch (:141-176) on the ORB rows; 1 = the L2 branch (:106-139, USE_SIFT) on the 32 ORB bytes the shipped
     * reference feeds it; 2 = the L2 branch on the 128-float rows of DSSS_DESC_SIFT128 (needs features extracted in that mode) */
    int32_t use_l2; double radius; int32_t bound_same, bound_diff; double l2_bound, ratio;
    int32_t scc_iters; double pix_err, merge_thr;
} dsss_match_params;
typedef struct {                 /* GTSAM LevenbergMarquardtParams() defaults + optimizer.cpp:154-160 */
    int32_t max_iters; double rel_tol, abs_tol, lambda0, lambda_factor, lambda_max, min_fidelity;
    int32_t add_noise;
} dsss_pg_params;
typedef struct { double rel[12]; double var[6]; double score; int32_t iters; int32_t pad_; double err0, err1; } dsss_lc;
typedef struct { int32_t a, b; double rel[12]; double var[6]; } dsss_lc_edge;

void dsss_mask_params_default(dsss_mask_params*);
void dsss_orb_params_default(dsss_orb_params*);
void dsss_match_params_default(dsss_match_params*);
void dsss_pg_params_default(dsss_pg_params*);

/* ------------------------------------------------------------------ context */
int  dsss_create(int device, int max_frames, dsss_ctx** out);
void dsss_destroy(dsss_ctx*);
const char* dsss_strerror(int code);
const char* dsss_last_error(const dsss_ctx*);
int  dsss_sync(dsss_ctx*);                       /* hipStreamSynchronize of the context stream */
void* dsss_stream(dsss_ctx*);                    /* hipStream_t, for event timing by the caller */
int  dsss_set_params(dsss_ctx*, const dsss_mask_params*, const dsss_orb_params*, const dsss_match_params*,
                     const dsss_pg_params*);     /* NULL keeps the current (default = reference) values */

/* ------------------------------------------------------------------ ranks (new: the reference is one process, one thread)
 * One process per GPU.  A context that joined a communicator shards the pose-graph solve of dsss_posegraph_solve[_edges]:
 * contiguous blocks of frames (poses) per rank, every rank eliminates its own block, and ONE all-reduce per LM trial sums the
 * reduced Hessian on the interface poses ([interface blocks | Schur complements | gradient]) before the small replicated
 * interface solve.  Every rank must make the same calls with the same frames and the same (all-gathered) LC edges.
 * dsss_comm_init: RCCL (ncclCommInitRank with the 128-byte id of dsss_comm_unique_id, distributed by the caller's launcher);
 * dsss_comm_init_callback: the caller sums a host buffer in place (tests on a one-GPU box, e.g. over gloo).               */
/* op 0: in-place sum over the ranks of n doubles in host_buf.  op 1: all-gather, host_buf holds world x n bytes, the caller's
 * own n bytes are in place at rank * n, the others are to be filled in.  Return 0 on success.                          */
typedef int (*dsss_comm_fn)(void* user, int op, void* host_buf, size_t n);
int dsss_comm_unique_id(void* id128_out);
int dsss_comm_init(dsss_ctx*, const void* id128, int rank, int world);
int dsss_comm_init_callback(dsss_ctx*, int rank, int world, dsss_comm_fn fn, void* user);
/* The same two operations on the DEVICE buffer itself, to be ordered on `stream` (a hipStream_t): for callers whose transport moves
 * device memory (another collective library), and for timing ONE rank of an N-rank job on a single GPU by replaying the sums a
 * lock-step run of all ranks recorded (tools/emulate_ranks.py, bench.py --emulate-rank).  op 0: n doubles at dev_buf, summed in
 * place; op 1: world x n bytes at dev_buf, own slice in place.                                                              */
typedef int (*dsss_comm_dev_fn)(void* user, int op, void* dev_buf, size_t n, void* stream);
int dsss_comm_init_device_callback(dsss_ctx*, int rank, int world, dsss_comm_dev_fn fn, void* user);
int dsss_comm_destroy(dsss_ctx*);
int dsss_comm_stats(dsss_ctx*, int* rank, int* world, double* allreduce_bytes, int64_t* allreduce_calls);
/* frames are owned in contiguous blocks: rank r owns frames [nframes r / world, nframes (r+1) / world)                     */
int dsss_comm_frame_owner(const dsss_ctx*, int nframes, int frame);
/* after every rank extracted the frames it owns: exchange the per-frame feature records (C1 of SURVEY.md 2a, one all-gather)
 * so that every rank holds the features of all nframes frames                                                             */
int dsss_features_allgather(dsss_ctx*, int nframes);
/* number of pose-graph partitions (default 0 = one per rank); more partitions than ranks only exercise the interface path */
int dsss_set_pg_partitions(dsss_ctx*, int nparts);

/* ------------------------------------------------------------------ Frame (frame.h:19-20, frame.cpp:18-55)
 * Frame::Frame(id, img CV_64F NxM, pose CV_64F Nx6 [roll pitch yaw x y z], altitudes[N], ground_ranges[M/2], anno)
 * raw may stay resident in HBM (device pointer): nothing is copied back to the host.
 * LIFETIME OF raw (the reference's Frame keeps a ref-counted cv::Mat, frame.cpp:23-28; a C pointer has no such thing):
 *   - device pointer: borrowed, must stay valid and unchanged until the frame is set again or the context is destroyed;
 *   - pageable host pointer: copied before the call returns, free to reuse afterwards;
 *   - PAGE-LOCKED host pointer (hipHostMalloc / hipHostRegister / torch pin_memory): NOT copied by this call -- the upload is
 *     deferred to dsss_extract / dsss_extract_many, which stream the images in on a copy stream under the kernels of the
 *     frames before them.  The buffer must stay valid and unchanged until the dsss_extract* call that covers this frame has
 *     RETURNED (with or without an error: error exits drain the copy stream first).  Setting the frame again, or
 *     destroying the context, before extracting it drops the pending upload without reading the buffer.
 * pose6 / alt / grange are copied before the call returns.                                                 */
int dsss_frame_set(dsss_ctx*, int id, const double* raw, int N, int M,
                   const double* pose6, const double* alt, const double* grange);
/* the same for n frames in one call (arrays of per-frame arguments; raw[i] may be NULL; ids distinct).  The
 * geometry of the whole call is staged once and uploaded with ONE copy; N and M must be below 65536.
 * As the reference's constructor normalises, masks and runs DetectFeature itself (frame.cpp:45-52), this call STARTS the extraction
 * of the frames whose raw image is a device pointer (asynchronously, with the parameters set at that moment) while it packs the
 * geometry; dsss_extract_many over the same frames then only finishes it.  Nothing observable changes: parameters set in between, another
 * list of frames or dsss_extract make the extraction run again from the start.                                                          */
int dsss_frames_set(dsss_ctx*, int n, const int* ids, const double* const* raw, const int* N, const int* M,
                    const double* const* pose6, const double* const* alt, const double* const* grange);
/* GetNormalizeSSS + GetFilteredMask + DetectFeature (frame.cpp:57-124,167-203) with the ORB descriptor
 * configuration of thirdparty/ORBextractor.cpp (operator() :1049-1113).  Features stay on the device.       */
int dsss_extract(dsss_ctx*, int id, int* n_kp_host);
int dsss_extract_many(dsss_ctx*, const int* ids, int n);      /* same, pipelined over frames */
/* stage taps for parity tests (host outputs; any may be NULL) */
int dsss_frame_get_norm(dsss_ctx*, int id, uint8_t* norm_host, uint8_t* mask_host);            /* Frame::norm_img, flt_mask */
int dsss_frame_get_level(dsss_ctx*, int id, int level, uint8_t* img_host, int* rows, int* cols);/* mvImagePyramid[level] */
int dsss_frame_get_candidates(dsss_ctx*, int id, int level, float* x_host, float* y_host, float* resp_host,
                              int cap, int* n);                                                /* vToDistributeKeys */
/* ORBextractor::DistributeOctTree (ORBextractor.cpp:539-763) as a HOST routine: the hot path runs the device version
 * (dsss_quadtree.hip); this one exists so the CPU test-suite can pin the list semantics against the oracle without a
 * GPU.  keep_idx_host needs n entries.                                                                          */
int dsss_host_quadtree(const float* x_host, const float* y_host, const float* resp_host, int n,
                       int minX, int maxX, int minY, int maxY, int quota, int32_t* keep_idx_host, int* n_keep);
/* Frame::kps / Frame::dst (+ the geo_img samples FEAmatcher.cpp:81-82 reads) */
int dsss_features_get(dsss_ctx*, int id, dsss_kp* kps_host, uint8_t* desc_host, double* geo_host, int cap, int* n);
/* Frame::dst as the SIFT call site would fill it (N x 128 CV_32F, integer-valued 0..255): the rows of DSSS_DESC_SIFT128, in the order of
 * dsss_features_get.  DSSS_E_STATE when the frame was extracted without them.  dsss_features_set_sift imports such rows for a frame whose
 * other features were given with dsss_features_set (values are rounded and clamped to 0..255: the device keeps them as bytes).          */
int dsss_features_get_sift(dsss_ctx*, int id, float* desc128_host, int cap, int* n);
int dsss_features_set_sift(dsss_ctx*, int id, const float* desc128_host, int n);
/* import features computed elsewhere (another rank's all-gather, or a test); geo/bbox may be NULL when the
 * frame geometry was given with dsss_frame_set (they are then recomputed on the device)                     */
int dsss_features_set(dsss_ctx*, int id, int N, int M, const dsss_kp* kps, const uint8_t* desc,
                      const double* geo, const double* bbox, int n);
/* geo bounding box = the minMaxLoc pairs of FEAmatcher.cpp:71-72 / util.cpp:21-26: xmin,xmax,ymin,ymax */
int dsss_frame_bbox(dsss_ctx*, int id, double* bbox_host);
/* Frame::geo_img in full -- frame.cpp:126-165 GetGeoImg, the two N x M CV_64F matrices (x, then y; row-major) of frame.h:40.  The hot
 * path never builds them (it uses dsss_frame_bbox and the geo samples of dsss_features_get); this is for callers that read the field. */
int dsss_frame_get_geo(dsss_ctx*, int id, double* x_host, double* y_host);
/* Util::ComputeIntersection (util.h:25, util.cpp:13-43) */
int dsss_overlap(dsss_ctx*, int id_s, int id_t, float* iou_host);
/* packed per-frame feature record for collectives (all-gather over RCCL): size and (de)serialisation */
size_t dsss_features_pack_bytes(const dsss_ctx*);
int dsss_features_pack(dsss_ctx*, int id, void* dev_or_host_buf);
int dsss_features_unpack(dsss_ctx*, int id, const void* dev_or_host_buf);

/* ------------------------------------------------------------------ FEAmatcher (FEAmatcher.h:20-33)
 * One call = FEAmatcher::RobustMatching for every listed (source,target) pair, batched on the device:
 * GeoNearNeighSearch both directions (:52-321) + ConsistentCheck (:323-405) + the rows RobustMatching appends
 * to Frame::corres_kps (:35-45) + Optimizer::GetKpsPairs (optimizer.cpp:575-639) on those rows.
 * Results stay on the device; the getters copy one pair out.                                               */
int dsss_match_pairs(dsss_ctx*, const int* src_ids, const int* tgt_ids, int npairs);
int dsss_match_get_dir(dsss_ctx*, int pair, int dir /*0: s->t, 1: t->s*/, int32_t* corres_nn_host,
                       int32_t* corres_host, int cap, int* scc_hist, int* scc_count, double* scc_model);
int dsss_match_get_rows(dsss_ctx*, int pair, double* rows6_host, int cap, int* nrows);   /* [id_s,id_t,y_s,x_s,y_t,x_t] */
int dsss_match_get_kp7(dsss_ctx*, int pair, double* kp7_host, int cap, int* n);          /* Vector7 of optimizer.cpp:625 */
int dsss_match_total(dsss_ctx*, int* total_rows, int* total_kp7);
/* 0 when the geo bounding boxes of the pair are disjoint: every keypoint is then skipped by FEAmatcher.cpp:84, no kernel runs */
int dsss_match_pair_active(dsss_ctx*, int pair, int* active_host);
/* FEAmatcher::DescriptorDistance (FEAmatcher.h:33) on device-resident descriptors of two frames */
int dsss_descriptor_distance(dsss_ctx*, int id_a, int ia, int id_b, int ib, int* dist_host);

/* ------------------------------------------------------------------ Optimizer (optimizer.h:43-67)
 * LoopClosingTFs (optimizer.cpp:641-982) for every kp pair produced by dsss_match_pairs.                   */
int dsss_lc_solve_all(dsss_ctx*);
int dsss_lc_get(dsss_ctx*, int pair, dsss_lc* out_host, int cap, int* n);
/* stand-alone form: kp7 given by the caller (Optimizer::LoopClosingTFs of ONE pair, optimizer.h:61-67)     */
int dsss_lc_solve(dsss_ctx*, int id_s, int id_t, const double* kp7, int n, dsss_lc* out_host);
/* the same for the kp7 lists of MANY pairs in one launch (the pair loop of TrajOptimizationAll, optimizer.cpp:35-97,
 * with kp7 built by the caller's GetKpsPairs from corres_kps or -- USE_ANNO = 1, optimizer.cpp:26,42-53 -- anno_kps).
 * kp7: pair_off[npairs] x 7 (host or device); pair_off: npairs + 1 ascending offsets (host).  Results stay on the device
 * as after dsss_match_pairs + dsss_lc_solve_all: dsss_lc_get / dsss_posegraph_select / dsss_posegraph_solve follow.     */
int dsss_lc_solve_pairs(dsss_ctx*, const int* src_ids, const int* tgt_ids, int npairs, const double* kp7, const int* pair_off_host);
/* LMTriaFactor + Optimizer::TriangulateOneLandmark (LMtriangulatefactor.cpp:10-27; optimizer.h:56-59, optimizer.cpp:984-1021)
 * for every kp7 row of one pair, as LoopClosingTFs calls it (optimizer.cpp:907-921: yaw-compensated DR poses, landmark
 * initialised as in :789-795).  out7_host: n x 7 = [x y z | |range_s err| |plane_s| |range_t err| |plane_t|]          */
int dsss_triangulate(dsss_ctx*, int id_s, int id_t, const double* kp7, int n, double* out7_host);
/* the same with the caller's poses and start point (the literal signature of optimizer.h:56-59, Ts = identity):
 * in27: n x [Tp_s R(9) t(3) | Tp_t R(9) t(3) | lm_ini(3)]; of kp7 only the slant ranges [2], [5] are read          */
int dsss_triangulate_poses(dsss_ctx*, const double* kp7, const double* in27, int n, double* out7_host);
/* TrajOptimizationAll (optimizer.h:43; optimizer.cpp:101-279): LC selection + batch LM over every ping of
 * every frame 0..nframes-1 (frames must have been given with dsss_frame_set). poses12_host: total x 12
 * (R row-major, t); rpy6_host: total x 6 "r p y x y z" as SaveTrajactoryAll writes (:1164-1214), may be NULL.
 * Page-locked output buffers make the download run at PCIe speed.                                             */
int dsss_posegraph_select(dsss_ctx*, int nframes, dsss_lc_edge* edges_host, int cap, int* n_edges);
int dsss_posegraph_solve(dsss_ctx*, int nframes, double* poses12_host, double* rpy6_host, double* stats4_host);
/* N3, the online use (optimizer.cpp:134-139, 262-272: one ISAM2 object, isam.update as pings arrive, calculateEstimate):
 * dsss_posegraph_update solves frames 0..nframes-1 like dsss_posegraph_solve, but (a) starts from the estimate the previous
 * update of this context left for the pings it covered (new pings start at DR o noise as in the reference), and (b) works on
 * the ACCUMULATED loop closures: every LC result set (dsss_lc_solve_all / dsss_lc_solve_pairs) is consumed once, by the next
 * update; its pairs must lie within the nframes frames.  dsss_posegraph_reset forgets estimate and edges.  Single rank.     */
int dsss_posegraph_update(dsss_ctx*, int nframes, double* poses12_host, double* rpy6_host, double* stats4_host);
/* The INCREMENTAL form (round 6): the same bookkeeping, but only the last `window_frames` frames are solved -- their pings and the loop
 * closures that end in them -- CONDITIONED on the estimate the earlier updates left for everything before: the window's first ping is pinned
 * there, a loop closure from a frozen ping into the window keeps its exact residual with the frozen end folded into its measurement, closures
 * between frozen pings drop out.  Cost of an update: the window, whatever the length of the survey (the global form re-analyses and
 * re-factorises the whole graph, O(F) per update, O(F^2) over a survey).  The frozen part is NOT revisited: what iSAM2 does for variables
 * below its relinearisation threshold (optimizer.cpp:134-137) done by age -- a later dsss_posegraph_update (global, warm-started from these
 * estimates: one or two LM trials) gives the batch optimum, which is all the reference ever reads (calculateEstimate after the loop, :279).
* poses12_host / rpy6_host (either may be NULL: nothing is downloaded) receive ALL pings of frames 0 .. nframes-1.  Falls back to the global
 * form while there is no frozen part (nframes <= window_frames); a window that would start in frames no update has covered (window_frames = 1:
 * the new frame alone) is extended backwards to the last frame that has an estimate, which anchors it.  Loop closures must end in the later ping (the pipeline's do).          */
int dsss_posegraph_update_window(dsss_ctx*, int nframes, int window_frames, double* poses12_host, double* rpy6_host, double* stats4_host);
int dsss_posegraph_reset(dsss_ctx*);
int dsss_posegraph_online_edges(dsss_ctx*);      /* loop closures accumulated so far (>= 0) */
/* Instrumentation (no reference counterpart: GTSAM keeps its elimination tree to itself): the panel levels of the last solve of this
 * context, four ints per level in launch order -- panel steps on the level, scalar columns of its widest panel, scalar rows below its
 * tallest one, 1 for a level of the replicated interface tree -- and the number of factorisations (LM trials) that solve ran.        */
int dsss_posegraph_schedule_get(dsss_ctx*, int* levels4_host, int cap_levels, int* n_levels, int* n_trials);
/* stand-alone form: explicit DR chain + edges                                                               */
int dsss_posegraph_solve_edges(dsss_ctx*, const double* dr6, int total, const dsss_lc_edge* edges, int ne,
                               double* poses12_host, double* stats4_host);

/* Host twin of the pose-graph linear solve (ordering + symbolic analysis + multifrontal factorisation of the reduced
 * system, diasss_amd/csrc/dsss_pg_sym.cpp) so that the CPU test-suite can pin the analysis the device kernels run on without
 * a GPU; the hot path runs the numeric phase in dsss_pg.hip.  Matrix: ns block variables (6x6 blocks); value index v < ns is
 * the diagonal block of variable v, ns + e the block H(edge_a[e], edge_b[e]); the first ns-1 edges must be the chain (e, e+1).
 * part (may be NULL): rank of every variable, non-decreasing.  x: ns x 6.  stats8: nnz(L) blocks, fronts, panels, levels,
 * front arena doubles, comm doubles, binned columns, largest front (block rows).                                          */
int dsss_host_pg_solve(int ns, const int32_t* edge_a, const int32_t* edge_b, int nedges, const double* cx, const double* cy,
                       const int32_t* part, int nparts, const double* aval36, const double* rhs6, double* x6_host, int64_t* stats8);
/* the same for the analysis ONE RANK of several runs (dsss_pg.hip, rank-local mode): its own variables plus the interface
 * variables iface_last (ascending), which are eliminated last as one dense front whatever their edges.  Any edge list (no
 * chain prefix).  stats8[5]: original values that land in the interface front.                                            */
int dsss_host_pg_solve_local(int ns, const int32_t* edge_a, const int32_t* edge_b, int nedges, const double* cx, const double* cy,
                             const int32_t* iface_last, int nlast, const double* aval36, const double* rhs6, double* x6_host, int64_t* stats8);
/* the same for ONE rank analysed BY PARTS (pg_symbolic_parts: K parts of equal size in the chain order, each ordered and analysed on
 * its own with the interface between the parts as the last, dense front, the results joined into one set of tables); the first
 * ns-1 edges must be the chain.                                                                                               */
int dsss_host_pg_solve_parts(int ns, const int32_t* edge_a, const int32_t* edge_b, int nedges, const double* cx, const double* cy,
                             int K, const double* aval36, const double* rhs6, double* x6_host, int64_t* stats8);

/* ------------------------------------------------------------------ instrumentation
 * accumulated GPU time (ms, HIP events on the context stream) and launch count per kernel family        */
#define DSSS_K_ROW_REDUCE   0   /* row_reduce_kernel: one f64 read of the waterfall */
#define DSSS_K_PRE_MISC     1   /* final_reduce + mask_init */
#define DSSS_K_NORMALIZE    2   /* normalize_kernel: f64 read, u8 write, hot-pixel scatter */
#define DSSS_K_PYRAMID      3   /* resize_kernel x (nlevels-1) */
#define DSSS_K_FAST         4   /* fast_cells_kernel */
#define DSSS_K_FAST_COMPACT 5   /* scan + gather of candidates */
#define DSSS_K_DESC         6   /* orient_desc_kernel */
#define DSSS_K_FILTER       7   /* mask_filter_kernel */
#define DSSS_K_MATCH        8   /* mt_grid_build_kernel + match_grid_kernel (or match_nn_kernel, all pairs) */
#define DSSS_K_SCC          9   /* scc_kernel */
#define DSSS_K_ROWS        10   /* pair_rows count/scan/write */
#define DSSS_K_LC          11   /* lc_kernel */
#define DSSS_K_PG          12   /* pose-graph LM loop (all its kernels) */
#define DSSS_K_QUADTREE    13   /* quadtree_kernel + collect (K4 on the device) */
#define DSSS_K_PG_ACC      14   /* pg_front_syrk_kernel (trailing update of the fronts: the bulk of the factorisation flops) */
#define DSSS_K_PG_DIAG     15   /* pg_front_diag_kernel */
#define DSSS_K_PG_TRSM     16   /* pg_front_trsm_kernel */
#define DSSS_K_PG_BWD      17   /* pg_front_bwd_kernel */
#define DSSS_K_PG_SUBTREE  18   /* pg_factor_subtree_kernel + pg_bwd_subtree_kernel */
#define DSSS_K_PG_ASM      19   /* pg_front_asm_kernel (extend-add) */
#define DSSS_K_PG_COMM     20   /* the reduced-Hessian all-reduce of a trial (work = bytes) */
#define DSSS_K_PG_RSU      21   /* pg_front_rsu_kernel: row solve + trailing update fused per tile (the levels with few tiles) */
#define DSSS_K_MATCH_DONE 22   /* no kernel of its own: work = the gate evaluations the matcher actually performed (its geo grid skips the cells out of reach; DSSS_K_MATCH's work is the reference's Na x Nb) */
#define DSSS_K_SIFT       23   /* sift_desc_kernel (DSSS_DESC_SIFT128): work = bytes of the 71 x 71 raw windows read + the 128-byte rows written */
#define DSSS_K_COUNT       24
int dsss_profile_enable(dsss_ctx*, int on);
int dsss_profile_get(dsss_ctx*, double* ms_host /*DSSS_K_COUNT*/, int64_t* launches_host /*DSSS_K_COUNT*/);
int dsss_profile_reset(dsss_ctx*);
/* algorithmic work accumulated next to the time of each slot while profiling is on: bytes for the streaming kernels
 * (slots 0..13), f64 flops for the pose-graph factorisation kernels (slots 14..18); definitions in DESIGN.md section 4 */
int dsss_profile_get_work(dsss_ctx*, double* work_host /*DSSS_K_COUNT*/);

#ifdef __cplusplus
}
#endif
#endif /* DSSS_H */
