// diasss_amd/csrc/dsss_internal.h -- shared state of libdsss.so (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include "../../include/dsss.h"

#define DSSS_MAX_LEVELS 8
#define DSSS_PI_REF 3.14159265359   // the reference's PI macro (frame.cpp:16, FEAmatcher.cpp:11, optimizer.cpp:19)

struct dsss_frame {
    int N = 0, M = 0;
    bool has_geom = false, has_raw = false, has_feat = false, has_norm = false;
    bool has_sift = false;            // the frame's rows of desc128 are valid (extracted with DSSS_DESC_SIFT128 or imported)
    const double* raw = nullptr;      // device; borrowed when the caller passed a device pointer
    double* raw_owned = nullptr;      // device; owned copy of a host image
    const double* raw_host = nullptr; // page-locked host image whose upload is still pending (dsss_extract_many streams it in under the kernels)
    bool raw_pending = false;
    double* pose6 = nullptr;          // device N x 6
    double* alt = nullptr;            // device N
    double* gr = nullptr;             // device M/2
    double* h_pack = nullptr;         // pinned host copy [pose6 N*6 | alt N | gr M/2] (also the pose-graph DR input)
    double* d_pack = nullptr;         // device copy, pose6 / alt / gr point into it
    size_t pack_cap = 0;
    hipEvent_t pack_ev = nullptr;     // recorded after the upload of h_pack: the staging area is reusable once it fired
    const double* h_geo = nullptr;    // host view of [pose6 | alt | gr]: h_pack, or a slice of a dsss_frames_set batch
    int gbatch = -1;                  // batch of dsss_frames_set the geometry lives in, -1: own d_pack / h_pack
    uint8_t* mask = nullptr;          // device N x M
    uint8_t* lvl[DSSS_MAX_LEVELS] = {nullptr};  // image pyramid, lvl[0] = normalised image
    int lrows[DSSS_MAX_LEVELS] = {0}, lcols[DSSS_MAX_LEVELS] = {0};
    size_t img_cap = 0;               // bytes allocated for mask / lvl[0]
    int nkp = 0;
    double bbox[4] = {0, 0, 0, 0};
    bool has_bbox = false, bbox_async = false;
    // FAST candidates of the last extraction, per level (host, for the stage tap)
    std::vector<float> cand_x[DSSS_MAX_LEVELS], cand_y[DSSS_MAX_LEVELS], cand_r[DSSS_MAX_LEVELS];
};

// geometry of a whole dsss_frames_set call: one pinned staging area, one device buffer, ONE upload
struct dsss_geo_batch { double* d = nullptr; double* h = nullptr; size_t cap = 0; int refs = 0; hipEvent_t ev = nullptr; };

struct dsss_comm;                        // dsss_comm.hip

struct dsss_prof {
    bool on = false;
    double ms[DSSS_K_COUNT] = {0};
    int64_t launches[DSSS_K_COUNT] = {0};
    double work[DSSS_K_COUNT] = {0};      // algorithmic bytes / flops (include/dsss.h)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // event pairs of the scopes that have been recorded but not read yet: reading them without a host synchronisation per
    // scope keeps the kernels back to back on the stream, so a short launch is timed as it runs in production (a synchronised
    // scope adds the idle-queue dispatch latency, 5-10 us, to every launch)
    struct rec { int k, nl; hipEvent_t e0, e1; };
    std::vector<rec> pending;
    std::vector<hipEvent_t> pool;
};
void dsss_prof_flush(dsss_ctx* c);       // synchronises the stream and folds the pending event pairs into ms[] / launches[]

struct dsss_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t xs[4] = {nullptr, nullptr, nullptr, nullptr};   // extra streams: frames of one extraction batch overlap on them
    hipEvent_t xev[4] = {nullptr, nullptr, nullptr, nullptr}; hipEvent_t xev_main = nullptr;
    hipEvent_t ex_lev_ev[DSSS_MAX_LEVELS] = {}, ex_cmp_ev[DSSS_MAX_LEVELS] = {}; hipEvent_t ex_side_ev[3] = {};    // extraction: FAST of level group g done (main stream), its candidate offsets known (side stream); quadtrees of a side stream done
    std::string err;
    dsss_mask_params mp;
    dsss_orb_params op;
    dsss_match_params mt;
    dsss_pg_params pg;
    int max_frames = 0;
    int kcap = 0;                       // per-frame feature capacity (multiple of 64)
    std::vector<dsss_frame> frames;
    // feature store, device, frame-major with stride kcap
    dsss_kp* kps = nullptr;             // [F][kcap]
    uint8_t* desc = nullptr;            // [F][kcap][32]
    uint8_t* desc128 = nullptr;         // [F][kcap][128]: the integer-valued floats of DSSS_DESC_SIFT128 kept as bytes; allocated on first use
    float* sift_w = nullptr;            // Gaussian window table of the SIFT descriptor, exp(-k / 512) (dsss_sift.hip)
    double* geo = nullptr;              // [F][kcap][2]
    int* nkp_dev = nullptr;             // [F]
    int* rows_dev = nullptr;            // [F] pings per frame
    int* cols_dev = nullptr;            // [F]
    double* bbox_dev = nullptr;         // [F][4]
    double* bbox_pinned = nullptr;      // [F][4] pinned host mirror, filled asynchronously
    bool bbox_pending = false;          // frames whose box has not been launched yet
    bool bbox_inflight = false; std::vector<int> bbox_inflight_ids; void* bbox_jobs_pinned = nullptr;      // boxes queued on the stream (dsss_bboxes_enqueue), not yet copied into dsss_frame::bbox
    void* bbox_jobs_dev = nullptr;      // [max_frames] job records of dsss_sync_bboxes (a hipMalloc / hipFree pair per call cost 0.15 ms)
    // extraction scratch (grown on demand)
    void* ex_scratch = nullptr; size_t ex_scratch_bytes = 0;
    void* ex_pinned = nullptr; size_t ex_pinned_bytes = 0;
    // extraction started by dsss_frames_set (everything but the kernel that needs the geometry): for which frames, under which parameters
    bool ex_eager_valid = false; std::vector<int> ex_eager_ids; dsss_orb_params ex_eager_op; dsss_mask_params ex_eager_mp;
    std::vector<dsss_geo_batch> gbatches;
    // pose-graph solver arena: device chunks kept between solves (dsss_pg.hip), bump-allocated, reset per solve
    std::vector<std::pair<void*, size_t>> pg_chunks; size_t pg_chunk_cur = 0, pg_chunk_off = 0;
    // matcher state
    void* mt_aux = nullptr; size_t mt_aux_bytes = 0;   // per-frame pointer tables + cv::RNG stream
    const double** d_ptrs = nullptr;                   // [3][max_frames]: alt, gr, pose6 device pointers
    int npairs = 0, nactive = 0;
    std::vector<int> pair_s, pair_t, pair_active;   // pair_active[p] = active index or -1
    int* act_s = nullptr; int* act_t = nullptr;     // device [nactive]
    int32_t* corres_nn = nullptr;       // [2*nactive][kcap]
    int32_t* corres = nullptr;          // [2*nactive][kcap]
    int* scc_hist = nullptr; int* scc_count = nullptr; double* scc_model = nullptr; // [2*nactive]
    int* row_cnt = nullptr; int* kp7_cnt = nullptr; int* row_off = nullptr; int* kp7_off = nullptr; // [nactive(+1)]
    std::vector<int> h_row_off, h_kp7_off;
    double* rows6 = nullptr; double* kp7 = nullptr; int* kp7_pair = nullptr; uint8_t* kp7_flip = nullptr;
    int total_rows = 0, total_kp7 = 0;
    size_t match_cap_pairs = 0, rows_cap = 0;
    // geo grid of the matcher: keypoints of the active pairs' frames sorted by cell (geo, descriptor, original index), cell offsets + tables
    double* mt_gs_geo = nullptr; uint8_t* mt_gs_desc = nullptr; int* mt_gs_idx = nullptr; size_t mt_gs_cap = 0;
    void* mt_cells = nullptr; size_t mt_cells_bytes = 0; unsigned long long mt_evals_host = 0;
    // LC results
    dsss_lc* lcs = nullptr; size_t lcs_cap = 0; bool has_lc = false;
    // pose-graph scratch
    void* pg_state = nullptr;
    // per-geometry extraction tables (dsss_extract.hip owns the type; freed through geoms_free by dsss_destroy)
    void* geoms = nullptr; void (*geoms_free)(void*) = nullptr;
    dsss_comm* comm = nullptr;          // ranks of one job (dsss_comm_init): null = single process
    int pg_parts = 0;                   // pose-graph partitions (0: one per rank); > ranks only to exercise the interface logic on few GPUs
    int* tmp_dev = nullptr;             // 64 ints of device scratch for one-value results (dsss_descriptor_distance)
    void* ag_buf = nullptr; size_t ag_cap = 0;                  // staging of dsss_features_allgather: world x (frames per rank) packed records, kept between calls
    void* ag_host = nullptr; size_t ag_host_cap = 0;            // page-locked landing place of the gathered headers (keypoint counts, boxes, sizes)
    double* xch_dev = nullptr; size_t xch_cap = 0;              // device scratch of the loop-closure exchange between ranks (dsss_posegraph_solve)
    void* pg_edges_host = nullptr; size_t pg_edges_cap = 0;     // page-locked staging of the selected LC edges (dsss_posegraph_solve)
    int* pg_ab_host = nullptr; size_t pg_ab_cap = 0;      // their end points as packed (a, b) pairs
    void* xch_host = nullptr; size_t xch_host_cap = 0;          // page-locked landing place of the gathered edge records of all ranks (bytes)
    double* pg_scal_host = nullptr;                             // page-locked landing place of the LM trial's scalars (8 doubles)
    std::vector<int> pg_last_levels;                            // schedule of the last solve, four ints per panel level: items, widest panel (scalar columns), tallest rows below, this rank's or the interface's (dsss_posegraph_schedule_get)
    int pg_last_trials = 0;                                     // factorisations of the last solve
    void* pg_stage = nullptr; size_t pg_stage_cap = 0;          // page-locked staging of the analysis tables: one upload per solve (pg_dev::flush)
    // online use (dsss_posegraph_update): the estimate of the previous update stays on the device, the LC edges accumulate
    void* pg_warm = nullptr; size_t pg_warm_cap = 0; int pg_warm_n = 0;   // pose_t[pg_warm_n]
    bool pg_online = false;             // set by dsss_posegraph_update: pg_solve_impl starts from pg_warm and leaves its result there
    int pg_win_f0 = 0, pg_win_p0 = 0;   // dsss_posegraph_update_window: first frame / first global pose of the window being solved (0: the whole graph)
    std::vector<dsss_lc_edge> pg_inc_edges; unsigned long long lc_gen = 0, pg_inc_gen = 0;    // lc_gen counts LC result sets; the last one consumed
    dsss_prof prof;
};

#define DSSS_FAIL(ctx, code, ...) do { char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__); (ctx)->err = _b; return (code); } while (0)
#define HIPCHK(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) { \
    char _b[512]; snprintf(_b, sizeof _b, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
    (ctx)->err = _b; return DSSS_E_HIP; } } while (0)

// kernel-family timing with HIP events on the context stream (dsss_profile_*)
struct dsss_scope {
    dsss_ctx* c; int k; int nl; hipEvent_t e0 = nullptr, e1 = nullptr;     // own event pair: scopes may nest (pose-graph solve)
    static hipEvent_t take(dsss_ctx* c) {
        hipEvent_t e = nullptr;
        if (!c->prof.pool.empty()) { e = c->prof.pool.back(); c->prof.pool.pop_back(); } else hipEventCreate(&e);
        return e;
    }
    dsss_scope(dsss_ctx* c_, int k_, double work = 0, int launches = 1) : c(c_), k(k_), nl(launches) {
        if (c->prof.on) { c->prof.work[k] += work; e0 = take(c); e1 = take(c); hipEventRecord(e0, c->stream); }
    }
    ~dsss_scope() {
        if (e0) {
            hipEventRecord(e1, c->stream);
            c->prof.pending.push_back({ k, nl, e0, e1 });
            if (c->prof.pending.size() >= 16384) dsss_prof_flush(c);
        }
    }
};

void dsss_extract_eager(dsss_ctx* c, const int* ids, int n);      // dsss_frames_set: start extracting the frames whose images are in HBM
int dsss_ensure_store(dsss_ctx* c);                 // allocate the feature store for the current kcap
int dsss_ensure_sift_store(dsss_ctx* c);            // ... and the 128-byte rows + the window table (DSSS_DESC_SIFT128)
struct ex_frame;
void dsss_launch_sift_desc(dsss_ctx* c, hipStream_t st, const ex_frame* d_exf, int kcap, int nb);      // dsss_sift.hip
int dsss_frame_geo_bbox(dsss_ctx* c, int id);       // device computation of the geo bounding box (asynchronous)
int dsss_bboxes_enqueue(dsss_ctx* c);               // queue the pending boxes on the context's stream (no synchronisation)
int dsss_sync_bboxes(dsss_ctx* c);                  // make dsss_frame::bbox valid on the host
int dsss_frame_kp_geo(dsss_ctx* c, int id, int n);  // geo lookup of the stored keypoints (frame.cpp:126-165)
void dsss_pg_free(dsss_ctx* c);
void dsss_comm_free(dsss_ctx* c);
int dsss_comm_rank(const dsss_ctx* c);
int dsss_comm_world(const dsss_ctx* c);
int dsss_comm_allreduce(dsss_ctx* c, double* dev, size_t n, hipStream_t st);
int dsss_comm_allgather(dsss_ctx* c, void* recv_dev, size_t bytes_per_rank, hipStream_t st);    // own slice in place at rank * bytes   // in-place sum over the ranks, ordered on st

// deterministic sin/cos shared by every kernel that must agree with the oracle bit for bit:
// Cody-Waite reduction + fdlibm kernel polynomials, plain IEEE mul/add only (library built -ffp-contract=off)
__host__ __device__ inline void dsss_sincos(double x, double* s, double* c)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;
    const double pio2_1t = 6.07710050650619224932e-11;
    double k = rint(x * invpio2);
    double r = (x - k * pio2_1) - k * pio2_1t;
    double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = ps * z + -2.50507602534068634195e-08;
    ps = ps * z + 2.75573137070700676789e-06;
    ps = ps * z + -1.98412698298579493134e-04;
    ps = ps * z + 8.33333333332248946124e-03;
    ps = ps * z + -1.66666666666666324348e-01;
    double sr = r + (r * z) * ps;
    double pc = -1.13596475577881948265e-11;
    pc = pc * z + 2.08757232129817482790e-09;
    pc = pc * z + -2.75573143513906633035e-07;
    pc = pc * z + 2.48015872894767294178e-05;
    pc = pc * z + -1.38888888888741095749e-03;
    pc = pc * z + 4.16666666666666019037e-02;
    double cr = (1.0 - 0.5 * z) + (z * z) * pc;
    long long q = ((long long)k) & 3;
    if (q == 0) { *s = sr; *c = cr; }
    else if (q == 1) { *s = cr; *c = -sr; }
    else if (q == 2) { *s = -sr; *c = -cr; }
    else { *s = -cr; *c = sr; }
}

// Frame::GetGeoImg for one bin (frame.cpp:126-165); port column 0 clamps the one-past-the-end read
__host__ __device__ inline void dsss_geo_at(const double* pose6, const double* gr, int M, int row, int col,
                                            double* x, double* y)
{
    const double* P = pose6 + (size_t)row * 6;
    int half = M / 2, idx; double ang;
    if (col >= half) { idx = col - half; ang = P[2] + DSSS_PI_REF / 2; }
    else { idx = half - col; if (idx > half - 1) idx = half - 1; ang = P[2] - DSSS_PI_REF / 2; }
    double s, c;
    dsss_sincos(ang, &s, &c);
    *x = (P[3] - 0.0) + gr[idx] * c;
    *y = (P[4] - 0.0) + gr[idx] * s;
}
