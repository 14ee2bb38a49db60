// diasss_amd/csrc/dsss_ctx.hip -- context, frame registration, feature store, geo lookup, overlap.
// Mirrors the data side of Diasss::Frame (/root/reference/src/core/frame.h:19-46, frame.cpp:18-55,126-165)
// and Util::ComputeIntersection (/root/reference/src/util/util.cpp:13-43).
#include "dsss_internal.h"
#include "dsss_pg_sym.h"
#include <algorithm>
#include <chrono>
#include <thread>
#include <atomic>
#include <cstdlib>

void dsss_prof_flush(dsss_ctx* c)
{
    if (!c || c->prof.pending.empty()) return;
    hipStreamSynchronize(c->stream);
    for (const dsss_prof::rec& r : c->prof.pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { c->prof.ms[r.k] += ms; c->prof.launches[r.k] += r.nl; }
        c->prof.pool.push_back(r.e0); c->prof.pool.push_back(r.e1);
    }
    c->prof.pending.clear();
}

extern "C" {

void dsss_mask_params_default(dsss_mask_params* p) { p->factor = 2.5; p->width = 10; p->r = 6; p->side = 150; }
void dsss_orb_params_default(dsss_orb_params* p) { p->nfeatures = 2000; p->scale = 1.2f; p->nlevels = 6; p->ini_th = 12; p->min_th = 7; p->descriptor = DSSS_DESC_ORB; }
void dsss_match_params_default(dsss_match_params* p)
{
    p->use_l2 = 0; p->radius = 8; p->bound_same = 88; p->bound_diff = 80; p->l2_bound = 350; p->ratio = 0.35;
    p->scc_iters = 1000; p->pix_err = 2.5; p->merge_thr = 2.5;
}
void dsss_pg_params_default(dsss_pg_params* p)
{
    p->max_iters = 100; p->rel_tol = 1e-5; p->abs_tol = 1e-5; p->lambda0 = 1e-5; p->lambda_factor = 10;
    p->lambda_max = 1e5; p->min_fidelity = 1e-3; p->add_noise = 1;
}

const char* dsss_strerror(int code)
{
    switch (code) {
    case DSSS_OK: return "ok";
    case DSSS_E_NODEVICE: return "no HIP device (libdsss has no CPU fallback)";
    case DSSS_E_ARG: return "bad argument";
    case DSSS_E_HIP: return "HIP runtime error";
    case DSSS_E_STATE: return "call order violated";
    case DSSS_E_CAPACITY: return "buffer too small";
    case DSSS_E_NUMERIC: return "numerical failure";
    case DSSS_E_COMM: return "communicator / RCCL failure";
    default: return "unknown error";
    }
}
const char* dsss_last_error(const dsss_ctx* c) { return c ? c->err.c_str() : "null context"; }

static int kcap_for(const dsss_orb_params& op) { return ((op.nfeatures + 3 * op.nlevels + 63) / 64 + 1) * 64; }

int dsss_create(int device, int max_frames, dsss_ctx** out)
{
    if (!out || max_frames <= 0) return DSSS_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return DSSS_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return DSSS_E_NODEVICE;
    dsss_ctx* c = new dsss_ctx();
    c->device = device;
    dsss_mask_params_default(&c->mp); dsss_orb_params_default(&c->op);
    dsss_match_params_default(&c->mt); dsss_pg_params_default(&c->pg);
    c->max_frames = max_frames;
    c->frames.resize(max_frames);
    c->kcap = kcap_for(c->op);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return DSSS_E_HIP; }
    hipEventCreate(&c->prof.e0); hipEventCreate(&c->prof.e1);
    // xs[1] carries the uploads (geometry, host-resident frames): nothing but copies.  It gets a priority of its own so that the runtime gives it a
    // hardware queue of its own -- streams of one priority are folded onto four hardware queues in creation order, and when the copy stream lands
    // on the queue of the compute stream (which it did or did not depending on how many streams the process had created before this context:
    // bench.py yes, a script that made one torch stream first no) every batch's kernels queue behind the NEXT batch's 2.4 ms of copies: the
    // PCIe-inclusive step 95 ms instead of 77 (round 6, found with a per-batch timer: kernels behind the upload 2.7 ms against 1.2)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    for (int i = 0; i < 4; ++i) {
        if (i == 1 && prio_hi != prio_lo) { if (hipStreamCreateWithPriority(&c->xs[i], hipStreamNonBlocking, prio_hi) != hipSuccess) { (void)hipGetLastError(); c->xs[i] = nullptr; } }
        if (!c->xs[i]) hipStreamCreateWithFlags(&c->xs[i], hipStreamNonBlocking);
        hipEventCreateWithFlags(&c->xev[i], hipEventDisableTiming);
    }
    for (int i = 0; i < DSSS_MAX_LEVELS; ++i) { hipEventCreateWithFlags(&c->ex_lev_ev[i], hipEventDisableTiming); hipEventCreateWithFlags(&c->ex_cmp_ev[i], hipEventDisableTiming); }
    for (int i = 0; i < 3; ++i) hipEventCreateWithFlags(&c->ex_side_ev[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&c->xev_main, hipEventDisableTiming);
    *out = c;
    return DSSS_OK;
}

static void free_frame(dsss_frame& f)
{
    hipFree(f.raw_owned); hipFree(f.d_pack); hipFree(f.mask);
    if (f.h_pack) hipHostFree(f.h_pack);
    if (f.pack_ev) hipEventDestroy(f.pack_ev);
    for (int l = 0; l < DSSS_MAX_LEVELS; ++l) hipFree(f.lvl[l]);
    f = dsss_frame();
}

static void free_match(dsss_ctx* c)
{
    hipFree(c->act_s); hipFree(c->act_t); hipFree(c->corres_nn); hipFree(c->corres);
    hipFree(c->scc_hist); hipFree(c->scc_count); hipFree(c->scc_model);
    hipFree(c->row_cnt); hipFree(c->kp7_cnt); hipFree(c->row_off); hipFree(c->kp7_off);
    hipFree(c->rows6); hipFree(c->kp7); hipFree(c->kp7_pair); hipFree(c->kp7_flip);
    c->act_s = c->act_t = nullptr; c->corres_nn = c->corres = nullptr;
    c->scc_hist = c->scc_count = nullptr; c->scc_model = nullptr;
    c->row_cnt = c->kp7_cnt = c->row_off = c->kp7_off = nullptr;
    c->rows6 = c->kp7 = nullptr; c->kp7_pair = nullptr; c->kp7_flip = nullptr;
    c->match_cap_pairs = 0; c->rows_cap = 0;
    hipFree(c->mt_gs_geo); hipFree(c->mt_gs_desc); hipFree(c->mt_gs_idx); hipFree(c->mt_cells);
    c->mt_gs_geo = nullptr; c->mt_gs_desc = nullptr; c->mt_gs_idx = nullptr; c->mt_cells = nullptr; c->mt_gs_cap = 0; c->mt_cells_bytes = 0;
}

static void free_store(dsss_ctx* c)
{
    hipFree(c->desc128); c->desc128 = nullptr;
    hipFree(c->kps); hipFree(c->desc); hipFree(c->geo); hipFree(c->nkp_dev); hipFree(c->rows_dev);
    hipFree(c->cols_dev); hipFree(c->bbox_dev);
    c->kps = nullptr; c->desc = nullptr; c->geo = nullptr; c->nkp_dev = nullptr; c->rows_dev = nullptr;
    c->cols_dev = nullptr; c->bbox_dev = nullptr;
}

void dsss_destroy(dsss_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto& f : c->frames) free_frame(f);
    for (auto& G : c->gbatches) { hipFree(G.d); if (G.h) hipHostFree(G.h); if (G.ev) hipEventDestroy(G.ev); }
    c->gbatches.clear();
    free_match(c); free_store(c);
    hipFree(c->lcs); hipFree(c->ex_scratch); hipFree(c->mt_aux); hipFree(c->tmp_dev); hipFree(c->sift_w);
    hipFree(c->ag_buf); if (c->ag_host) hipHostFree(c->ag_host); hipFree(c->xch_dev);
    if (c->pg_edges_host) hipHostFree(c->pg_edges_host);
    if (c->pg_ab_host) hipHostFree(c->pg_ab_host);
    if (c->xch_host) hipHostFree(c->xch_host);
    if (c->pg_stage) hipHostFree(c->pg_stage);
    if (c->pg_scal_host) hipHostFree(c->pg_scal_host);
    if (c->pg_warm) hipFree(c->pg_warm);
    if (c->geoms && c->geoms_free) c->geoms_free(c->geoms);
    if (c->ex_pinned) hipHostFree(c->ex_pinned);
    if (c->bbox_pinned) hipHostFree(c->bbox_pinned);
    hipFree(c->bbox_jobs_dev); if (c->bbox_jobs_pinned) hipHostFree(c->bbox_jobs_pinned);
    dsss_pg_free(c); dsss_comm_free(c);
    hipEventDestroy(c->prof.e0); hipEventDestroy(c->prof.e1);
    dsss_prof_flush(c);
    for (hipEvent_t e : c->prof.pool) hipEventDestroy(e);
    c->prof.pool.clear();
    for (int i = 0; i < 4; ++i) { if (c->xs[i]) hipStreamDestroy(c->xs[i]); if (c->xev[i]) hipEventDestroy(c->xev[i]); }
    for (int i = 0; i < DSSS_MAX_LEVELS; ++i) { if (c->ex_lev_ev[i]) hipEventDestroy(c->ex_lev_ev[i]); if (c->ex_cmp_ev[i]) hipEventDestroy(c->ex_cmp_ev[i]); }
    for (int i = 0; i < 3; ++i) if (c->ex_side_ev[i]) hipEventDestroy(c->ex_side_ev[i]);
    if (c->xev_main) hipEventDestroy(c->xev_main);
    hipStreamDestroy(c->stream);
    delete c;
}

int dsss_set_pg_partitions(dsss_ctx* c, int nparts) { if (!c || nparts < 0 || nparts > 4096) return DSSS_E_ARG; c->pg_parts = nparts; return DSSS_OK; }
int dsss_sync(dsss_ctx* c) { if (!c) return DSSS_E_ARG; HIPCHK(c, hipStreamSynchronize(c->stream)); return DSSS_OK; }
void* dsss_stream(dsss_ctx* c) { return c ? (void*)c->stream : nullptr; }

int dsss_set_params(dsss_ctx* c, const dsss_mask_params* mp, const dsss_orb_params* op, const dsss_match_params* mt,
                    const dsss_pg_params* pg)
{
    if (!c) return DSSS_E_ARG;
    if (mp) c->mp = *mp;
    if (op) {
        if (op->nlevels < 1 || op->nlevels > DSSS_MAX_LEVELS || op->nfeatures < 1 || !(op->scale > 1.0f) ||
            (op->descriptor != DSSS_DESC_ORB && op->descriptor != DSSS_DESC_SIFT128))
            DSSS_FAIL(c, DSSS_E_ARG, "orb params out of range");
        int k = kcap_for(*op);
        if (k != c->kcap) {
            if (c->kps) { HIPCHK(c, hipStreamSynchronize(c->stream)); free_store(c); free_match(c); for (auto& f : c->frames) { f.has_feat = false; f.has_sift = false; } }
            c->kcap = k;
        }
        c->op = *op;
    }
    if (mt) {
        if (mt->use_l2 < 0 || mt->use_l2 > 2) DSSS_FAIL(c, DSSS_E_ARG, "match params: use_l2 must be 0, 1 or 2");
        c->mt = *mt;
    }
    if (pg) c->pg = *pg;
    return DSSS_OK;
}

} // extern "C"

int dsss_ensure_store(dsss_ctx* c)
{
    if (c->kps) return DSSS_OK;
    size_t F = (size_t)c->max_frames, K = (size_t)c->kcap;
    HIPCHK(c, hipMalloc(&c->kps, F * K * sizeof(dsss_kp)));
    HIPCHK(c, hipMalloc(&c->desc, F * K * 32));
    HIPCHK(c, hipMalloc(&c->geo, F * K * 2 * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->nkp_dev, F * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->rows_dev, F * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->cols_dev, F * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->bbox_dev, F * 4 * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(c->nkp_dev, 0, F * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->rows_dev, 0, F * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->cols_dev, 0, F * sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(c->bbox_dev, 0, F * 4 * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(c->desc, 0, F * K * 32, c->stream));
    return DSSS_OK;
}

int dsss_ensure_sift_store(dsss_ctx* c)
{
    int rc = dsss_ensure_store(c); if (rc) return rc;
    if (!c->desc128) {
        const size_t bytes = (size_t)c->max_frames * c->kcap * 128;
        HIPCHK(c, hipMalloc(&c->desc128, bytes));
        HIPCHK(c, hipMemsetAsync(c->desc128, 0, bytes, c->stream));
    }
    if (!c->sift_w) {
        // exp(-k / 512) by repeated multiplication in double: the loop of oracle/orc_sift.c:orc_sift_weights, same bits
        const int n = 2 * 28 * 28 + 1;
        std::vector<float> w(n);
        const double q = 0.9980487811074755;
        double v = 1.0;
        for (int k = 0; k < n; ++k) { w[k] = (float)v; v = v * q; }
        HIPCHK(c, hipMalloc(&c->sift_w, n * sizeof(float)));
        HIPCHK(c, hipMemcpy(c->sift_w, w.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    return DSSS_OK;
}

// ---- geo bounding box: the four cv::minMaxLoc scans of the N x M geo image (FEAmatcher.cpp:71-72,
// util.cpp:21-26) collapse to per-row extremes because x = px + g*c is monotone in g for fixed c:
// only the smallest and largest ground range of each side can be extreme.  One wave per 64 rows.
struct bbox_job { const double* pose6; const double* gr; int N, M, id, pad; };
__global__ void geo_bbox_kernel(const bbox_job* __restrict__ jobs, double* __restrict__ bbox_all, int* __restrict__ rows_all, int* __restrict__ cols_all)
{
    const bbox_job J = jobs[blockIdx.x];
    const double* pose6 = J.pose6; const double* gr = J.gr; const int N = J.N, M = J.M;
    double* bbox = bbox_all + (size_t)J.id * 4; int* rows_out = rows_all + J.id; int* cols_out = cols_all + J.id;
    __shared__ double sgmin[2], sgmax[2];
    __shared__ double red[4][256];
    int half = M / 2;
    if (threadIdx.x < 2) {
        // starboard uses gr[0..half-1]; port uses gr[min(half-col, half-1)] for col 0..half-1 = gr[1..half-1]
        int lo = threadIdx.x == 0 ? 0 : (half > 1 ? 1 : 0);
        double mn = gr[lo], mx = gr[lo];
        for (int k = lo + 1; k < half; ++k) { double g = gr[k]; mn = g < mn ? g : mn; mx = g > mx ? g : mx; }
        sgmin[threadIdx.x] = mn; sgmax[threadIdx.x] = mx;
    }
    __syncthreads();
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int row = threadIdx.x; row < N; row += blockDim.x) {
        const double* P = pose6 + (size_t)row * 6;
        for (int side = 0; side < 2; ++side) {
            double ang = side == 0 ? P[2] + DSSS_PI_REF / 2 : P[2] - DSSS_PI_REF / 2;
            double s, c; dsss_sincos(ang, &s, &c);
            for (int e = 0; e < 2; ++e) {
                double g = e ? sgmax[side] : sgmin[side];
                double x = (P[3] - 0.0) + g * c, y = (P[4] - 0.0) + g * s;
                xmin = x < xmin ? x : xmin; xmax = x > xmax ? x : xmax;
                ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax;
            }
        }
    }
    red[0][threadIdx.x] = xmin; red[1][threadIdx.x] = xmax; red[2][threadIdx.x] = ymin; red[3][threadIdx.x] = ymax;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] = fmin(red[0][threadIdx.x], red[0][threadIdx.x + s]);
            red[1][threadIdx.x] = fmax(red[1][threadIdx.x], red[1][threadIdx.x + s]);
            red[2][threadIdx.x] = fmin(red[2][threadIdx.x], red[2][threadIdx.x + s]);
            red[3][threadIdx.x] = fmax(red[3][threadIdx.x], red[3][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) bbox[threadIdx.x] = red[threadIdx.x][0];
    if (threadIdx.x == 0) { *rows_out = N; *cols_out = M; }          // per-frame dimensions used by the matcher kernels
}

int dsss_frame_geo_bbox(dsss_ctx* c, int id)
{
    dsss_frame& f = c->frames[id];
    if (!f.has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no geometry", id);
    int rc = dsss_ensure_store(c); if (rc) return rc;
    if (c->bbox_inflight) { rc = dsss_sync_bboxes(c); if (rc) return rc; }      // (a box on its way belongs to the geometry the frame had: take it in before the frame changes)
    f.has_bbox = true; f.bbox_async = true; c->bbox_pending = true;     // computed for all pending frames at once (dsss_bboxes_enqueue)
    return DSSS_OK;
}

// one launch for every frame whose geo box is pending (one workgroup per frame) and one copy back into the page-locked mirror, QUEUED on
// the context's stream: the extraction queues it behind its own kernels, so that the synchronisation that ends the extraction brings the
// boxes back as well (launched by the matcher, the boxes were a launch, a copy and a round trip of their own between the two stages)
int dsss_bboxes_enqueue(dsss_ctx* c)
{
    if (!c->bbox_pending || c->bbox_inflight) return DSSS_OK;
    if (!c->bbox_pinned) HIPCHK(c, hipHostMalloc((void**)&c->bbox_pinned, (size_t)c->max_frames * 4 * sizeof(double), hipHostMallocDefault));
    if (!c->bbox_jobs_pinned) HIPCHK(c, hipHostMalloc((void**)&c->bbox_jobs_pinned, (size_t)c->max_frames * sizeof(bbox_job), hipHostMallocDefault));
    bbox_job* jobs = static_cast<bbox_job*>(c->bbox_jobs_pinned);
    c->bbox_inflight_ids.clear();
    for (int f = 0; f < c->max_frames; ++f)
        if (c->frames[f].bbox_async) { const dsss_frame& fr = c->frames[f]; jobs[c->bbox_inflight_ids.size()] = bbox_job{ fr.pose6, fr.gr, fr.N, fr.M, f, 0 }; c->bbox_inflight_ids.push_back(f); }
    c->bbox_pending = false;
    const size_t nj = c->bbox_inflight_ids.size();
    if (nj == 0) return DSSS_OK;
    if (!c->bbox_jobs_dev) HIPCHK(c, hipMalloc(&c->bbox_jobs_dev, (size_t)c->max_frames * sizeof(bbox_job)));
    bbox_job* d_jobs = static_cast<bbox_job*>(c->bbox_jobs_dev);
    hipError_t e = hipMemcpyAsync(d_jobs, jobs, nj * sizeof(bbox_job), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) { hipLaunchKernelGGL(geo_bbox_kernel, dim3((unsigned)nj), dim3(256), 0, c->stream, d_jobs, c->bbox_dev, c->rows_dev, c->cols_dev); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(c->bbox_pinned, c->bbox_dev, (size_t)c->max_frames * 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) { c->bbox_pending = true; HIPCHK(c, e); }
    c->bbox_inflight = true;
    return DSSS_OK;
}

// make dsss_frame::bbox valid on the host
int dsss_sync_bboxes(dsss_ctx* c)
{
    for (int round = 0; round < 2 && (c->bbox_pending || c->bbox_inflight); ++round) {      // (boxes on their way first, then whatever was set since)
        int rc = dsss_bboxes_enqueue(c); if (rc) return rc;
        if (c->bbox_inflight) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (int id : c->bbox_inflight_ids)
                if (c->frames[id].bbox_async) { memcpy(c->frames[id].bbox, c->bbox_pinned + (size_t)id * 4, 4 * sizeof(double)); c->frames[id].bbox_async = false; }      // (not a box the caller has set itself since: dsss_features_set)
            c->bbox_inflight = false; c->bbox_inflight_ids.clear();
        }
    }
    return DSSS_OK;
}

// geo_img sample at (int(pt.y), int(pt.x)) of every stored keypoint (FEAmatcher.cpp:81-82,89-90) without
// materialising the 2 x N x M f64 geo image
__global__ void kp_geo_kernel(const dsss_kp* __restrict__ kps, int n, const double* __restrict__ pose6,
                              const double* __restrict__ gr, int M, double* __restrict__ geo)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int row = (int)kps[i].y, col = (int)kps[i].x;
    double x, y;
    dsss_geo_at(pose6, gr, M, row, col, &x, &y);
    geo[2 * i] = x; geo[2 * i + 1] = y;
}

int dsss_frame_kp_geo(dsss_ctx* c, int id, int n)
{
    dsss_frame& f = c->frames[id];
    if (n <= 0) return DSSS_OK;
    hipLaunchKernelGGL(kp_geo_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream,
                       c->kps + (size_t)id * c->kcap, n, f.pose6, f.gr, f.M, c->geo + (size_t)id * c->kcap * 2);
    HIPCHK(c, hipGetLastError());
    return DSSS_OK;
}

extern "C" {

// dsss_frame_set in three steps so that dsss_frames_set can run the middle one (pure host copies into the pinned
// staging areas, the bulk of the time) on several threads.
static void release_gbatch(dsss_ctx* c, dsss_frame& f)
{
    if (f.gbatch >= 0) { c->gbatches[f.gbatch].refs--; f.gbatch = -1; f.pose6 = nullptr; f.alt = nullptr; f.gr = nullptr; f.h_geo = nullptr; }
}
static int frame_prepare(dsss_ctx* c, int id, int N, int M, const double* pose6, const double* alt, const double* grange)
{
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range [0,%d)", id, c->max_frames);
    if (N <= 0 || M < 4 || (M & 1) || !pose6 || !alt || !grange) DSSS_FAIL(c, DSSS_E_ARG, "bad frame geometry N=%d M=%d", N, M);
    dsss_frame& f = c->frames[id];
    release_gbatch(c, f);
    if (f.N != N || f.M != M) { HIPCHK(c, hipStreamSynchronize(c->stream)); free_frame(f); }
    f.N = N; f.M = M;
    const size_t pack = (size_t)N * 6 + N + M / 2;
    if (f.pack_cap < pack) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(f.d_pack); if (f.h_pack) hipHostFree(f.h_pack);
        f.d_pack = nullptr; f.h_pack = nullptr;
        HIPCHK(c, hipMalloc(&f.d_pack, pack * sizeof(double)));
        HIPCHK(c, hipHostMalloc((void**)&f.h_pack, pack * sizeof(double), hipHostMallocDefault));
        f.pack_cap = pack;
    }
    f.pose6 = f.d_pack; f.alt = f.d_pack + (size_t)N * 6; f.gr = f.alt + N; f.h_geo = f.h_pack;
    if (f.pack_ev) HIPCHK(c, hipEventSynchronize(f.pack_ev));        // the pinned staging area may still feed the previous upload
    else HIPCHK(c, hipEventCreateWithFlags(&f.pack_ev, hipEventDisableTiming));
    return DSSS_OK;
}
static hipError_t frame_fill(dsss_ctx* c, int id, double* dst_pack, const double* pose6, const double* alt, const double* grange)
{
    dsss_frame& f = c->frames[id];
    auto to_host = [&](double* dst, const double* src, size_t n) -> hipError_t {
        hipPointerAttribute_t at;
        const bool dev = (hipPointerGetAttributes(&at, src) == hipSuccess) && at.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        if (dev) return hipMemcpy(dst, src, n * sizeof(double), hipMemcpyDeviceToHost);
        memcpy(dst, src, n * sizeof(double));               // plain host memory: no runtime call
        return hipSuccess;
    };
    hipError_t e = to_host(dst_pack, pose6, (size_t)f.N * 6);
    if (e == hipSuccess) e = to_host(dst_pack + (size_t)f.N * 6, alt, (size_t)f.N);
    if (e == hipSuccess) e = to_host(dst_pack + (size_t)f.N * 7, grange, (size_t)(f.M / 2));
    return e;
}
static int frame_submit(dsss_ctx* c, int id, const double* raw, bool own_upload)
{
    dsss_frame& f = c->frames[id];
    const int N = f.N, M = f.M;
    if (own_upload) {
        const size_t pack = (size_t)N * 6 + N + M / 2;
        HIPCHK(c, hipMemcpyAsync(f.d_pack, f.h_pack, pack * sizeof(double), hipMemcpyHostToDevice, c->stream));   // pinned: truly asynchronous
        HIPCHK(c, hipEventRecord(f.pack_ev, c->stream));
    }
    f.has_geom = true; f.has_feat = false; f.has_norm = false; f.nkp = 0;          // no synchronisation: the sources above live in the context
    if (raw) {
        hipPointerAttribute_t at;
        bool on_dev = (hipPointerGetAttributes(&at, raw) == hipSuccess) && at.type == hipMemoryTypeDevice;
        (void)hipGetLastError();
        f.raw_pending = false; f.raw_host = nullptr;
        if (on_dev) { f.raw = raw; }
        else {
            if (!f.raw_owned) HIPCHK(c, hipMalloc(&f.raw_owned, (size_t)N * M * sizeof(double)));
            const bool pinned = (hipPointerGetAttributes(&at, raw) == hipSuccess) && at.type == hipMemoryTypeHost;
            (void)hipGetLastError();
            // a page-locked image is not copied here: the extraction streams it in, sub-batch by sub-batch, on a copy stream while
            // the kernels of the previous sub-batch run (the caller keeps the buffer alive until dsss_extract* returns).  Pageable
            // memory cannot be copied asynchronously: it goes up now.
            if (pinned) { f.raw_host = raw; f.raw_pending = true; }
            else HIPCHK(c, hipMemcpy(f.raw_owned, raw, (size_t)N * M * sizeof(double), hipMemcpyHostToDevice));
            f.raw = f.raw_owned;
        }
        f.has_raw = true;
    } else f.has_raw = false;
    return dsss_frame_geo_bbox(c, id);
}

int dsss_frame_set(dsss_ctx* c, int id, const double* raw, int N, int M, const double* pose6, const double* alt,
                   const double* grange)
{
    if (!c) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = dsss_ensure_store(c); if (rc) return rc;
    rc = frame_prepare(c, id, N, M, pose6, alt, grange); if (rc) return rc;
    HIPCHK(c, frame_fill(c, id, c->frames[id].h_pack, pose6, alt, grange));
    return frame_submit(c, id, raw, true);
}

int dsss_frames_set(dsss_ctx* c, int n, const int* ids, const double* const* raw, const int* N, const int* M,
                    const double* const* pose6, const double* const* alt, const double* const* grange)
{
    if (!c || n < 0 || (n > 0 && (!ids || !N || !M || !pose6 || !alt || !grange))) return DSSS_E_ARG;
    if (n == 0) return DSSS_OK;
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(c, hipSetDevice(c->device));
    int rc = dsss_ensure_store(c); if (rc) return rc;
    // the geometry of the whole call goes through ONE pinned staging area and ONE upload (200 separate 116 KB copies
    // cost 3-4 ms of copy-engine latency at C3); the frames point into the batch's device buffer
    std::vector<size_t> off(n + 1, 0);
    for (int i = 0; i < n; ++i) {
        const int id = ids[i];
        if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range [0,%d)", id, c->max_frames);
        if (N[i] <= 0 || M[i] < 4 || (M[i] & 1) || !pose6[i] || !alt[i] || !grange[i]) DSSS_FAIL(c, DSSS_E_ARG, "bad frame geometry N=%d M=%d", N[i], M[i]);
        dsss_frame& f = c->frames[id];
        release_gbatch(c, f);
        if (f.N != N[i] || f.M != M[i]) { HIPCHK(c, hipStreamSynchronize(c->stream)); free_frame(f); }
        f.N = N[i]; f.M = M[i];
        off[i + 1] = off[i] + (((size_t)N[i] * 7 + M[i] / 2 + 63) & ~(size_t)63);
    }
    const size_t total = off[n];
    int b = -1;
    for (size_t k = 0; k < c->gbatches.size(); ++k)
        if (c->gbatches[k].refs == 0 && c->gbatches[k].cap >= total && (b < 0 || c->gbatches[k].cap < c->gbatches[b].cap)) b = (int)k;
    if (b < 0) {
        for (size_t k = 0; k < c->gbatches.size() && b < 0; ++k) if (c->gbatches[k].refs == 0) b = (int)k;     // grow an idle one
        if (b < 0) { c->gbatches.push_back(dsss_geo_batch()); b = (int)c->gbatches.size() - 1; }
        dsss_geo_batch& G = c->gbatches[b];
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(G.d); if (G.h) hipHostFree(G.h);
        G.d = nullptr; G.h = nullptr; G.cap = 0;
        HIPCHK(c, hipMalloc(&G.d, total * sizeof(double)));
        HIPCHK(c, hipHostMalloc((void**)&G.h, total * sizeof(double), hipHostMallocDefault));
        G.cap = total;
        if (!G.ev) HIPCHK(c, hipEventCreateWithFlags(&G.ev, hipEventDisableTiming));
    }
    dsss_geo_batch& G = c->gbatches[b];
    HIPCHK(c, hipEventSynchronize(G.ev));                   // the staging area may still feed its previous upload
    // The frames point into the batch's device buffer first; then the extraction of the frames whose images are already in HBM is
    // STARTED (everything but the last kernel, which samples the geometry) and runs while this thread packs the geometry into the
    // staging area; the upload follows on the copy stream and the context's stream waits for it behind those kernels.  The
    // reference's Frame constructor normalises, masks and runs DetectFeature itself (frame.cpp:45-52); dsss_extract_many then
    // only finishes.  (The 1.3 ms of host work of a 200-frame call were idle time of the GPU.)
    for (int i = 0; i < n; ++i) {
        dsss_frame& f = c->frames[ids[i]];
        f.gbatch = b; G.refs++;
        f.pose6 = G.d + off[i]; f.alt = f.pose6 + (size_t)f.N * 6; f.gr = f.alt + f.N; f.h_geo = G.h + off[i];
        rc = frame_submit(c, ids[i], raw ? raw[i] : nullptr, false); if (rc) return rc;
    }
    HIPCHK(c, hipEventRecord(c->xev[0], c->stream));        // whatever is queued so far may still read the buffer's previous contents
    const auto t1 = std::chrono::steady_clock::now();
    if (raw) dsss_extract_eager(c, ids, n);
    const auto t2 = std::chrono::steady_clock::now();
    const int T_env = getenv("DSSS_FS_THREADS") ? atoi(getenv("DSSS_FS_THREADS")) : 0;
    const int T = T_env > 0 ? T_env : (n >= 16 ? 4 : 1);       // (eight threads were slower than four: the pointer-attribute queries of frame_fill serialise)
    std::vector<hipError_t> errs(T, hipSuccess);
    auto work = [&](int t) { for (int i = t; i < n; i += T) { const hipError_t e = frame_fill(c, ids[i], G.h + off[i], pose6[i], alt[i], grange[i]); if (e != hipSuccess) errs[t] = e; } };
    dsss_pool_run(T, work);
    for (int t = 0; t < T; ++t) if (errs[t] != hipSuccess) { c->ex_eager_valid = false; HIPCHK(c, errs[t]); }
    const auto t3 = std::chrono::steady_clock::now();
    HIPCHK(c, hipStreamWaitEvent(c->xs[1], c->xev[0], 0));
    HIPCHK(c, hipMemcpyAsync(G.d, G.h, total * sizeof(double), hipMemcpyHostToDevice, c->xs[1]));
    HIPCHK(c, hipEventRecord(G.ev, c->xs[1]));
    HIPCHK(c, hipStreamWaitEvent(c->stream, G.ev, 0));      // every later consumer of the geometry is ordered behind the upload
    if (getenv("DSSS_EX_VERBOSE")) { auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[dsss frames_set] %d frames in %.1f us: bookkeeping %.1f, eager extraction queued %.1f, geometry packed (%d threads) %.1f, upload queued %.1f\n", n, us(t0, std::chrono::steady_clock::now()), us(t0, t1), us(t1, t2), T, us(t2, t3), us(t3, std::chrono::steady_clock::now())); }
    return DSSS_OK;
}

int dsss_features_set(dsss_ctx* c, int id, int N, int M, const dsss_kp* kps, const uint8_t* desc, const double* geo,
                      const double* bbox, int n)
{
    if (!c) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range", id);
    if (n < 0 || n > c->kcap) DSSS_FAIL(c, DSSS_E_CAPACITY, "%d features exceed the per-frame capacity %d", n, c->kcap);
    if (n > 0 && (!kps || !desc)) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = dsss_ensure_store(c); if (rc) return rc;
    dsss_frame& f = c->frames[id];
    if (!f.has_geom) { f.N = N; f.M = M; }
    else if (f.N != N || f.M != M) DSSS_FAIL(c, DSSS_E_ARG, "frame %d geometry mismatch", id);
    size_t K = c->kcap;
    if (n > 0) {
        HIPCHK(c, hipMemcpyAsync(c->kps + id * K, kps, (size_t)n * sizeof(dsss_kp), hipMemcpyDefault, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->desc + id * K * 32, desc, (size_t)n * 32, hipMemcpyDefault, c->stream));
    }
    if (geo) { if (n > 0) HIPCHK(c, hipMemcpyAsync(c->geo + id * K * 2, geo, (size_t)n * 2 * sizeof(double), hipMemcpyDefault, c->stream)); }
    else {
        if (!f.has_geom) DSSS_FAIL(c, DSSS_E_STATE, "geo == NULL needs dsss_frame_set first");
        rc = dsss_frame_kp_geo(c, id, n); if (rc) return rc;
    }
    if (bbox) {
        f.bbox_async = false;
        HIPCHK(c, hipMemcpy(f.bbox, bbox, 4 * sizeof(double), hipMemcpyDefault));
        HIPCHK(c, hipMemcpyAsync(c->bbox_dev + (size_t)id * 4, f.bbox, 4 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        f.has_bbox = true;
    } else if (!f.has_bbox) DSSS_FAIL(c, DSSS_E_STATE, "bbox == NULL needs dsss_frame_set first");
    HIPCHK(c, hipMemcpyAsync(c->nkp_dev + id, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->rows_dev + id, &f.N, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->cols_dev + id, &f.M, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    f.nkp = n; f.has_feat = true; f.has_sift = false;
    return DSSS_OK;
}

int dsss_features_set_sift(dsss_ctx* c, int id, const float* d128, int n)
{
    if (!c) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range", id);
    dsss_frame& f = c->frames[id];
    if (!f.has_feat || n != f.nkp) DSSS_FAIL(c, DSSS_E_STATE, "frame %d: dsss_features_set with the same n first (have %d, got %d)", id, f.has_feat ? f.nkp : -1, n);
    if (n > 0 && !d128) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = dsss_ensure_sift_store(c); if (rc) return rc;
    std::vector<uint8_t> b((size_t)n * 128);
    for (size_t i = 0; i < b.size(); ++i) { const float v = std::nearbyint(d128[i]); b[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
    if (n > 0) HIPCHK(c, hipMemcpy(c->desc128 + (size_t)id * c->kcap * 128, b.data(), b.size(), hipMemcpyHostToDevice));
    f.has_sift = true;
    return DSSS_OK;
}

int dsss_features_get_sift(dsss_ctx* c, int id, float* d128, int cap, int* n)
{
    if (!c) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range", id);
    dsss_frame& f = c->frames[id];
    if (!f.has_feat || !f.has_sift) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no SIFT rows (extract with dsss_orb_params.descriptor = DSSS_DESC_SIFT128)", id);
    if (n) *n = f.nkp;
    if (cap < f.nkp) DSSS_FAIL(c, DSSS_E_CAPACITY, "caller capacity %d < %d features", cap, f.nkp);
    if (f.nkp > 0) {
        if (!d128) return DSSS_E_ARG;
        std::vector<uint8_t> b((size_t)f.nkp * 128);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(b.data(), c->desc128 + (size_t)id * c->kcap * 128, b.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < b.size(); ++i) d128[i] = (float)b[i];
    }
    return DSSS_OK;
}

int dsss_features_get(dsss_ctx* c, int id, dsss_kp* kps, uint8_t* desc, double* geo, int cap, int* n)
{
    if (!c) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range", id);
    dsss_frame& f = c->frames[id];
    if (!f.has_feat) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no features", id);
    if (n) *n = f.nkp;
    if (cap < f.nkp) DSSS_FAIL(c, DSSS_E_CAPACITY, "caller capacity %d < %d features", cap, f.nkp);
    size_t K = c->kcap;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (f.nkp > 0) {
        if (kps) HIPCHK(c, hipMemcpy(kps, c->kps + id * K, (size_t)f.nkp * sizeof(dsss_kp), hipMemcpyDeviceToHost));
        if (desc) HIPCHK(c, hipMemcpy(desc, c->desc + id * K * 32, (size_t)f.nkp * 32, hipMemcpyDeviceToHost));
        if (geo) HIPCHK(c, hipMemcpy(geo, c->geo + id * K * 2, (size_t)f.nkp * 2 * sizeof(double), hipMemcpyDeviceToHost));
    }
    return DSSS_OK;
}

int dsss_frame_bbox(dsss_ctx* c, int id, double* bbox)
{
    if (!c || !bbox) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range", id);
    if (!c->frames[id].has_bbox) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no bounding box", id);
    { int rc = dsss_sync_bboxes(c); if (rc) return rc; }
    memcpy(bbox, c->frames[id].bbox, 4 * sizeof(double));
    return DSSS_OK;
}

// Frame::geo_img in full (frame.cpp:126-165): the product never materialises it -- every consumer on the hot path takes its extremes
// (dsss_frame_bbox) or samples it at keypoints (dsss_features_get) -- but the reference's field is the N x M pair of matrices, and a caller
// that wants exactly that gets it here: one kernel over the pixels (the arithmetic of dsss_geo_at, what geo_bbox_kernel and kp_geo_kernel
// evaluate), two N x M f64 downloads.  Off the hot path (32 MB per frame at 2000 x 1024).
__global__ __launch_bounds__(256) void geo_image_kernel(const double* __restrict__ pose6, const double* __restrict__ gr, int N, int M,
                                                        double* __restrict__ gx, double* __restrict__ gy)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)N * M) return;
    const int row = (int)(i / M), col = (int)(i - (size_t)row * M);
    double x, y;
    dsss_geo_at(pose6, gr, M, row, col, &x, &y);
    gx[i] = x; gy[i] = y;
}
int dsss_frame_get_geo(dsss_ctx* c, int id, double* x_host, double* y_host)
{
    if (!c || !x_host || !y_host) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range", id);
    dsss_frame& f = c->frames[id];
    if (!f.has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no geometry", id);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)f.N * f.M;
    double* d = nullptr;
    HIPCHK(c, hipMalloc(&d, 2 * n * sizeof(double)));
    hipLaunchKernelGGL(geo_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, f.pose6, f.gr, f.N, f.M, d, d + n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(x_host, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(y_host, d + n, n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d);
    HIPCHK(c, e);
    return DSSS_OK;
}

// Util::ComputeIntersection (util.cpp:13-43): float arithmetic on the double extrema; host code because it is
// 20 flops on values that already live on the host (the expensive part, the 4 minMaxLoc scans, is the bbox kernel)
int dsss_overlap(dsss_ctx* c, int id_s, int id_t, float* iou)
{
    if (!c || !iou) return DSSS_E_ARG;
    if (id_s < 0 || id_s >= c->max_frames || id_t < 0 || id_t >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id out of range");
    const dsss_frame &a = c->frames[id_s], &b = c->frames[id_t];
    if (!a.has_bbox || !b.has_bbox) DSSS_FAIL(c, DSSS_E_STATE, "frames need dsss_frame_set first");
    { int rc = dsss_sync_bboxes(c); if (rc) return rc; }
    float output = 0.0f;
    double sx_min = a.bbox[0], sx_max = a.bbox[1], sy_min = a.bbox[2], sy_max = a.bbox[3];
    double tx_min = b.bbox[0], tx_max = b.bbox[1], ty_min = b.bbox[2], ty_max = b.bbox[3];
    float x_dist_ol = (float)(std::min(sx_max, tx_max) - std::max(sx_min, tx_min));
    float y_dist_ol = (float)(std::min(ty_max, sy_max) - std::max(sy_min, ty_min));
    if (x_dist_ol > 0 && y_dist_ol > 0) {
        float area_ol = x_dist_ol * y_dist_ol;
        float area_s = (float)(std::abs(sx_max - sx_min) * std::abs(sy_max - sy_min));
        float area_t = (float)(std::abs(tx_max - tx_min) * std::abs(ty_max - ty_min));
        output = area_ol / (area_s + area_t - area_ol);
    }
    *iou = output;
    return DSSS_OK;
}

// packed record for collectives: [int32 n, N, M, pad][bbox 4 f64][kps kcap][desc kcap*32][geo kcap*2]( + [desc128 kcap*128] under DSSS_DESC_SIFT128)
size_t dsss_features_pack_bytes(const dsss_ctx* c)
{
    size_t K = c->kcap;
    return 16 + 32 + K * sizeof(dsss_kp) + K * 32 + K * 16 + (c->op.descriptor == DSSS_DESC_SIFT128 ? K * 128 : 0);
}
int dsss_features_pack(dsss_ctx* c, int id, void* buf)
{
    if (!c || !buf) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id out of range");
    dsss_frame& f = c->frames[id];
    if (!f.has_feat) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no features", id);
    { int rc = dsss_sync_bboxes(c); if (rc) return rc; }
    size_t K = c->kcap; char* p = (char*)buf;
    int32_t hdr[4] = { f.nkp, f.N, f.M, 0 };
    HIPCHK(c, hipMemcpyAsync(p, hdr, 16, hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(p + 16, f.bbox, 32, hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(p + 48, c->kps + id * K, K * sizeof(dsss_kp), hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(p + 48 + K * sizeof(dsss_kp), c->desc + id * K * 32, K * 32, hipMemcpyDefault, c->stream));
    HIPCHK(c, hipMemcpyAsync(p + 48 + K * sizeof(dsss_kp) + K * 32, c->geo + id * K * 2, K * 16, hipMemcpyDefault, c->stream));
    if (c->op.descriptor == DSSS_DESC_SIFT128) {
        if (!f.has_sift) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no SIFT rows", id);
        HIPCHK(c, hipMemcpyAsync(p + 48 + K * sizeof(dsss_kp) + K * 48, c->desc128 + (size_t)id * K * 128, K * 128, hipMemcpyDefault, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DSSS_OK;
}
int dsss_features_unpack(dsss_ctx* c, int id, const void* buf)
{
    if (!c || !buf) return DSSS_E_ARG;
    size_t K = c->kcap; const char* p = (const char*)buf;
    int32_t hdr[4]; double bbox[4];
    HIPCHK(c, hipMemcpy(hdr, p, 16, hipMemcpyDefault));
    HIPCHK(c, hipMemcpy(bbox, p + 16, 32, hipMemcpyDefault));
    int rc = dsss_features_set(c, id, hdr[1], hdr[2], (const dsss_kp*)(p + 48), (const uint8_t*)(p + 48 + K * sizeof(dsss_kp)),
                               (const double*)(p + 48 + K * sizeof(dsss_kp) + K * 32), bbox, hdr[0]);
    if (rc == DSSS_OK && c->op.descriptor == DSSS_DESC_SIFT128) {
        rc = dsss_ensure_sift_store(c); if (rc) return rc;
        HIPCHK(c, hipMemcpy(c->desc128 + (size_t)id * K * 128, p + 48 + K * sizeof(dsss_kp) + K * 48, K * 128, hipMemcpyDefault));
        c->frames[id].has_sift = true;
    }
    return rc;
}

int dsss_comm_frame_owner(const dsss_ctx* c, int nframes, int frame)
{
    const int world = c ? dsss_comm_world(c) : 1;
    if (nframes <= 0 || frame < 0 || frame >= nframes) return -1;
    int r = (int)(((long long)frame * world) / nframes);
    while (r + 1 < world && (long long)nframes * (r + 1) / world <= frame) ++r;
    while (r > 0 && (long long)nframes * r / world > frame) --r;
    return r;
}
// All frames of a rank in ONE launch each way (the per-frame dsss_features_pack / _unpack -- five copies and a synchronisation
// per frame -- cost 9 ms per step at 200 frames and 8 ranks, more than the all-gather moves in 0.2 ms).  Record layout as in
// dsss_features_pack.  frame -> owner by the block rule of dsss_comm_frame_owner.
__device__ inline int ag_owner(int nframes, int world, int f)
{
    int r = (int)(((long long)f * world) / nframes);
    while (r + 1 < world && (long long)nframes * (r + 1) / world <= f) ++r;
    while (r > 0 && (long long)nframes * r / world > f) --r;
    return r;
}
__global__ __launch_bounds__(256) void ag_copy_kernel(int nframes, int world, int rank, int unpack, char* __restrict__ buf, size_t slice, size_t nb, int K,
                                                      dsss_kp* __restrict__ kps, uint8_t* __restrict__ desc, double* __restrict__ geo, int* __restrict__ nkp,
                                                      int* __restrict__ rows, int* __restrict__ cols, double* __restrict__ bbox, uint8_t* __restrict__ desc128)
{
    const int f = blockIdx.y;
    const int r = ag_owner(nframes, world, f);
    if (unpack ? r == rank : r != rank) return;           // pack: own frames into the own slice; unpack: everybody else's out of theirs
    const int g0 = (int)((long long)nframes * r / world);
    char* rec = buf + (size_t)r * slice + (size_t)(f - g0) * nb;
    const size_t kb = (size_t)K * sizeof(dsss_kp), db = (size_t)K * 32, gb = (size_t)K * 16;     // multiples of 16 (K is a multiple of 64)
    uint4* r4 = reinterpret_cast<uint4*>(rec + 48);
    uint4* k4 = reinterpret_cast<uint4*>(reinterpret_cast<char*>(kps) + (size_t)f * kb);
    uint4* d4 = reinterpret_cast<uint4*>(desc + (size_t)f * db);
    uint4* g4 = reinterpret_cast<uint4*>(reinterpret_cast<char*>(geo) + (size_t)f * gb);
    uint4* s4 = desc128 ? reinterpret_cast<uint4*>(desc128 + (size_t)f * K * 128) : nullptr;
    const size_t nk = kb / 16, nd = db / 16, ng = gb / 16, ns = desc128 ? (size_t)K * 8 : 0, tot = nk + nd + ng + ns;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (size_t)gridDim.x * 256) {
        uint4* own = i < nk ? k4 + i : (i < nk + nd ? d4 + (i - nk) : (i < nk + nd + ng ? g4 + (i - nk - nd) : s4 + (i - nk - nd - ng)));
        if (unpack) *own = r4[i]; else r4[i] = *own;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int* h = reinterpret_cast<int*>(rec); double* bb = reinterpret_cast<double*>(rec + 16);
        if (unpack) { nkp[f] = h[0]; rows[f] = h[1]; cols[f] = h[2]; for (int q = 0; q < 4; ++q) bbox[(size_t)f * 4 + q] = bb[q]; }
        else { h[0] = nkp[f]; h[1] = rows[f]; h[2] = cols[f]; h[3] = 0; for (int q = 0; q < 4; ++q) bb[q] = bbox[(size_t)f * 4 + q]; }
    }
}

int dsss_features_allgather(dsss_ctx* c, int nframes)
{
    if (!c || nframes <= 0 || nframes > c->max_frames) return DSSS_E_ARG;
    const int world = dsss_comm_world(c), rank = dsss_comm_rank(c);
    if (world == 1) return DSSS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const int f0 = (int)((long long)nframes * rank / world), f1 = (int)((long long)nframes * (rank + 1) / world);
    for (int f = f0; f < f1; ++f) if (!c->frames[f].has_feat) DSSS_FAIL(c, DSSS_E_STATE, "frame %d (owned by this rank) has no features", f);
    { int rc = dsss_sync_bboxes(c); if (rc) return rc; }     // the boxes of the own frames are on the device
    const size_t nb = dsss_features_pack_bytes(c);
    int per = 0;
    for (int r = 0; r < world; ++r) per = std::max(per, (int)((long long)nframes * (r + 1) / world - (long long)nframes * r / world));
    const size_t slice = (size_t)per * nb;
    if (c->ag_cap < slice * world) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(c->ag_buf); c->ag_buf = nullptr; c->ag_cap = 0;
        HIPCHK(c, hipMalloc(&c->ag_buf, slice * world)); c->ag_cap = slice * world;
    }
    const size_t hbytes = (size_t)c->max_frames * (3 * sizeof(int) + 4 * sizeof(double)) + 16;
    if (c->ag_host_cap < hbytes) {
        if (c->ag_host) hipHostFree(c->ag_host);
        c->ag_host = nullptr; c->ag_host_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->ag_host, hbytes, hipHostMallocDefault)); c->ag_host_cap = hbytes;
    }
    char* d_buf = static_cast<char*>(c->ag_buf);
    const hipStream_t st = c->stream;
    const dim3 grid(8, nframes);
    uint8_t* d128 = nullptr;
    if (c->op.descriptor == DSSS_DESC_SIFT128) { int r2 = dsss_ensure_sift_store(c); if (r2) return r2; d128 = c->desc128; }
    hipLaunchKernelGGL(ag_copy_kernel, grid, dim3(256), 0, st, nframes, world, rank, 0, d_buf, slice, nb, c->kcap, c->kps, c->desc, c->geo, c->nkp_dev, c->rows_dev, c->cols_dev, c->bbox_dev, d128);
    HIPCHK(c, hipGetLastError());
    int rc = dsss_comm_allgather(c, d_buf, slice, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ag_copy_kernel, grid, dim3(256), 0, st, nframes, world, rank, 1, d_buf, slice, nb, c->kcap, c->kps, c->desc, c->geo, c->nkp_dev, c->rows_dev, c->cols_dev, c->bbox_dev, d128);
    HIPCHK(c, hipGetLastError());
    // the host's view of the gathered frames: counts, sizes and boxes in one download
    int* h_n = static_cast<int*>(c->ag_host); int* h_r = h_n + c->max_frames; int* h_c = h_r + c->max_frames;
    double* h_bb = reinterpret_cast<double*>(h_c + c->max_frames + (c->max_frames & 1));
    HIPCHK(c, hipMemcpyAsync(h_n, c->nkp_dev, sizeof(int) * nframes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h_r, c->rows_dev, sizeof(int) * nframes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h_c, c->cols_dev, sizeof(int) * nframes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(h_bb, c->bbox_dev, sizeof(double) * 4 * nframes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    for (int f = 0; f < nframes; ++f) {
        if (f >= f0 && f < f1) continue;
        dsss_frame& fr = c->frames[f];
        if (h_n[f] < 0 || h_n[f] > c->kcap) DSSS_FAIL(c, DSSS_E_CAPACITY, "gathered frame %d: %d features exceed the per-frame capacity %d", f, h_n[f], c->kcap);
        if (!fr.has_geom) { fr.N = h_r[f]; fr.M = h_c[f]; }
        else if (fr.N != h_r[f] || fr.M != h_c[f]) DSSS_FAIL(c, DSSS_E_ARG, "gathered frame %d: geometry mismatch (%d x %d here, %d x %d at its owner)", f, fr.N, fr.M, h_r[f], h_c[f]);
        fr.nkp = h_n[f]; fr.has_feat = true; fr.has_sift = d128 != nullptr;
        memcpy(fr.bbox, h_bb + (size_t)f * 4, 4 * sizeof(double)); fr.has_bbox = true; fr.bbox_async = false;
    }
    return DSSS_OK;
}

int dsss_profile_enable(dsss_ctx* c, int on) { if (!c) return DSSS_E_ARG; if (!on) dsss_prof_flush(c); c->prof.on = on != 0; return DSSS_OK; }
int dsss_profile_reset(dsss_ctx* c)
{
    if (!c) return DSSS_E_ARG;
    dsss_prof_flush(c);
    for (int i = 0; i < DSSS_K_COUNT; ++i) { c->prof.ms[i] = 0; c->prof.launches[i] = 0; c->prof.work[i] = 0; }
    return DSSS_OK;
}
int dsss_profile_get_work(dsss_ctx* c, double* work)
{
    if (!c || !work) return DSSS_E_ARG;
    for (int i = 0; i < DSSS_K_COUNT; ++i) work[i] = c->prof.work[i];
    return DSSS_OK;
}
int dsss_profile_get(dsss_ctx* c, double* ms, int64_t* launches)
{
    if (!c) return DSSS_E_ARG;
    dsss_prof_flush(c);
    for (int i = 0; i < DSSS_K_COUNT; ++i) { if (ms) ms[i] = c->prof.ms[i]; if (launches) launches[i] = c->prof.launches[i]; }
    return DSSS_OK;
}

} // extern "C"
