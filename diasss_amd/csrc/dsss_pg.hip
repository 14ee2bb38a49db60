// diasss_amd/csrc/dsss_pg.hip -- pose-graph assembly and batch Levenberg-Marquardt solve on the device.
// Replaces the GTSAM NonlinearFactorGraph + iSAM2 of Optimizer::TrajOptimizationAll
// (/root/reference/src/core/optimizer.cpp:101-279): prior on X0 (:164-170), odometry BetweenFactor chain over every
// ping of every frame (:173-200, sigmas :24-28), at most one loop-closure BetweenFactor per target ping
// (:203-258, "last pair wins, first kp in it", score > 0, Diagonal::Variances), initial values DR o noise (:150-160).
// LM schedule = GTSAM LevenbergMarquardtParams() defaults, same loop as oracle/orc_posegraph.c.
//
// Linear algebra per LM trial (all f64 on the device, host only steers):
//   1. per-factor residuals + Jacobians, per-pose 6x6 Hessian blocks (block tridiagonal chain + LC blocks);
//   2. Schur complement of every chain segment between two LC-touched poses ("separators") onto its end points
//      -- segments are independent, one thread each, 6x6 block Thomas recursion;
//   3. the reduced system over the separators (chain couplings + LC blocks) is factorised by a sparse block
//      Cholesky: nested-dissection ordering and symbolic analysis on the host (dsss_pg_sym.cpp, once per solve);
//      small subtrees column by column inside one workgroup each, everything above them MULTIFRONTAL: dense fronts,
//      extend-add of the children's update matrices, panel Cholesky / row solve / trailing update on the f64 matrix cores;
//   4. back-substitution through the segments.
#include "dsss_internal.h"
#include <utility>
#include "dsss_pg_kernels.h"
#include <algorithm>
#include <numeric>
#include <random>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <mutex>
#include <future>

// ------------------------------------------------------------------ host: device memory of one solve
namespace {

// ranges per parallel phase of the analysis: eight for a graph of C3's size (more forks cost what they gain at 23 k separators; 16 gain
// another 10 % on an idle 128-core host), up to twenty-four from a few hundred thousand separators on (C5: 635 k), where a phase is
// milliseconds of work.  The worker pool (dsss_pg_sym.cpp) holds up to 23 threads.
inline int sym_threads(int ns) { const int env = getenv("DSSS_SYM_THREADS") ? atoi(getenv("DSSS_SYM_THREADS")) : 0; if (env > 0) return env;
                                const unsigned hc = std::thread::hardware_concurrency(); return (int)std::min(ns >= 131072 ? 24u : 8u, std::max(1u, hc)); }

struct pg_dev {
    // device memory of one solve comes from the context's arena: a few large chunks that stay allocated between solves,
    // so a solve costs no hipMalloc / hipFree once the arena has grown to its working size
    dsss_ctx* ctx = nullptr;
    static constexpr size_t CHUNK = (size_t)256 << 20;
    template <typename T> int alloc(dsss_ctx* c, T** p, size_t n) {
        if (!ctx) { ctx = c; c->pg_chunk_cur = 0; c->pg_chunk_off = 0; }
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        for (;;) {
            if (c->pg_chunk_cur < c->pg_chunks.size()) {
                auto& ch = c->pg_chunks[c->pg_chunk_cur];
                if (c->pg_chunk_off + bytes <= ch.second) { *p = (T*)((char*)ch.first + c->pg_chunk_off); c->pg_chunk_off += bytes; return DSSS_OK; }
                ++c->pg_chunk_cur; c->pg_chunk_off = 0;
                continue;
            }
            void* q = nullptr; const size_t sz = std::max(bytes, CHUNK);
            HIPCHK(c, hipMalloc(&q, sz));
            c->pg_chunks.push_back({ q, sz });
        }
    }
    template <typename T> int upload(dsss_ctx* c, T** p, const std::vector<T>& v) { int rc = alloc(c, p, v.size()); if (rc) return rc; if (!v.empty()) HIPCHK(c, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return DSSS_OK; }
    // The tables of the analysis (some forty arrays, a few MB) go up as ONE copy: `later` books a slice of a block, `flush` copies
    // the arrays into the context's page-locked staging area on the worker pool, issues one asynchronous upload on `st` and sets the
    // device pointers.  (Forty synchronous copies from pageable memory cost 0.7 ms per solve.)  The vectors must live until flush.
    struct pend { void** p; const void* src; size_t bytes, off; };
    std::vector<pend> pending; size_t pend_total = 0;
    template <typename T> void later(T** p, const std::vector<T>& v) {
        *p = nullptr;
        pending.push_back({ (void**)p, v.data(), v.size() * sizeof(T), pend_total });
        pend_total += (std::max<size_t>(v.size(), 1) * sizeof(T) + 255) & ~(size_t)255;
    }
    hipEvent_t stage_ev = nullptr; bool stage_ev_live = false;      // recorded behind the upload of a flush: the staging area is busy until it fires
    int flush(dsss_ctx* c, hipStream_t st) {
        if (pending.empty()) return DSSS_OK;
        if (stage_ev_live) { hipEventSynchronize(stage_ev); stage_ev_live = false; }
        char* dev = nullptr;
        int rc = alloc(c, &dev, pend_total); if (rc) return rc;
        if (c->pg_stage_cap < pend_total) {
            if (c->pg_stage) hipHostFree(c->pg_stage);
            c->pg_stage = nullptr; c->pg_stage_cap = 0;
            const size_t cap = pend_total + pend_total / 4;
            HIPCHK(c, hipHostMalloc(&c->pg_stage, cap, hipHostMallocDefault));
            c->pg_stage_cap = cap;
        }
        if (stage_ev_live) { hipEventSynchronize(stage_ev); stage_ev_live = false; }      // the previous upload out of the staging area has left it
        char* stage = static_cast<char*>(c->pg_stage);
        const int T = pend_total > ((size_t)1 << 20) ? 4 : 1;
        dsss_pool_run(T, [&](int t) { for (size_t k = t; k < pending.size(); k += T) if (pending[k].bytes) memcpy(stage + pending[k].off, pending[k].src, pending[k].bytes); });
        HIPCHK(c, hipMemcpyAsync(dev, stage, pend_total, hipMemcpyHostToDevice, st));
        if (!stage_ev) stage_ev = event();
        if (stage_ev) { hipEventRecord(stage_ev, st); stage_ev_live = true; }
        for (const pend& q : pending) *q.p = dev + q.off;
        pending.clear(); pend_total = 0;
        return DSSS_OK;
    }
    std::vector<hipEvent_t> events;
    hipEvent_t event() { hipEvent_t e = nullptr; hipEventCreateWithFlags(&e, hipEventDisableTiming); events.push_back(e); return e; }
    void release() { stage_ev_live = false; stage_ev = nullptr; if (ctx) { hipStreamSynchronize(ctx->stream); ctx->pg_chunk_cur = 0; ctx->pg_chunk_off = 0; } for (hipEvent_t e : events) if (e) hipEventDestroy(e); events.clear(); pending.clear(); pend_total = 0; }
};

} // namespace

void dsss_pg_free(dsss_ctx* c) { for (auto& ch : c->pg_chunks) hipFree(ch.first); c->pg_chunks.clear(); c->pg_chunk_cur = 0; c->pg_chunk_off = 0; }

// batch LM over `total` poses with `ne` LC edges (host).  The DR rows (total x 6) are either one host array (dr6) or,
// with dr6 == NULL, the rows of frames 0 .. nframes-1 of the context: read on the host from the frames' pinned copies
// (only the separator poses are looked at) and gathered on the device straight from the frames' device copies.
// `ends` (optional): the (a, b) pairs of the edges packed 8 bytes apart (pg_select_impl)
// Window mode (dsss_posegraph_update_window; c->pg_win_f0 > 0): the chain is the poses of frames f0 .. f0 + nframes - 1 only, pose 0 of it is pinned
// at its estimate of the previous update (the prior's measurement), everything else starts from that estimate where there is one, and the
// result goes back into the context's warm buffer at the window's offset; poses12 / rpy6 then receive the WHOLE trajectory out of that buffer.
static int pg_solve_impl(dsss_ctx* c, const double* dr6, int total, const dsss_lc_edge* edges, int ne, double* poses12, double* stats4, double* rpy6 = nullptr,
                         int nframes = 0, const int* ends = nullptr)
{
    std::vector<int> foff;
    const int f0 = (!dr6 && c->pg_online) ? c->pg_win_f0 : 0, win_p0 = f0 > 0 ? c->pg_win_p0 : 0;
    if (!dr6) { foff.assign(nframes + 1, 0); for (int f = 0; f < nframes; ++f) foff[f + 1] = foff[f] + c->frames[f0 + f].N; }
    const int n = total;
    if (n < 2) DSSS_FAIL(c, DSSS_E_ARG, "pose graph needs at least 2 poses");
    // ranks: contiguous blocks of frames (of poses when the DR chain comes without frames) per partition, contiguous partitions
    // per rank.  A rank owns the poses [mp0, mp1): their chain factors, the LC edges that end in them, their segments.
    const int world = dsss_comm_world(c), rank = dsss_comm_rank(c);
    int nparts = std::max(world, c->pg_parts > 0 ? c->pg_parts : world);
    nparts = std::min(nparts, dr6 ? std::max(1, n / 4) : std::max(1, nframes));
    if (nparts < world) DSSS_FAIL(c, DSSS_E_ARG, "%d ranks need at least %d frames", world, world);
    std::vector<int> pbound(nparts + 1, n);
    for (int p = 0; p < nparts; ++p) pbound[p] = dr6 ? (int)((long long)n * p / nparts) : foff[(int)((long long)nframes * p / nparts)];
    for (int p = 0; p < nparts; ++p) if (pbound[p + 1] <= pbound[p]) DSSS_FAIL(c, DSSS_E_ARG, "empty pose-graph partition %d", p);
    const int part_lo = (int)((long long)nparts * rank / world), part_hi = (int)((long long)nparts * (rank + 1) / world);
    int mp0 = pbound[part_lo], mp1 = pbound[part_hi];                     // (final once the partition boundaries have moved to their cheapest cuts, below)
    const auto T0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    const double PI = DSSS_PI_REF;
    pg_weights W;
    { const double wgt1 = 0.001, wgt2 = 10;                                   // optimizer.cpp:24,28
      const double so[6] = { wgt1 * PI / 180, wgt1 * PI / 180, 0.1 * wgt1 * wgt2 * PI / 180, wgt1 * wgt2, wgt1 * wgt2, wgt1 };
      for (int k = 0; k < 6; ++k) { W.prior[k] = 1.0 / 0.000001; W.odo[k] = 1.0 / so[k]; } }
    // DR poses, measurements and initial values are produced on the device (pg_init_kernel) further down
    std::vector<int> ea(ne), eb(ne), eo(ne); std::vector<pose_t> emeas; std::vector<double> ew;      // (the measurements' 1.7 MB are allocated where they are filled, beside the analysis: touching fresh pages here is time the GPU waits for)
    {   // the end points first: they are all the analysis needs (the measurements are unpacked beside it, below).  The records are 152 bytes
        // apart: at C5 (351 k edges, 53 MB) one thread spent 3.6 ms here before the analysis could start, so large edge sets go by ranges
        const int TE = ne >= 65536 ? 8 : 1;
        std::vector<int> bad(TE, -1);
        dsss_pool_run(TE, [&](int t) {
            const int e0 = (int)((long long)ne * t / TE), e1 = (int)((long long)ne * (t + 1) / TE);
            for (int e = e0; e < e1; ++e) {
                if (ends) { ea[e] = ends[2 * (size_t)e]; eb[e] = ends[2 * (size_t)e + 1]; } else { ea[e] = edges[e].a; eb[e] = edges[e].b; }
                eo[e] = std::max(ea[e], eb[e]);
                if ((ea[e] < 0 || ea[e] >= n || eb[e] < 0 || eb[e] >= n || ea[e] == eb[e]) && bad[t] < 0) bad[t] = e;
            }
        });
        for (int t = 0; t < TE; ++t) if (bad[t] >= 0) DSSS_FAIL(c, DSSS_E_ARG, "LC edge %d out of range", bad[t]);
    }
    // Two levels of chain elimination.  TRUE separators (the unknowns of the sparse factorisation): LC-touched poses, the first and
    // the last pose, the last pose of every partition.  CHUNK ends: every PG_CHUNK-th pose as well, which bounds the sequential
    // depth of the per-segment block-Thomas recursion (one thread per segment).  Pass 1 condenses every chunk onto its two ends
    // (poses -> "level-1" chain of true separators + chunk ends); pass 2 condenses the runs of chunk ends between two true
    // separators the same way (same kernel, on the level-1 chain).  Exact: only the elimination order changes.
    // (All of this in time proportional to the separators, not to the poses: three passes over 400 k poses cost a millisecond of the
    // solve's serial host preparation.)  True separators are marked in a bit set and read back in order.
    const double t_p1 = ms_since(T0);
    std::vector<unsigned long long> tbits(((size_t)n + 63) / 64, 0ull);
    auto mark = [&](int i) { tbits[(size_t)i >> 6] |= 1ull << (i & 63); };
    mark(0); mark(n - 1);
    for (int e = 0; e < ne; ++e) { mark(ea[e]); mark(eb[e]); }
    if (nparts > 1 && ne > 0) {
        // Where the partitions end (round 5).  The boundaries used to be frame boundaries; the interface of the partitioned solve -- the
        // separators every rank factorises again, summed by the all-reduce -- is the set of poses whose loop closures cross a boundary, and
        // how many cross depends on WHERE the chain is cut (dsss_pg_sym.cpp, chain_cut: between 1 and 200 at C3).  Every boundary moves to
        // the cheapest gap between two loop-closure poses within a third of a partition's length of its frame boundary: cost of a gap = loop
        // closures that span it, all gaps priced by one difference array.  Every rank holds all edges (exchanged before the solve), so every
        // rank moves the boundaries to the same places.  Frames, features and matches stay sharded by frame: only pose ownership moves.
        std::vector<int> lc_ends; lc_ends.reserve((size_t)2 * ne + 2);
        for (size_t w = 0; w < tbits.size(); ++w)
            for (unsigned long long bits = tbits[w]; bits; bits &= bits - 1) lc_ends.push_back((int)(w * 64) + __builtin_ctzll(bits));
        std::vector<int> pre(tbits.size() + 1, 0);
        for (size_t w = 0; w < tbits.size(); ++w) pre[w + 1] = pre[w] + __builtin_popcountll(tbits[w]);
        auto eidx = [&](int pose) { return pre[(size_t)pose >> 6] + __builtin_popcountll(tbits[(size_t)pose >> 6] & ((1ull << (pose & 63)) - 1ull)); };
        const int m = (int)lc_ends.size();
        std::vector<int> cross(m + 1, 0);                                  // cross[i]: loop closures that span the gap between lc_ends[i - 1] and lc_ends[i]
        for (int e = 0; e < ne; ++e) { const int lo = eidx(std::min(ea[e], eb[e])), hi = eidx(std::max(ea[e], eb[e])); cross[lo + 1]++; cross[hi + 1]--; }
        for (int i = 1; i <= m; ++i) cross[i] += cross[i - 1];
        const int width = n / nparts / 3;
        for (int p = 1; p < nparts; ++p) {
            const int target = pbound[p];
            int best = -1; long long bcost = 1LL << 60, bdist = 0;
            const int i0 = (int)(std::lower_bound(lc_ends.begin(), lc_ends.end(), target - width) - lc_ends.begin());
            for (int i = std::max(i0, 1); i < m && lc_ends[i - 1] + 1 <= target + width; ++i) {      // boundary between lc_ends[i - 1] and lc_ends[i]: partition p starts at lc_ends[i - 1] + 1
                const int start = lc_ends[i - 1] + 1;
                if (start <= pbound[p - 1] || start < target - width) continue;
                const long long d = std::llabs((long long)start - target);
                if (cross[i] < bcost || (cross[i] == bcost && d < bdist)) { bcost = cross[i]; bdist = d; best = start; }
            }
            if (best > pbound[p - 1] && best < n) pbound[p] = best;
        }
        for (int p = 0; p < nparts; ++p) if (pbound[p + 1] <= pbound[p]) DSSS_FAIL(c, DSSS_E_ARG, "empty pose-graph partition %d", p);
        mp0 = pbound[part_lo]; mp1 = pbound[part_hi];
    }
    for (int p = 1; p < nparts; ++p) mark(pbound[p] - 1);                // a partition ends on a separator: segments never straddle ranks
    const int chunk = 16;                                                // (8 and 24 reach the same optimum; measured flat in round 2)
    // pass 2 is sequential over the chunk ends between two true separators: a gap of more than 16 chunks (frame boundaries
    // without keypoints reach 900 poses) gets true separators of its own, the first chunk end at least 16 chunks after the last one
    const int run = 16 * chunk;
    std::vector<int> sep1, sep_pose, t2;                                 // level-1 chain (poses); true separators (poses; positions in sep1)
    sep_pose.reserve((size_t)2 * ne + nparts + n / run + 8);
    {
        int last = 0;
        for (size_t w = 0; w < tbits.size(); ++w)
            for (unsigned long long bits = tbits[w]; bits; bits &= bits - 1) {
                const int b = (int)(w * 64) + __builtin_ctzll(bits);
                for (;;) {                                               // fill the gap (last, b)
                    const long long nx = ((long long)last + run + chunk - 1) / chunk * chunk;
                    if (nx >= b) break;
                    sep_pose.push_back((int)nx); last = (int)nx;
                }
                sep_pose.push_back(b); last = b;
            }
    }
    const double t_q1 = ms_since(T0);
    // index of a true separator among the true separators = number of marked poses below it: word prefix + popcount (a binary search
    // per loop-closure end point cost 0.6 ms)
    for (int v : sep_pose) mark(v);                                      // the gap fillers too
    std::vector<int> tpre(tbits.size() + 1, 0);
    for (size_t w = 0; w < tbits.size(); ++w) tpre[w + 1] = tpre[w] + __builtin_popcountll(tbits[w]);
    auto sidx = [&](int pose) { return tpre[(size_t)pose >> 6] + __builtin_popcountll(tbits[(size_t)pose >> 6] & ((1ull << (pose & 63)) - 1ull)); };
    const int ns = (int)sep_pose.size(), nseg = ns - 1;
    std::vector<std::pair<int, int>> redges;
    redges.reserve((size_t)ns + ne);
    const double t_q3 = ms_since(T0);
    const double t_p2 = ms_since(T0);                   // (the reduced edges themselves are written by the analysis thread, first thing)
    const auto T1 = std::chrono::steady_clock::now();
    // The analysis of the reduced system runs on a host thread of its own while this thread sets up everything that does not depend
    // on it -- device arrays of the pose chain, initial values, the first linearisation and the chain part of the first LM trial
    // (assembly, both segment passes): the device works through those while the host orders and analyses.
    std::vector<double> cx(ns), cy(ns);                       // separator coordinates: filled below, read by the analysis after its adjacency pass
    pg_sym S;
    pg_sched SO, SI;
    std::vector<int> sym_part(ns);
    { int p = 0; for (int k = 0; k < ns; ++k) { while (p + 1 < nparts && sep_pose[k] >= pbound[p + 1]) ++p; sym_part[k] = p; } }      // (the separators ascend: one sweep)
    std::promise<void> bottom_prom; std::future<void> bottom_fut = bottom_prom.get_future();
    std::promise<void> lists_prom; std::future<void> lists_fut = lists_prom.get_future(); bool lists_signalled = false;
    std::promise<void> coords_prom; std::future<void> coords_fut = coords_prom.get_future();
    bool bottom_signalled = false;
    // (Rounds 4 - 5 also had the ordering on the device, dsss_pg_nd.hip: coordinate medians only.  The chain-order cut of round 5 -- 12 panel
    // levels instead of 29 at C3 -- exists on the host only, the device kernels were off by default from then on and were removed in round 6.)
    // RANK-LOCAL ANALYSIS (round 6; several ranks).  Rounds 2 - 5 had every rank order and analyse the WHOLE reduced graph -- 2.6 ms of serial
    // host work per solve at C3 whatever the number of GPUs, and arenas for everybody's fronts.  Now a rank analyses its OWN separators plus
    // the interface: I = the separators with a neighbour on a higher rank (exactly the nodes whose diagonal blocks take contributions from
    // two ranks -- the ownership rule of the numeric phase; every edge between two ranks has its lower end in I, so without I the ranks'
    // interiors do not touch).  Every rank holds all edges, so every rank finds the same I.  The local graph is ordered by the nested
    // dissection with I PRESCRIBED last as one dense front (pg_sym_opts::iface_last): a rank eliminates its interior, what is left on I
    // is its share of the reduced Hessian, the ranks' interface fronts -- identical in layout -- are summed IN PLACE by the all-reduce, and
    // every rank factorises the sum and substitutes back into its interior.  No structure is exchanged at all.  Kernels index separators
    // and values globally (chain order), so the local tables are translated once: perm_g / dest_g / ifslot_g / ifsep_g and the fronts'
    // value references.  A dense interface of more than PG_LOCAL_IFACE_MAX nodes (or DSSS_PG_LOCAL=0) keeps the replicated analysis with
    // its interface TREE; so does one rank holding several partitions (world == 1: the tests of the partition logic).
    bool local = false;
    std::vector<int> perm_g, dest_g, ifslot_g, ifsep_g, loc_of, glob_of, ledge_g;
    std::vector<std::pair<int, int>> ledges;
    std::vector<double> cxl, cyl;
    std::thread sym_thread([&] {
        pg_sym_opts opt; opt.threads = sym_threads(ns);
        opt.on_bottom_ready = [&] { bottom_signalled = true; bottom_prom.set_value(); };
        opt.on_lists_ready = [&] { lists_signalled = true; lists_prom.set_value(); };
        opt.before_order = [&] { coords_fut.wait(); };
        opt.lists_on_device = true;                          // (the bins' update lists, update map and root-boundary indices: built on the device, below)
        for (int k = 0; k + 1 < ns; ++k) redges.push_back({ k, k + 1 });      // the reduced graph: the chain of the separators, then the loop closures
        for (int e = 0; e < ne; ++e) redges.push_back({ sidx(ea[e]), sidx(eb[e]) });
        const double bin_cost = getenv("DSSS_PG_BIN_COST") ? atof(getenv("DSSS_PG_BIN_COST")) : 600;   // ~ update-list iterations + 20 per column; measured optimum at C3 (500-700)
        opt.bin_cost = bin_cost; pg_sym_opts_env(opt);
        if (world > 1 && !(getenv("DSSS_PG_LOCAL") && atoi(getenv("DSSS_PG_LOCAL")) == 0)) {
            std::vector<int> rank_of_part(nparts, 0);
            for (int r = 0; r < world; ++r) for (int p = (int)((long long)nparts * r / world); p < (int)((long long)nparts * (r + 1) / world); ++p) rank_of_part[p] = r;
            std::vector<char> isif(ns, 0);
            for (const auto& e : redges) {
                const int ra = rank_of_part[sym_part[e.first]], rb = rank_of_part[sym_part[e.second]];
                if (ra < rb) isif[e.first] = 1; else if (rb < ra) isif[e.second] = 1;
            }
            int nif_l = 0; for (int k = 0; k < ns; ++k) nif_l += isif[k];
            if (nif_l <= PG_LOCAL_IFACE_MAX) {
                local = true;
                loc_of.assign(ns, -1);
                for (int k = 0; k < ns; ++k)
                    if (isif[k] || rank_of_part[sym_part[k]] == rank) { loc_of[k] = (int)glob_of.size(); if (isif[k]) opt.iface_last.push_back(loc_of[k]); glob_of.push_back(k); }
                for (size_t g = 0; g < redges.size(); ++g) {
                    const int a = loc_of[redges[g].first], b = loc_of[redges[g].second];
                    if (a >= 0 && b >= 0 && a != b) { ledges.push_back({ a, b }); ledge_g.push_back((int)g); }
                }
            }
        }
        if (local) {
            const int nsl = (int)glob_of.size();
            cxl.resize(nsl); cyl.resize(nsl);
            opt.before_order = [&] { coords_fut.wait(); for (size_t i = 0; i < glob_of.size(); ++i) { cxl[i] = cx[glob_of[i]]; cyl[i] = cy[glob_of[i]]; } };
            opt.on_bottom_ready = nullptr; opt.on_lists_ready = nullptr;      // (nothing goes up early: the tables below come last)
            // a rank's share of the bins leaves most of the chip empty, and the bins kernel lasts as long as its longest bin: shorter bins, more
            // of the tree in the fronts (the slowest rank of 8 at C3: 12.2 -> 11.5 ms; on one GPU, where 877 bins fill the chip, 600 is the optimum)
            if (!getenv("DSSS_PG_BIN_COST") && nsl < 16384) opt.bin_cost = 120;      // (a rank of C5 has thousands of bins of its own: the one-GPU optimum holds there)
            opt.threads = sym_threads(nsl);
            pg_symbolic(nsl, ledges, 0, cxl.data(), cyl.data(), nullptr, 1, opt, S);
            pg_build_schedule(S, 0, 1, SO);
            pg_build_schedule(S, -1, 0, SI);
            const int nval_g = ns + (int)redges.size();
            perm_g.assign(ns, -1); dest_g.assign(nval_g, -1); ifslot_g.assign(ns, -1);
            for (int i = 0; i < nsl; ++i) { perm_g[glob_of[i]] = S.perm[i]; dest_g[glob_of[i]] = S.dest_bin[i]; }
            for (size_t le = 0; le < ledges.size(); ++le) dest_g[ns + ledge_g[le]] = S.dest_bin[nsl + le];
            for (int& v : S.fa_src) v = v >= S.nval ? nval_g + (v - S.nval) : (v < nsl ? glob_of[v] : ns + ledge_g[v - nsl]);
            for (size_t q = 0; q < S.iface_seps.size(); ++q) { const int k = glob_of[S.iface_seps[q]]; ifsep_g.push_back(k); ifslot_g[k] = (int)q; }
        } else {
            // ONE rank, one partition: the analysis BY PARTS (pg_symbolic_parts, round 6).  The phases of pg_symbolic gain nothing from threads at
            // C3's size (one thread 4.0 ms, eight 3.4: a dozen fork / joins around 0.1 - 0.3 ms of work each), whole parts do: the chain order is
            // cut into K parts where few loop closures cross (the gap with the fewest spanning loop closures within a third of a part of the
            // equal-count position: one difference array over the separators prices them all), every part is ordered and analysed on its own
            // thread with the interface between the parts as the last dense front, and the tables are joined.
            bool by_parts = false;
            // (parts of about a thousand separators, at most 8, up to C3's size -- C2: 3 292 separators in 3 parts, step 7.6 -> 6.6 ms --; 16 from 64 k separators on)
            int K = getenv("DSSS_PG_PARTS_ANALYSIS") ? atoi(getenv("DSSS_PG_PARTS_ANALYSIS")) : (ns >= 1500 ? (ns < 65536 ? std::min(8, std::max(2, ns / 1000)) : 16) : 0);
            if (!getenv("DSSS_PG_PARTS_ANALYSIS")) { const int hw = (int)std::thread::hardware_concurrency(); K = hw >= 4 ? std::min(K, hw) : 0; }      // (the parts need threads of their own: on a host with fewer than four the one graph is the shorter analysis)
            if (nparts == 1 && K >= 2 && ne > 0) {
                K = std::min(K, ns / 8);
                std::vector<int> cross(ns + 1, 0), vpart(ns, 0);
                for (size_t g = (size_t)ns - 1; g < redges.size(); ++g) { const int lo = std::min(redges[g].first, redges[g].second), hi = std::max(redges[g].first, redges[g].second); cross[lo + 1]++; cross[hi + 1]--; }
                for (int k = 1; k <= ns; ++k) cross[k] += cross[k - 1];      // cross[g]: loop closures that span the gap between separators g - 1 and g
                int prev = 0, p_cur = 0;
                std::vector<int> starts;
                for (int p = 1; p < K; ++p) {
                    const int target = (int)((long long)ns * p / K), width = ns / K / 3;
                    int best = -1, bcost = 1 << 30, bdist = 0;
                    for (int g = std::max(target - width, prev + 1); g <= std::min(target + width, ns - 1); ++g) {
                        const int d = std::abs(g - target);
                        if (cross[g] < bcost || (cross[g] == bcost && d < bdist)) { bcost = cross[g]; bdist = d; best = g; }
                    }
                    if (best > prev && bcost <= std::max(PG_PARTS_CUT_MAX, ns / 16384)) { starts.push_back(best); prev = best; }      // (an expensive boundary is left out: its two parts stay one)
                }
                for (int k = 0; k < ns; ++k) { while (p_cur < (int)starts.size() && k >= starts[p_cur]) ++p_cur; vpart[k] = p_cur; }
                if (!starts.empty()) by_parts = pg_symbolic_parts(ns, redges, nseg, cx.data(), cy.data(), vpart.data(), (int)starts.size() + 1, PG_PARTS_IFACE_MAX, opt, S);
            }
            if (!by_parts) pg_symbolic(ns, redges, nseg, cx.data(), cy.data(), nparts > 1 ? sym_part.data() : nullptr, nparts, opt, S);
            // launch lists: this rank's interior fronts, then (after the all-reduce) the replicated interface fronts
            pg_build_schedule(S, part_lo, part_hi, SO);
            if (nparts > 1) pg_build_schedule(S, -1, 0, SI);
        }
        if (!lists_signalled) lists_prom.set_value();
        if (!bottom_signalled) bottom_prom.set_value();      // (several partitions: nothing is ready early)
    });
    struct pg_joiner { std::thread& t; ~pg_joiner() { if (t.joinable()) t.join(); } } sym_join{ sym_thread };      // every return path waits for the thread before its data goes away
    struct pg_coords_guard { std::promise<void>& p; bool done = false; void set() { if (!done) { done = true; p.set_value(); } } ~pg_coords_guard() { set(); } } coords_guard{ coords_prom };
    // device state
    pg_dev dv;
    int rc = DSSS_OK;
    // every error exit: let the analysis thread go, wait for it AND for what it queued on the ordering's stream, and only then hand the
    // arena back (released first, the next solve could reuse memory the ordering's kernels of this one still write)
    auto abandon = [&] {
        coords_guard.set();
        if (sym_thread.joinable()) sym_thread.join();
        dv.release();
    };
#define TRY(x) do { rc = (x); if (rc) { abandon(); return rc; } } while (0)
    // DR rows on the device first: the separator coordinates for the ordering come back from there (host reads of the
    // frames' pinned copies are slow).  The analysis thread is started BEFORE they are back and before the rest of this thread's
    // preparation (the level-1 chain of the device's chain condensation, the segment orders): it builds its adjacency first and waits for
    // the coordinates where it first needs them (pg_sym_opts::before_order).
    double* d_dr6; double* d_sxy; int* d_sep; int* d_sep1; int* d_t2; int* d_ord1; int* d_ord2;
    TRY(dv.alloc(c, &d_dr6, (size_t)n * 6)); TRY(dv.alloc(c, &d_sxy, (size_t)ns * 2)); dv.later(&d_sep, sep_pose);
    std::vector<int> ord1, ord2;                            // (alive until the second flush below)
    std::vector<unsigned long long> fp;
    unsigned long long* d_fp = nullptr; int* d_foff = nullptr;
    std::vector<double> sxy((size_t)ns * 2);
    {
        if (!dr6) {
            fp.resize(nframes);
            for (int f = 0; f < nframes; ++f) fp[f] = (unsigned long long)(uintptr_t)c->frames[f0 + f].pose6;
            dv.later(&d_fp, fp); dv.later(&d_foff, foff);
        }
        TRY(dv.flush(c, c->stream));                          // true separators, frame pointers: one upload
        hipError_t e = hipSuccess;
        if (dr6) e = hipMemcpyAsync(d_dr6, dr6, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice, c->stream);
        else {      // one gather launch over the frames' device copies (a device-to-device copy per frame cost 0.5 ms of launches at 200 frames)
            hipLaunchKernelGGL(pg_gather_dr_kernel, dim3(8, nframes), dim3(256), 0, c->stream, d_fp, d_foff, d_dr6);
            e = hipGetLastError();
        }
        if (e == hipSuccess) { hipLaunchKernelGGL(pg_sep_xy_kernel, dim3((ns + 255) / 256), dim3(256), 0, c->stream, ns, d_sep, d_dr6, d_sxy); e = hipGetLastError(); }
        if (e != hipSuccess) { abandon(); HIPCHK(c, e); }
    }
    const double t_prep0 = ms_since(T0);
    {   // the coordinates come back while the analysis builds its adjacency: hand them over
        hipError_t e = hipMemcpyAsync(sxy.data(), d_sxy, sxy.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // the library's stream does not synchronise with the null stream
        if (e != hipSuccess) { abandon(); HIPCHK(c, e); }
        for (int k = 0; k < ns; ++k) { cx[k] = sxy[2 * (size_t)k]; cy[k] = sxy[2 * (size_t)k + 1]; }
        coords_guard.set();
    }
    emeas.resize(ne); ew.resize((size_t)ne * 6);
    {   // measurements and weights of the loop closures (the analysis is running and has its coordinates); large edge sets by ranges, as above
        const int TE = ne >= 65536 ? 4 : 1;
        std::vector<int> bad_var(TE, -1), bad_rel(TE, -1);
        dsss_pool_run(TE, [&](int t) {
            const int e0 = (int)((long long)ne * t / TE), e1 = (int)((long long)ne * (t + 1) / TE);
            for (int e = e0; e < e1; ++e) {
                for (int k = 0; k < 9; ++k) emeas[e].R[k] = edges[e].rel[k];
                for (int k = 0; k < 3; ++k) emeas[e].t[k] = edges[e].rel[9 + k];
                for (int k = 0; k < 6; ++k) {
                    if ((!(edges[e].var[k] > 0) || !std::isfinite(edges[e].var[k])) && bad_var[t] < 0) bad_var[t] = e * 6 + k;
                    ew[(size_t)e * 6 + k] = 1.0 / std::sqrt(edges[e].var[k]);
                }
                for (int k = 0; k < 12; ++k) if (!std::isfinite(edges[e].rel[k]) && bad_rel[t] < 0) bad_rel[t] = e;
            }
        });
        for (int t = 0; t < TE; ++t) {
            if (bad_var[t] >= 0) { abandon(); DSSS_FAIL(c, DSSS_E_NUMERIC, "LC edge %d: variance %d is not finite and positive", bad_var[t] / 6, bad_var[t] % 6); }
            if (bad_rel[t] >= 0) { abandon(); DSSS_FAIL(c, DSSS_E_NUMERIC, "LC edge %d: relative pose is not finite", bad_rel[t]); }
        }
    }
    // level-1 chain = true separators merged with the chunk ends 0, chunk, 2 chunk ...; segment orders: only the device reads them
    const double t_m0 = ms_since(T0);
    sep1.reserve(sep_pose.size() + n / chunk + 2); t2.reserve(sep_pose.size());
    for (size_t it = 0, m = 0; it < sep_pose.size() || m < (size_t)n;) {
        const long long a = it < sep_pose.size() ? sep_pose[it] : (1LL << 40), bm = m < (size_t)n ? (long long)m : (1LL << 40);
        if (a <= bm) { t2.push_back((int)sep1.size()); sep1.push_back((int)a); ++it; if (a == bm) m += chunk; }
        else { sep1.push_back((int)bm); m += chunk; }
    }
    const double t_q2 = ms_since(T0) - t_m0;
    const int ns1 = (int)sep1.size(), nseg1 = ns1 - 1;
    // this rank's range of the level-1 chain (its poses are [mp0, mp1))
    const int kp0 = (int)(std::lower_bound(sep1.begin(), sep1.end(), mp0) - sep1.begin()), kp1 = (int)(std::lower_bound(sep1.begin(), sep1.end(), mp1) - sep1.begin());
    dv.later(&d_sep1, sep1); dv.later(&d_t2, t2);
    {   // segments of both passes in descending order of length (stable counting sort: ties stay in chain order)
        auto by_length = [](const std::vector<int>& ends, std::vector<int>& ord) {
            const int m = (int)ends.size() - 1;
            ord.resize(std::max(m, 1));
            int maxlen = 0;
            for (int k = 0; k < m; ++k) maxlen = std::max(maxlen, ends[k + 1] - ends[k]);
            std::vector<int> cnt(maxlen + 2, 0);
            for (int k = 0; k < m; ++k) cnt[maxlen - (ends[k + 1] - ends[k]) + 1]++;
            for (int l = 0; l <= maxlen; ++l) cnt[l + 1] += cnt[l];
            for (int k = 0; k < m; ++k) ord[cnt[maxlen - (ends[k + 1] - ends[k])]++] = k;
        };
        by_length(sep1, ord1); by_length(t2, ord2);
        dv.later(&d_ord1, ord1); dv.later(&d_ord2, ord2);
    }
    TRY(dv.flush(c, c->stream));                              // level-1 chain and segment orders: one upload
    const double t_prep = ms_since(T0);
    if (getenv("DSSS_PG_VERBOSE")) fprintf(stderr, "[dsss pg prep] edges %.2f ms, separators %.2f ms (bits %.2f mid %.2f redges %.2f), coordinates launched at %.2f ms; beside the analysis: level-1 chain %.2f ms, rest %.2f ms\n", t_p1, t_p2 - t_p1, t_q1 - t_p1, t_q3 - t_q1, t_p2 - t_q3, t_prep0, t_q2, t_prep - t_prep0 - t_q2);
    const bool verbose = getenv("DSSS_PG_VERBOSE") != nullptr;

    // incidence lists of the poses (edge order): only the device kernels read them, so they are built while the analysis runs
    std::vector<int> adj_ptr(n + 1, 0), adj_edge(2 * (size_t)ne);
    for (int e = 0; e < ne; ++e) { adj_ptr[ea[e] + 1]++; adj_ptr[eb[e] + 1]++; }
    for (int i = 0; i < n; ++i) adj_ptr[i + 1] += adj_ptr[i];
    { std::vector<int> fill(adj_ptr.begin(), adj_ptr.end() - 1);
      for (int e = 0; e < ne; ++e) { adj_edge[fill[ea[e]]++] = e << 1; adj_edge[fill[eb[e]]++] = (e << 1) | 1; } }
    // loop closures that repeat an unordered pose pair (see pg_scatter_lc_kernel).  The pipeline's edges -- a < b, b strictly
    // ascending -- cannot: they skip the sort
    std::vector<int> lc_link;
    {
        bool plain = true;
        for (int e = 0; e < ne && plain; ++e) plain = ea[e] < eb[e] && (e == 0 || eb[e - 1] < eb[e]);
        if (!plain) {
            std::vector<std::pair<unsigned long long, int>> key(ne);
            for (int e = 0; e < ne; ++e) key[e] = { ((unsigned long long)(unsigned)std::min(ea[e], eb[e]) << 32) | (unsigned)std::max(ea[e], eb[e]), e };
            std::sort(key.begin(), key.end());                           // (pair, edge index): the members of a group in edge order
            bool dups = false;
            for (int k = 0; k + 1 < ne && !dups; ++k) dups = key[k].first == key[k + 1].first;
            if (dups) {
                lc_link.assign((size_t)2 * ne, -1);
                for (int k = 0; k < ne; ++k) {
                    lc_link[2 * (size_t)key[k].second] = (k == 0 || key[k - 1].first != key[k].first) ? 1 : 0;
                    if (k + 1 < ne && key[k + 1].first == key[k].first) lc_link[2 * (size_t)key[k].second + 1] = key[k + 1].second;
                }
            }
        }
    }
    // ---- early device set-up (nothing here reads S)
    pose_t *d_X, *d_Xn, *d_meas, *d_emeas; int *d_ea, *d_eb, *d_eo, *d_adj_ptr, *d_adj_edge, *d_perm;
    double *d_ew, *d_r, *d_Ji, *d_D, *d_C, *d_g, *d_delta, *d_E, *d_Dl, *d_gi, *d_sDL, *d_sDR, *d_sGL, *d_sGR, *d_sS, *d_L, *d_x, *d_part, *d_scal;
    double *d_F, *d_R, *d_ubin, *d_aval;
    int* d_binperm = nullptr;
    int *d_colptr, *d_rowidx, *d_rlptr, *d_rlcol, *d_rlpos, *d_rlrow, *d_binptr, *d_bincols, *d_dest, *d_fail, *d_map; long long* d_mapptr;
    int *d_binroot_ptr, *d_binroot_idx, *d_broot_b, *d_broot_of_col, *d_anc_first, *d_anc_rel, *d_rel, *d_fa_src, *d_fa_col, *d_fa_tr, *d_frows, *d_xr_ptr, *d_xr_child, *d_xr_row, *d_fa_rowptr;
    long long* d_broot_uoff; pg_front* d_FD; pg_child* d_CH;
    double* d_red;
    const int nf = n + ne, nblk = (nf + 255) / 256;
    TRY(dv.alloc(c, &d_X, n)); TRY(dv.alloc(c, &d_Xn, n)); TRY(dv.alloc(c, &d_meas, n)); TRY(dv.upload(c, &d_emeas, emeas));
    TRY(dv.upload(c, &d_ea, ea)); TRY(dv.upload(c, &d_eb, eb)); TRY(dv.upload(c, &d_eo, eo)); TRY(dv.upload(c, &d_ew, ew));
    TRY(dv.upload(c, &d_adj_ptr, adj_ptr)); TRY(dv.upload(c, &d_adj_edge, adj_edge));
    int* d_lc_link = nullptr; if (!lc_link.empty()) TRY(dv.upload(c, &d_lc_link, lc_link));
    TRY(dv.alloc(c, &d_r, (size_t)nf * 6)); TRY(dv.alloc(c, &d_Ji, (size_t)nf * 36));
    // a second set of residuals and Jacobians: the linearisation that measures a trial's error at X (+) delta IS the next iteration's
    // linearisation when the trial is accepted (same kernel, same point, same bits) -- it writes them here and the sets swap
    double *d_r2, *d_Ji2; TRY(dv.alloc(c, &d_r2, (size_t)nf * 6)); TRY(dv.alloc(c, &d_Ji2, (size_t)nf * 36));
    TRY(dv.alloc(c, &d_D, (size_t)n * 36)); TRY(dv.alloc(c, &d_C, (size_t)n * 36)); TRY(dv.alloc(c, &d_g, (size_t)n * 6)); TRY(dv.alloc(c, &d_delta, (size_t)n * 6));
    TRY(dv.alloc(c, &d_E, (size_t)n * 36)); TRY(dv.alloc(c, &d_Dl, (size_t)n * 36)); TRY(dv.alloc(c, &d_gi, (size_t)n * 6));
    TRY(dv.alloc(c, &d_sDL, (size_t)nseg1 * 36)); TRY(dv.alloc(c, &d_sDR, (size_t)nseg1 * 36)); TRY(dv.alloc(c, &d_sGL, (size_t)nseg1 * 6));
    TRY(dv.alloc(c, &d_sGR, (size_t)nseg1 * 6)); TRY(dv.alloc(c, &d_sS, (size_t)nseg1 * 36));
    // level-1 chain (true separators + chunk ends) and its condensation onto the true separators (pass 2)
    double *d_D1, *d_C1, *d_g1, *d_E1, *d_Dl1, *d_gi1, *d_delta1, *d_s2DL, *d_s2DR, *d_s2GL, *d_s2GR, *d_s2S;
    TRY(dv.alloc(c, &d_D1, (size_t)ns1 * 36)); TRY(dv.alloc(c, &d_C1, (size_t)ns1 * 36)); TRY(dv.alloc(c, &d_g1, (size_t)ns1 * 6)); TRY(dv.alloc(c, &d_delta1, (size_t)ns1 * 6));
    TRY(dv.alloc(c, &d_E1, (size_t)ns1 * 36)); TRY(dv.alloc(c, &d_Dl1, (size_t)ns1 * 36)); TRY(dv.alloc(c, &d_gi1, (size_t)ns1 * 6));
    TRY(dv.alloc(c, &d_s2DL, (size_t)std::max(nseg, 1) * 36)); TRY(dv.alloc(c, &d_s2DR, (size_t)std::max(nseg, 1) * 36)); TRY(dv.alloc(c, &d_s2GL, (size_t)std::max(nseg, 1) * 6));
    TRY(dv.alloc(c, &d_s2GR, (size_t)std::max(nseg, 1) * 6)); TRY(dv.alloc(c, &d_s2S, (size_t)std::max(nseg, 1) * 36));
    double* d_rdiag; TRY(dv.alloc(c, &d_rdiag, (size_t)ns * 6));      // reciprocal diagonals of the binned columns' pivots (forward -> backward substitution)
    TRY(dv.alloc(c, &d_x, (size_t)ns * 6)); TRY(dv.alloc(c, &d_part, (size_t)nblk)); TRY(dv.alloc(c, &d_scal, 8)); TRY(dv.alloc(c, &d_fail, 1)); TRY(dv.alloc(c, &d_red, 8));
    hipStream_t st = c->stream;
#define HCK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { c->err = std::string(#x) + ": " + hipGetErrorString(_e); abandon(); return DSSS_E_HIP; } } while (0)
    // sums over the factors are partial on every rank: one small all-reduce makes them global (and identical everywhere)
    auto reduce_scalars = [&](double* host3, int* failed) -> int {
        if (world > 1) {
            hipLaunchKernelGGL(pg_comm_scal_kernel, dim3(1), dim3(64), 0, st, d_scal, d_fail, d_red);
            int rc2 = dsss_comm_allreduce(c, d_red, 4, st); if (rc2) { abandon(); return rc2; }
            double h4[4];
            HCK(hipMemcpyAsync(h4, d_red, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
            HCK(hipStreamSynchronize(st));
            host3[0] = h4[0]; host3[1] = h4[1]; host3[2] = h4[2]; *failed = h4[3] != 0.0;
        } else {
            // into page-locked memory: a copy to the caller's stack is staged by the runtime and waits for it twice per trial
            if (!c->pg_scal_host) HCK(hipHostMalloc((void**)&c->pg_scal_host, 8 * sizeof(double), hipHostMallocDefault));
            HCK(hipMemcpyAsync(c->pg_scal_host, d_scal, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
            HCK(hipMemcpyAsync(c->pg_scal_host + 4, d_fail, sizeof(int), hipMemcpyDeviceToHost, st));
            HCK(hipStreamSynchronize(st));
            host3[0] = c->pg_scal_host[0]; host3[1] = c->pg_scal_host[1]; host3[2] = c->pg_scal_host[2];
            *failed = *reinterpret_cast<const int*>(c->pg_scal_host + 4);
        }
        return DSSS_OK;
    };
    auto error_of = [&](const pose_t* Xd, double* out) -> int {
        hipLaunchKernelGGL(pg_linearize_kernel<false>, dim3(nblk), dim3(256), 0, st, n, ne, Xd, d_meas, W, d_ea, d_eb, d_eo, d_emeas, d_ew, (double*)nullptr, (double*)nullptr, d_part, mp0, mp1);
        hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal);
        double h3[3]; int f0 = 0;
        HCK(hipMemsetAsync(d_fail, 0, sizeof(int), st));
        int rc2 = reduce_scalars(h3, &f0); if (rc2) return rc2;
        *out = h3[0];
        return DSSS_OK;
    };
    {   // initial values
        double* d_norm = nullptr;
        if (c->pg.add_noise) {
            const long long need_pairs = 3LL * n;
            long long natt = (long long)(need_pairs * 1.32) + 4096;          // acceptance rate pi/4
            for (int attempt = 0;; ++attempt) {
                double* d_pairs; int* d_flags; int* d_bsum; int* d_total;
                const int nb = (int)((natt + 4095) / 4096);
                TRY(dv.alloc(c, &d_pairs, (size_t)natt * 2)); TRY(dv.alloc(c, &d_flags, (size_t)natt)); TRY(dv.alloc(c, &d_bsum, (size_t)nb)); TRY(dv.alloc(c, &d_total, 1));
                if (!d_norm) TRY(dv.alloc(c, &d_norm, (size_t)need_pairs * 2));
                const long long nthr = (natt + RNG_PER_THREAD - 1) / RNG_PER_THREAD;
                hipLaunchKernelGGL(pg_rng_attempts_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, natt, d_pairs, d_flags);
                hipLaunchKernelGGL(pg_flag_blocksum_kernel, dim3(nb), dim3(256), 0, st, d_flags, natt, d_bsum);
                hipLaunchKernelGGL(pg_flag_scan_kernel, dim3(1), dim3(256), 0, st, d_bsum, nb, d_total);
                hipLaunchKernelGGL(pg_flag_compact_kernel, dim3(nb), dim3(256), 0, st, d_flags, d_pairs, natt, d_bsum, need_pairs, d_norm);
                int total_ok = 0;
                HCK(hipMemcpyAsync(&total_ok, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
                HCK(hipStreamSynchronize(st));
                if (total_ok >= need_pairs) break;
                if (attempt > 3) { abandon(); DSSS_FAIL(c, DSSS_E_NUMERIC, "normal generator: not enough accepted attempts"); }
                natt *= 2;
            }
        }
        hipLaunchKernelGGL(pg_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_dr6, d_norm, c->pg.add_noise, d_X, d_meas);
        // online use (dsss_posegraph_update): the pings the previous update covered start from its estimate, the new ones where
        // the reference puts them (dead reckoning o noise, optimizer.cpp:150-160)
        if (c->pg_online && c->pg_warm_n > win_p0) {
            const pose_t* warm = static_cast<const pose_t*>(c->pg_warm) + win_p0;
            HCK(hipMemcpyAsync(d_X, warm, (size_t)std::min(n, c->pg_warm_n - win_p0) * sizeof(pose_t), hipMemcpyDeviceToDevice, st));
            // a window is CONDITIONED on the frozen part of the trajectory through its first pose: the prior (sigma 1e-6) holds it where the
            // previous update left it instead of at its dead-reckoned pose
            if (win_p0 > 0) HCK(hipMemcpyAsync(d_meas, warm, sizeof(pose_t), hipMemcpyDeviceToDevice, st));
        }
    }
    double lambda = c->pg.lambda0, err = 0, err0 = 0, cur = 0;
    int iters = 0, nfact = 0;
    TRY(error_of(d_X, &err));
    err0 = err;
    // the chain part of a trial: per-pose blocks, pass 1 (chunks of poses onto their ends), the level-1 chain, pass 2 (runs of
    // chunk ends onto the true separators)
    auto chain_part = [&]() {
        hipMemsetAsync(d_fail, 0, sizeof(int), st);
        // blocks of the ends of the level-1 chain only: pass 1 forms those of the interior poses from the Jacobians (pg_segment_kernel<true>)
        hipLaunchKernelGGL(pg_assemble_kernel, dim3((ns1 + PG_ASM_POSES - 1) / PG_ASM_POSES), dim3(6 * PG_ASM_POSES), 0, st, n, W, d_r, d_Ji, d_adj_ptr, d_adj_edge, d_ew, d_scal + 3, d_D, d_C, d_g, d_eo, mp0, mp1, d_sep1, ns1);
        hipLaunchKernelGGL(pg_segment_kernel<true>, dim3((nseg1 + 31) / 32), dim3(256), 0, st, nseg1, d_ord1, d_sep1, d_D, d_C, d_g, d_E, d_Dl, d_gi, d_sDL, d_sDR, d_sGL, d_sGR, d_sS, d_fail, mp0, mp1, d_r, d_Ji, W, d_scal + 3);
        hipLaunchKernelGGL(pg_chain1_kernel, dim3((unsigned)(((long long)ns1 * 42 + 255) / 256)), dim3(256), 0, st, ns1, d_sep1, d_D, d_g, d_sDL, d_sDR, d_sGL, d_sGR, d_sS, d_D1, d_C1, d_g1, mp0, mp1);
        if (nseg > 0) hipLaunchKernelGGL(pg_segment_kernel<false>, dim3((nseg + 31) / 32), dim3(256), 0, st, nseg, d_ord2, d_t2, d_D1, d_C1, d_g1, d_E1, d_Dl1, d_gi1, d_s2DL, d_s2DR, d_s2GL, d_s2GR, d_s2S, d_fail, kp0, kp1, (const double*)nullptr, (const double*)nullptr, W, (const double*)nullptr);
    };
    const bool will_iterate = err > 0 && c->pg.max_iters > 0;
    bool pre_lin = false, pre_chain = false;
    if (will_iterate) {                              // first linearisation and the chain part of the first trial, before the analysis is in
        hipLaunchKernelGGL(pg_linearize_kernel<true>, dim3(nblk), dim3(256), 0, st, n, ne, d_X, d_meas, W, d_ea, d_eb, d_eo, d_emeas, d_ew, d_r, d_Ji, d_part, mp0, mp1);
        hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal);
        HCK(hipMemcpyAsync(d_scal + 3, &lambda, sizeof(double), hipMemcpyHostToDevice, st));
        chain_part();
        pre_lin = pre_chain = true;
    }
    // ---- The bottom of the tree (ordering, column structures, bins, update lists, destinations) is final about a millisecond before the
    // fronts and the schedule are: with one partition its tables go up as soon as the analysing thread says so, and the scatter and
    // the bins of the FIRST trial run while the host finishes the analysis.
    size_t nnzL = 0; int nval = 0;
    size_t ncv = 0, nif = 0, comm_total = 8;
    double *d_comm = nullptr, *d_avalif = nullptr, *d_xif = nullptr, *d_commU = nullptr;
    int *d_ifslot = nullptr, *d_ifsep = nullptr;
    std::vector<int> ifslot;
    int bin_lo = 0, bin_hi = 0, nbins = 0;
    bool pre_bins = false;
    // The bins' update lists, update-map offsets and root-boundary indices are built on the device from the column structures
    // (pg_rl_count_kernel ... pg_build_map_kernel): everything they read -- colptr, rowidx, the binned flags, the subtree roots -- is final
    // BEFORE the host packs the bins (pg_sym_opts::on_lists_ready), so they are uploaded and the ten kernels run while it does.
    bool lists_done = false;
    auto upload_lists = [&]() -> int {
        nnzL = S.rowidx.size();
        const int ns = S.ns;                                     // (the analysed graph: this rank's own in the rank-local mode)
        int rc2 = DSSS_OK;
        char* d_binned = nullptr; int* d_rootof = nullptr;
        dv.later(&d_colptr, S.colptr); dv.later(&d_rowidx, S.rowidx); dv.later(&d_binned, S.binned); dv.later(&d_rootof, S.root_of);
        if ((rc2 = dv.flush(c, st))) return rc2;
        int *d_cnt, *d_cur, *d_bs32, *d_tot32; long long *d_bs64, *d_tot64;
        const int nsb = (ns + 1023) / 1024;
        if ((rc2 = dv.alloc(c, &d_cnt, (size_t)ns)) || (rc2 = dv.alloc(c, &d_cur, (size_t)ns)) || (rc2 = dv.alloc(c, &d_bs32, (size_t)nsb)) || (rc2 = dv.alloc(c, &d_tot32, 1)) ||
            (rc2 = dv.alloc(c, &d_bs64, (size_t)nsb)) || (rc2 = dv.alloc(c, &d_tot64, 1)) || (rc2 = dv.alloc(c, &d_rlptr, (size_t)ns + 1)) || (rc2 = dv.alloc(c, &d_mapptr, (size_t)ns + 1)) ||
            (rc2 = dv.alloc(c, &d_rlcol, nnzL)) || (rc2 = dv.alloc(c, &d_rlpos, nnzL)) || (rc2 = dv.alloc(c, &d_rlrow, nnzL)) ||      // (an entry of L is on at most one list)
            (rc2 = dv.alloc(c, &d_anc_first, (size_t)ns)) || (rc2 = dv.alloc(c, &d_anc_rel, nnzL))) return rc2;
        if (nsb > 1024 * 1024) DSSS_FAIL(c, DSSS_E_CAPACITY, "%d separators", ns);
        hipMemsetAsync(d_cnt, 0, sizeof(int) * (size_t)ns, st); hipMemsetAsync(d_cur, 0, sizeof(int) * (size_t)ns, st);
        hipMemsetAsync(d_anc_first, 0, sizeof(int) * (size_t)ns, st); hipMemsetAsync(d_anc_rel, 0xff, sizeof(int) * nnzL, st);
        const dim3 gcol((ns + 255) / 256);
        hipLaunchKernelGGL(pg_rl_count_kernel, gcol, dim3(256), 0, st, ns, d_colptr, d_rowidx, d_binned, d_cnt);
        hipLaunchKernelGGL((pg_scan_block_kernel<int, 0>), dim3(nsb), dim3(1024), 0, st, ns, d_cnt, d_colptr, d_rlptr, d_bs32);
        hipLaunchKernelGGL((pg_scan_tops_kernel<int>), dim3(1), dim3(1024), 0, st, nsb, d_bs32, d_tot32);
        hipLaunchKernelGGL((pg_scan_add_kernel<int>), dim3(nsb), dim3(1024), 0, st, ns, d_rlptr, d_bs32, d_tot32);
        hipLaunchKernelGGL((pg_scan_block_kernel<long long, 1>), dim3(nsb), dim3(1024), 0, st, ns, d_cnt, d_colptr, d_mapptr, d_bs64);
        hipLaunchKernelGGL((pg_scan_tops_kernel<long long>), dim3(1), dim3(1024), 0, st, nsb, d_bs64, d_tot64);
        hipLaunchKernelGGL((pg_scan_add_kernel<long long>), dim3(nsb), dim3(1024), 0, st, ns, d_mapptr, d_bs64, d_tot64);
        hipLaunchKernelGGL(pg_rl_fill_kernel, gcol, dim3(256), 0, st, ns, d_colptr, d_rowidx, d_binned, d_rlptr, d_cur, d_rlcol, d_rlpos);
        hipLaunchKernelGGL(pg_rl_sort_kernel, dim3((ns + 3) / 4), dim3(256), 0, st, ns, d_rlptr, d_rlcol, d_rlpos, d_rlrow, d_fail);
        hipLaunchKernelGGL(pg_anc_rel_kernel, gcol, dim3(256), 0, st, ns, d_colptr, d_rowidx, d_binned, d_rootof, d_anc_first, d_anc_rel);
        // the totals stay on the device: a binned column has at most 42 blocks and an entry of L is on at most one list, so the map
        // has at most 42 nnz(L) entries -- allocated to that bound (54 MB at C3), filled and built up to the device-side totals
        const long long mapsz = 42LL * (long long)nnzL; const int nupd = (int)nnzL;
        if (mapsz > (1LL << 31)) DSSS_FAIL(c, DSSS_E_CAPACITY, "update map bound of %lld entries", mapsz);
        if ((rc2 = dv.alloc(c, &d_map, (size_t)mapsz))) return rc2;
        hipLaunchKernelGGL(pg_fill_map_kernel, dim3(2048), dim3(256), 0, st, d_map, d_tot64);
        if (nupd > 0) hipLaunchKernelGGL(pg_build_map_kernel, dim3((nupd + 255) / 256), dim3(256), 0, st, nupd, d_rlrow, d_rlptr, d_rlcol, d_rlpos, d_colptr, d_rowidx, d_mapptr, d_map, d_tot32);
        lists_done = true;
        return DSSS_OK;
    };
    auto upload_bottom = [&]() -> int {
        nnzL = S.rowidx.size(); nval = local ? (int)dest_g.size() : (int)S.dest_bin.size();
        ncv = S.comm_vals.size(); nif = S.iface_seps.size();
        comm_total = ncv * 36 + nif * 6 + (size_t)S.comm_doubles + 8;
        int rc2 = DSSS_OK;
        if (!lists_done && (rc2 = upload_lists())) return rc2;
        dv.later(&d_perm, local ? perm_g : S.perm);
        if ((rc2 = dv.alloc(c, &d_L, nnzL * 36))) return rc2;
        if ((rc2 = dv.alloc(c, &d_ubin, (size_t)S.ubin_doubles))) return rc2;
        // value array of the fronts; its tail IS the buffer the all-reduce sums: [interface values | interface right-hand sides |
        // update matrices that cross into the interface | 8 scalars]
        if ((rc2 = dv.alloc(c, &d_aval, (size_t)nval * 36 + comm_total))) return rc2;
        d_comm = d_aval + (size_t)nval * 36; d_avalif = d_comm; d_xif = d_comm + ncv * 36; d_commU = d_xif + nif * 6;
        if (local) ifslot.swap(ifslot_g);
        else { ifslot.assign(ns, -1); for (size_t q = 0; q < nif; ++q) ifslot[S.iface_seps[q]] = (int)q; }
        dv.later(&d_ifslot, ifslot); dv.later(&d_ifsep, local ? ifsep_g : S.iface_seps);
        dv.later(&d_binptr, S.binptr); dv.later(&d_bincols, S.bincols); dv.later(&d_binperm, S.bin_perm);
        dv.later(&d_dest, local ? dest_g : S.dest_bin);
        dv.later(&d_binroot_ptr, S.binroot_ptr); dv.later(&d_binroot_idx, S.binroot_idx); dv.later(&d_broot_b, S.broot_b); dv.later(&d_broot_uoff, S.broot_uoff);
        dv.later(&d_broot_of_col, S.broot_of_col);
        if ((rc2 = dv.flush(c, st))) return rc2;
        // this rank's bins are one contiguous range (bins never straddle partitions, partitions are ascending in the order)
        { const int nb_all = (int)S.binptr.size() - 1, sel_lo = local ? 0 : part_lo, sel_hi = local ? 1 : part_hi;
          bin_lo = 0; while (bin_lo < nb_all && S.bin_part[bin_lo] < sel_lo) ++bin_lo; bin_hi = bin_lo; while (bin_hi < nb_all && S.bin_part[bin_hi] < sel_hi) ++bin_hi; }
        nbins = bin_hi - bin_lo;
        return DSSS_OK;
    };
    // the bottom part of a trial: reduced system into the factor / the value array, then the bins
    auto bottom_trial = [&](double bins_flops) {
        hipMemsetAsync(d_L, 0, nnzL * 36 * sizeof(double), st);
        if (nparts > 1) hipMemsetAsync(d_comm, 0, comm_total * sizeof(double), st);
        hipLaunchKernelGGL(pg_scatter_base_kernel, dim3((unsigned)(((long long)ns * 78 + 255) / 256)), dim3(256), 0, st, ns, d_t2, d_perm, d_D1, d_g1, d_s2DL, d_s2DR, d_s2GL, d_s2GR, d_s2S, d_dest, d_L, d_aval, d_x,
                           d_ifslot, d_avalif, d_xif, kp0, kp1);
        if (ne > 0) hipLaunchKernelGGL(pg_scatter_lc_kernel, dim3((unsigned)(((long long)ne * 36 + 255) / 256)), dim3(256), 0, st, n, ne, ns, d_Ji, d_ew, d_dest, d_L, d_aval, d_avalif, d_eo, mp0, mp1, d_lc_link);
        if (nbins > 0) { dsss_scope s1(c, DSSS_K_PG_SUBTREE, bins_flops);      // flops of the binned columns
                         hipLaunchKernelGGL(pg_factor_subtree_kernel, dim3(nbins), dim3(256), 0, st, d_binperm + bin_lo, d_binptr, d_bincols, d_colptr, d_rlptr, d_rlcol, d_rlpos, d_mapptr, d_map, d_L, d_x, d_fail,
                                            d_binroot_ptr, d_binroot_idx, d_broot_b, d_broot_uoff, d_broot_of_col, d_anc_first, d_anc_rel, d_ubin, d_rdiag); }
    };
    const bool early_bottom = nparts == 1 && will_iterate && pre_chain;
    if (early_bottom) {
        lists_fut.wait(); TRY(upload_lists());
        bottom_fut.wait();
        TRY(upload_bottom());
        bottom_trial(0.0);                                   // (its flop count is known when the analysis has finished: added below)
        pre_bins = true;
    }
    sym_thread.join();
    if (S.ownership_violations) { abandon(); DSSS_FAIL(c, DSSS_E_STATE, "pose-graph analysis: %d separators with a higher-rank neighbour are not interface", S.ownership_violations); }
    const double t_sym = ms_since(T1);
    const auto T2 = std::chrono::steady_clock::now();
    const int nfr = (int)S.f_c0.size(), npan = S.npanels;
    if (!early_bottom) TRY(upload_bottom());
    else if (c->prof.on) c->prof.work[DSSS_K_PG_SUBTREE] += std::max(0.0, S.flops_factor - S.flops_fronts);
    if (verbose)
        fprintf(stderr, "[dsss pg] rank %d/%d parts %d (own %d..%d, poses %d..%d)  %s analysis of %d separators  poses %d  LC edges %d  separators %d (interface %zu)  nnz(L) blocks %zu  bins %d (%d cols)  fronts %d (largest %d block rows, arena %.0f MB)  panels %d in %d levels  all-reduce %.1f MB\n",
                rank, world, nparts, part_lo, part_hi, mp0, mp1, local ? "rank-local" : "replicated", S.ns, n, ne, ns, S.iface_seps.size(), nnzL, (int)S.binptr.size() - 1, (int)S.bincols.size(), nfr, S.max_front_n, S.front_doubles * 8e-6, npan, S.nlev,
                (S.comm_doubles + 36.0 * S.comm_vals.size() + 6.0 * S.iface_seps.size()) * 8e-6);

    // rank-local mode: the interface front is the LAST front of the arena and its right-hand side sits directly behind it, so that one
    // all-reduce sums both in place
    const int f_if = local && !S.iface_seps.empty() ? nfr - 1 : -1;
    const size_t if_ld = f_if >= 0 ? (size_t)S.f_ld[f_if] : 0, if_count = if_ld * if_ld + if_ld;
    if (f_if >= 0 && (S.f_part[f_if] != -1 || S.f_off[f_if] + (long long)(if_ld * if_ld) != S.front_doubles)) { abandon(); DSSS_FAIL(c, DSSS_E_STATE, "pose-graph analysis: the interface front is not the last one"); }
    TRY(dv.alloc(c, &d_F, (size_t)S.front_doubles + if_ld)); TRY(dv.alloc(c, &d_R, (size_t)S.frhs_doubles));
    const int rsu_max = PG_RSU_MAX_TILES;               // levels with more tiles run the row solve and the trailing update as two launches
    const bool use_rsu = true;
    const int rsu32_max = PG_RSU32_MAX_TILES;           // levels with at most this many 64 x 64 tiles run them as 32 x 32 quarters
    double* d_FL = nullptr;                              // second front arena: L21 of the levels that run the fused kernel
    if (use_rsu) TRY(dv.alloc(c, &d_FL, (size_t)S.front_doubles));
    int *d_pk_child, *d_pk_row; pg_pack* d_PK;
    dv.later(&d_rel, S.rel); dv.later(&d_fa_src, S.fa_src); dv.later(&d_fa_col, S.fa_col); dv.later(&d_fa_tr, S.fa_tr);
    dv.later(&d_frows, S.f_rows); dv.later(&d_xr_ptr, S.xr_ptr); dv.later(&d_xr_child, S.xr_child);
    dv.later(&d_xr_row, S.xr_row); dv.later(&d_fa_rowptr, S.fa_rowptr);
    struct dsched { int *lv_front, *lv_step, *asm_front, *asm_row, *tile_item, *tile_ij; } DO = {}, DI = {};
    for (int w2 = 0; w2 < 2; ++w2) {
        const pg_sched& H = w2 ? SI : SO; dsched& Dv = w2 ? DI : DO;
        dv.later(&Dv.lv_front, H.lv_front); dv.later(&Dv.lv_step, H.lv_step); dv.later(&Dv.asm_front, H.asmrow_front); dv.later(&Dv.asm_row, H.asmrow_row);
        dv.later(&Dv.tile_item, H.tile_item); dv.later(&Dv.tile_ij, H.tile_ij);
    }
    int n_pack = 0;
    {   // front and child descriptors (the children point straight at the update matrices: F22 of a front, U of a bin root)
        std::vector<pg_front> FD(nfr); std::vector<pg_child> CH(S.ch_kind.size());
        for (int f = 0; f < nfr; ++f) {
            pg_front& d = FD[f];
            d.off = S.f_off[f]; d.roff = S.f_roff[f]; d.ld = S.f_ld[f]; d.n6 = 6 * S.f_n[f]; d.s6 = 6 * S.f_s[f]; d.c0 = S.f_c0[f];
            d.rowptr = S.f_rowptr[f]; d.pan0 = S.f_pan0[f]; d.ch0 = S.ch_ptr[f]; d.ch1 = S.ch_ptr[f + 1]; d.fa0 = S.fa_ptr[f]; d.fa1 = S.fa_ptr[f + 1];
            if (f == f_if) d.roff = (d_F + S.front_doubles) - d_R;      // (both come out of the solver's one arena)
        }
        for (size_t q = 0; q < CH.size(); ++q) {
            pg_child& d = CH[q]; d.relptr = S.ch_relptr[q];
            if (S.ch_kind[q]) { const int ri = S.ch_id[q]; d.cb = S.broot_b[ri]; d.cld = 6 * d.cb; d.U = d_ubin + S.broot_uoff[ri]; d.g = d.U + (size_t)d.cld * d.cld; }
            else { const int gf = S.ch_id[q]; d.cb = S.f_n[gf] - S.f_s[gf]; d.cld = S.f_ld[gf]; d.U = d_F + S.f_off[gf] + (size_t)(6 * S.f_s[gf]) * d.cld + 6 * S.f_s[gf]; d.g = d_R + S.f_roff[gf] + 6 * S.f_s[gf]; }
        }
        // children that cross from an interior into the interface are read from the summed buffer; the owner packs them there
        std::vector<int> comm_of_front(nfr, -1), comm_of_broot(S.broot.size(), -1);
        for (size_t q = 0; q < S.comm_kind.size(); ++q) (S.comm_kind[q] ? comm_of_broot[S.comm_id[q]] : comm_of_front[S.comm_id[q]]) = (int)q;
        std::vector<pg_pack> PK(S.comm_kind.size()); std::vector<int> pk_child, pk_row;
        for (int f = 0; f < nfr; ++f) {
            if (S.f_part[f] >= 0) continue;
            for (int q = S.ch_ptr[f]; q < S.ch_ptr[f + 1]; ++q) {
                const int cq = S.ch_kind[q] ? comm_of_broot[S.ch_id[q]] : comm_of_front[S.ch_id[q]];
                if (cq < 0) continue;
                pg_child& d = CH[q];
                pg_pack& k = PK[cq]; k.U = d.U; k.g = d.g; k.cld = d.cld; k.cb = d.cb; k.dst = d_commU + S.comm_off[cq];
                d.U = k.dst; d.cld = 6 * d.cb; d.g = d.U + (size_t)d.cld * d.cld;
                if (S.comm_part[cq] >= part_lo && S.comm_part[cq] < part_hi) for (int r2 = 0; r2 < d.cb; ++r2) { pk_child.push_back(cq); pk_row.push_back(r2); }
            }
        }
        dv.later(&d_FD, FD); dv.later(&d_CH, CH); dv.later(&d_PK, PK); dv.later(&d_pk_child, pk_child); dv.later(&d_pk_row, pk_row);
        TRY(dv.flush(c, st));                                   // FD, CH, PK and the lists above are still alive here
        n_pack = (int)pk_child.size();
    }
    const int max_n6 = std::max(SO.max_n6, SI.max_n6);
    c->pg_last_levels.clear();
    for (const pg_sched* H : { &SO, &SI })
        for (int l = 0; l < H->nlev; ++l) {
            const int nit = H->lv_ptr[l + 1] - H->lv_ptr[l];
            if (nit > 0) { c->pg_last_levels.push_back(nit); c->pg_last_levels.push_back(H->max_w6[l]); c->pg_last_levels.push_back(H->max_rows[l]); c->pg_last_levels.push_back(H == &SI ? 1 : 0); }
        }
    double* d_Tinv; TRY(dv.alloc(c, &d_Tinv, (size_t)std::max(npan, 1) * PG_NB4 * 16));
    double* d_bwp = nullptr;                                // partial sums of the split back-substitution products (tall fronts only)
    {
        size_t need = 0;
        for (const pg_sched* H : { &SO, &SI })
            for (int l = 0; l < H->nlev; ++l)
                if (64 * H->trsm_chunks[l] > PG_BWD_SPLIT) need = std::max(need, (size_t)(H->lv_ptr[l + 1] - H->lv_ptr[l]) * ((64 * H->trsm_chunks[l] + PG_BWD_RC - 1) / PG_BWD_RC) * 96);
        if (need > 0) TRY(dv.alloc(c, &d_bwp, need));
    }
    const int bwd_lds = (int)(((PG_PW * 6) * PG_BWD2_LD + PG_NB4 * 16 + 10 * (PG_PW * 6) + std::min(max_n6, PG_BWD2_SX) + 16) * sizeof(double));
    if (bwd_lds > 160 * 1024) { abandon(); DSSS_FAIL(c, DSSS_E_CAPACITY, "front of %d scalar rows: back-substitution needs %d B of LDS", max_n6, bwd_lds); }
    {   // the back-substitution keeps L11, the slot sums and x2 in dynamic LDS
        hipFuncSetAttribute((const void*)pg_front_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
    }
    const double t_up = ms_since(T2);
    const auto T3 = std::chrono::steady_clock::now();
    dsss_scope sc(c, DSSS_K_PG);
    if (will_iterate) do {                        // NonlinearOptimizer::defaultOptimize returns before iterating when maxIterations is reached
        cur = err;
        double oldLin = err;                                                   // linear error at delta = 0 == the error at X (same sum, already global)
        if (!pre_lin) {
            hipLaunchKernelGGL(pg_linearize_kernel<true>, dim3(nblk), dim3(256), 0, st, n, ne, d_X, d_meas, W, d_ea, d_eb, d_eo, d_emeas, d_ew, d_r, d_Ji, d_part, mp0, mp1);
            hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal);
        }
        pre_lin = false;
        for (;;) {
            // ---- solve (H + lambda I) delta = -g ; lambda lives in device memory
            if (!pre_chain) HCK(hipMemcpyAsync(d_scal + 3, &lambda, sizeof(double), hipMemcpyHostToDevice, st));
            {
                if (!pre_chain) chain_part();
                pre_chain = false;
                if (!pre_bins) bottom_trial(std::max(0.0, S.flops_factor - S.flops_fronts));
                pre_bins = false;
                // fronts, level by level: assemble the fronts that start here, then one panel step of every active front
                // part: 0 everything, 1 the assemblies only, 2 everything but the assemblies (rank-local mode: the all-reduce sits between the two)
                auto run_levels = [&](const pg_sched& H, const dsched& Dv, int part = 0) {
                    for (int l = 0; l < H.nlev; ++l) {
                        const int nas = part == 2 ? 0 : H.asmrow_ptr[l + 1] - H.asmrow_ptr[l], nit = part == 1 ? 0 : H.lv_ptr[l + 1] - H.lv_ptr[l], ntl = H.tile_ptr[l + 1] - H.tile_ptr[l];
                        const int* itf = Dv.lv_front + H.lv_ptr[l]; const int* its = Dv.lv_step + H.lv_ptr[l];
                        if (nas > 0) { dsss_scope s2(c, DSSS_K_PG_ASM);
                            hipLaunchKernelGGL(pg_front_asm_kernel, dim3(nas), dim3(256), 0, st, Dv.asm_front + H.asmrow_ptr[l], Dv.asm_row + H.asmrow_ptr[l], d_FD, d_CH, d_rel, d_xr_ptr, d_xr_child, d_xr_row,
                                               d_fa_rowptr, d_fa_src, d_fa_col, d_fa_tr, d_aval, d_x, d_F, d_R); }
                        if (nit == 0) continue;
                        { dsss_scope s3(c, DSSS_K_PG_DIAG, H.fl_diag[l]);
                          hipLaunchKernelGGL(pg_front_diag4_kernel, dim3(nit), dim3(256), 0, st, itf, its, d_FD, d_F, d_R, d_fail, d_Tinv); }
                        if (use_rsu && ntl > 0 && ntl <= rsu_max) {
                            dsss_scope s45(c, DSSS_K_PG_RSU, H.fl_trsm[l] + H.fl_syrk[l]);
                            if (ntl <= rsu32_max) hipLaunchKernelGGL(pg_front_rsu_kernel<32>, dim3(4 * ntl), dim3(256), 0, st, itf, its, d_FD, Dv.tile_item + H.tile_ptr[l], Dv.tile_ij + H.tile_ptr[l], d_F, d_FL, d_R, d_Tinv);
                            else hipLaunchKernelGGL(pg_front_rsu_kernel<64>, dim3(ntl), dim3(512), 0, st, itf, its, d_FD, Dv.tile_item + H.tile_ptr[l], Dv.tile_ij + H.tile_ptr[l], d_F, d_FL, d_R, d_Tinv);
                        } else if (H.trsm_chunks[l] > 0) {
                            { dsss_scope s4(c, DSSS_K_PG_TRSM, H.fl_trsm[l]);
                              hipLaunchKernelGGL(pg_front_trsm2_kernel, dim3(nit, H.trsm_chunks[l]), dim3(256), 0, st, itf, its, d_FD, d_F, d_R, d_Tinv); }
                            { dsss_scope s5(c, DSSS_K_PG_ACC, H.fl_syrk[l]);
                              if (ntl > 0) hipLaunchKernelGGL(pg_front_syrk_kernel, dim3(ntl), dim3(256), 0, st, itf, its, d_FD, Dv.tile_item + H.tile_ptr[l], Dv.tile_ij + H.tile_ptr[l], d_F); }
                        }
                    }
                };
                auto run_levels_bwd = [&](const pg_sched& H, const dsched& Dv) {
                    for (int l = H.nlev - 1; l >= 0; --l) {
                        const int nit = H.lv_ptr[l + 1] - H.lv_ptr[l];
                        if (nit == 0) continue;
                        dsss_scope s6(c, DSSS_K_PG_BWD, H.fl_bwd[l], 2);
                        const int ntl = H.tile_ptr[l + 1] - H.tile_ptr[l];
                        const double* Fl = (use_rsu && ntl > 0 && ntl <= rsu_max) ? d_FL : d_F;
                        const int nch = 64 * H.trsm_chunks[l] > PG_BWD_SPLIT ? (64 * H.trsm_chunks[l] + PG_BWD_RC - 1) / PG_BWD_RC : 0;      // tall fronts on this level: their L21^T x2 by many workgroups
                        if (nch > 0) hipLaunchKernelGGL(pg_front_bwd_part_kernel, dim3(nch, nit), dim3(1024), 0, st, Dv.lv_front + H.lv_ptr[l], Dv.lv_step + H.lv_ptr[l], d_FD, d_frows, Fl, d_x, d_bwp, nch);
                        hipLaunchKernelGGL(pg_front_bwd2_kernel, dim3(nit), dim3(1024), bwd_lds, st, Dv.lv_front + H.lv_ptr[l], Dv.lv_step + H.lv_ptr[l], d_FD, d_frows, d_F, Fl, d_R, d_x, d_Tinv,
                                           nch > 0 ? (const double*)d_bwp : (const double*)nullptr, nch);
                    }
                };
                run_levels(SO, DO);
                if (local) {
                    // rank-local analysis: this rank's share of the reduced Hessian on the interface IS its interface front -- its own original
                    // values and right-hand sides plus the update matrices of its interior, assembled as any front is -- and the all-reduce
                    // sums the fronts (and the right-hand sides behind them) in place; then every rank factorises the sum
                    if (f_if >= 0) {
                        hipLaunchKernelGGL(pg_comm_xif_kernel, dim3(((int)nif + 255) / 256), dim3(256), 0, st, (int)nif, d_ifsep, d_perm, d_xif, d_x);
                        hipMemsetAsync(d_F + S.f_off[f_if], 0, if_count * sizeof(double), st);      // (the assembly writes the lower block triangle only)
                        run_levels(SI, DI, 1);
                        { dsss_scope s8(c, DSSS_K_PG_COMM, (double)if_count * 8);
                          int rc2 = dsss_comm_allreduce(c, d_F + S.f_off[f_if], if_count, st); if (rc2) { abandon(); return rc2; } }
                        run_levels(SI, DI, 2);
                        run_levels_bwd(SI, DI);
                    }
                } else if (nparts > 1) {
                    // the reduced Hessian on the interface: this rank's update matrices next to its share of the interface values and
                    // right-hand sides, summed over the ranks by ONE all-reduce; then the small replicated interface factorisation
                    if (n_pack > 0) hipLaunchKernelGGL(pg_comm_pack_kernel, dim3(n_pack), dim3(256), 0, st, d_pk_child, d_pk_row, d_PK);
                    { dsss_scope s8(c, DSSS_K_PG_COMM, (double)comm_total * 8);
                      int rc2 = dsss_comm_allreduce(c, d_comm, comm_total - 8, st); if (rc2) { abandon(); return rc2; } }
                    if (nif > 0) hipLaunchKernelGGL(pg_comm_xif_kernel, dim3(((int)nif + 255) / 256), dim3(256), 0, st, (int)nif, d_ifsep, d_perm, d_xif, d_x);
                    run_levels(SI, DI);
                    run_levels_bwd(SI, DI);
                }
                run_levels_bwd(SO, DO);
                if (nbins > 0) { dsss_scope s7(c, DSSS_K_PG_SUBTREE);
                    hipLaunchKernelGGL(pg_bwd_subtree_kernel, dim3(nbins), dim3(64), 0, st, d_binperm + bin_lo, d_binptr, d_bincols, d_colptr, d_rowidx, d_L, d_x, d_rdiag); }
                hipLaunchKernelGGL(pg_sep_delta_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, ns, d_t2, d_perm, d_x, d_delta1);
                if (nseg > 0) hipLaunchKernelGGL(pg_backsub_kernel<false>, dim3((nseg + 31) / 32), dim3(256), 0, st, nseg, d_ord2, d_t2, d_C1, d_E1, d_Dl1, d_gi1, d_delta1, kp0, kp1, (const double*)nullptr, W);
                hipLaunchKernelGGL(pg_sep_delta_kernel, dim3((ns1 + 255) / 256), dim3(256), 0, st, ns1, d_sep1, (const int*)nullptr, d_delta1, d_delta);
                hipLaunchKernelGGL(pg_backsub_kernel<true>, dim3((nseg1 + 31) / 32), dim3(256), 0, st, nseg1, d_ord1, d_sep1, d_C, d_E, d_Dl, d_gi, d_delta, mp0, mp1, d_Ji, W);
                hipLaunchKernelGGL(pg_linerr_kernel, dim3(nblk), dim3(256), 0, st, n, ne, W, d_ea, d_eb, d_eo, d_ew, d_r, d_Ji, d_delta, d_part, mp0, mp1);
                hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal + 1);
            }
            ++nfact;
            // X and Xn swap between trials, so these two stay outside the captured graph
            hipLaunchKernelGGL(pg_retract_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, d_X, d_delta, d_Xn);
            hipLaunchKernelGGL(pg_linearize_kernel<true>, dim3(nblk), dim3(256), 0, st, n, ne, d_Xn, d_meas, W, d_ea, d_eb, d_eo, d_emeas, d_ew, d_r2, d_Ji2, d_part, mp0, mp1);
            hipLaunchKernelGGL(pg_final_sum_kernel, dim3(1), dim3(256), 0, st, d_part, nblk, 0.5, d_scal + 2);
            HCK(hipGetLastError());
            double h[3]; int failed = 0;
            { int rc2 = reduce_scalars(h, &failed); if (rc2) return rc2; }
            const bool ok = !failed && std::isfinite(h[1]);
            bool success = false, stop = false;
            double newErr = 0;
            if (ok) {
                const double linChange = oldLin - h[1];
                if (linChange >= 0) {
                    newErr = h[2];
                    const double costChange = err - newErr;
                    if (linChange > 2.220446049250313e-16 * oldLin) success = (costChange / linChange) > c->pg.min_fidelity;
                    if (std::fabs(costChange) < c->pg.rel_tol * err) stop = true;
                }
            }
            if (success) { std::swap(d_X, d_Xn); std::swap(d_r, d_r2); std::swap(d_Ji, d_Ji2); pre_lin = true; err = newErr; lambda /= c->pg.lambda_factor; ++iters; break; }
            else if (!stop) { lambda *= c->pg.lambda_factor; if (lambda >= c->pg.lambda_max) break; }
            else break;
        }
    } while (iters < c->pg.max_iters && !((err <= 0) || ((cur - err) / cur <= c->pg.rel_tol) || ((cur - err) <= c->pg.abs_tol)) && std::isfinite(cur));
    const auto T4 = std::chrono::steady_clock::now();
    const double t_lm = ms_since(T3);
    if (world > 1) {
        // every rank holds the final values of its own poses [mp0, mp1): ONE all-gather of equal slices (the longest rank's range; own slice in
        // place) and a copy per rank put the whole trajectory everywhere.  (Rounds 2 - 5 zeroed the others' poses and all-reduced all n x 12
        // doubles: a ring moves twice the bytes for a sum of zeros -- 38.4 MB per step at C3, the largest collective of a step by far.)
        size_t maxlen = 0;
        std::vector<int> r_lo(world), r_len(world);
        for (int r = 0; r < world; ++r) {
            r_lo[r] = pbound[(int)((long long)nparts * r / world)]; r_len[r] = pbound[(int)((long long)nparts * (r + 1) / world)] - r_lo[r];
            maxlen = std::max(maxlen, (size_t)r_len[r]);
        }
        const size_t slice = maxlen * sizeof(pose_t);
        char* d_gath; TRY(dv.alloc(c, &d_gath, slice * (size_t)world));
        HCK(hipMemcpyAsync(d_gath + slice * (size_t)rank, d_X + mp0, (size_t)(mp1 - mp0) * sizeof(pose_t), hipMemcpyDeviceToDevice, st));
        { int rc2 = dsss_comm_allgather(c, d_gath, slice, st); if (rc2) { abandon(); return rc2; } }
        for (int r = 0; r < world; ++r)
            if (r != rank && r_len[r] > 0) HCK(hipMemcpyAsync(d_X + r_lo[r], d_gath + slice * (size_t)r, (size_t)r_len[r] * sizeof(pose_t), hipMemcpyDeviceToDevice, st));
    }
    static_assert(sizeof(pose_t) == 12 * sizeof(double), "pose_t layout");
    if (c->pg_online) {
        const size_t all = (size_t)win_p0 + (size_t)n;
        if (c->pg_warm_cap < all) {
            HCK(hipStreamSynchronize(st));
            const size_t cap = all + all / 2 + 1024;                        // the graph grows by a frame per update
            void* nw = nullptr;
            HCK(hipMalloc(&nw, cap * sizeof(pose_t)));
            if (win_p0 > 0 && c->pg_warm) {                                 // (a window keeps the frozen part in front of it)
                const hipError_t e = hipMemcpy(nw, c->pg_warm, (size_t)win_p0 * sizeof(pose_t), hipMemcpyDeviceToDevice);
                if (e != hipSuccess) { hipFree(nw); HCK(e); }
            }
            if (c->pg_warm) hipFree(c->pg_warm);
            c->pg_warm = nw; c->pg_warm_cap = cap;
        }
        HCK(hipMemcpyAsync(static_cast<pose_t*>(c->pg_warm) + win_p0, d_X, (size_t)n * sizeof(pose_t), hipMemcpyDeviceToDevice, st));
        c->pg_warm_n = (int)all;
    }
    const pose_t* d_out = win_p0 > 0 ? static_cast<const pose_t*>(c->pg_warm) : d_X;      // a window reports the whole trajectory
    const int n_out = win_p0 > 0 ? win_p0 + n : n;
    if (poses12)        // pose_t is 12 contiguous doubles (R row-major, t): straight into the caller's buffer
        HCK(hipMemcpyAsync(poses12, d_out, (size_t)n_out * sizeof(pose_t), hipMemcpyDeviceToHost, st));
    if (rpy6) {
        double* d_rpy;
        TRY(dv.alloc(c, &d_rpy, (size_t)n_out * 6));
        hipLaunchKernelGGL(pg_rpy_kernel, dim3((n_out + 255) / 256), dim3(256), 0, st, n_out, d_out, d_rpy);
        HCK(hipMemcpyAsync(rpy6, d_rpy, (size_t)n_out * 6 * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    HCK(hipStreamSynchronize(st));
    if (stats4) { stats4[0] = iters; stats4[1] = err0; stats4[2] = err; stats4[3] = lambda; }
    c->pg_last_trials = nfact;
    dv.release();
    if (verbose) fprintf(stderr, "[dsss pg] LM iterations %d  factorisations %d  err %.6g -> %.6g | host prep %.1f ms, symbolic %.1f ms, alloc+upload %.1f ms, LM loop %.1f ms, download %.1f ms\n",
                         iters, nfact, err0, err, t_prep, t_sym, t_up, t_lm, ms_since(T4));
#undef TRY
#undef HCK
    return DSSS_OK;
}

// ------------------------------------------------------------------ LC selection (optimizer.cpp:203-258)
// For target frame t, ping j: the LAST pair (s, t) in pair order holding a kp whose target ping is j wins, and
// within it the FIRST such kp.  One 64-bit atomicMax per kp on key = (pair rank << 32) | (~index in pair).
__global__ __launch_bounds__(256) void lc_select_kernel(const double* __restrict__ kp7, int n, const int* __restrict__ kp7_pair,
                                                        const int* __restrict__ kp7_off, const int* __restrict__ act_t,
                                                        const int* __restrict__ frame_off, unsigned long long* __restrict__ slot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = kp7_pair[i];
    const int t = act_t[p];
    const int ping = (int)kp7[(size_t)i * 7 + 3];
    const unsigned long long key = ((unsigned long long)(unsigned)(p + 1) << 32) | (unsigned long long)(0xffffffffu - (unsigned)(i - kp7_off[p]));
    atomicMax(&slot[frame_off[t] + ping], key);
}

// flag of global pose g: it won a loop closure and that measurement scored > 0 (optimizer.cpp:234)
__global__ __launch_bounds__(256) void lc_edge_flag_kernel(const unsigned long long* __restrict__ slot, int total, const int* __restrict__ kp7_off,
                                                           const dsss_lc* __restrict__ lcs, int* __restrict__ flags)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const unsigned long long key = slot[g];
    int f = 0;
    if (key) {
        const int p = (int)(key >> 32) - 1, k = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
        const dsss_lc& m = lcs[kp7_off[p] + k];
        // score > 0 (optimizer.cpp:234); a non-finite score (fin == 0) or a variance that is not finite and positive
        // (the undamped 15x15 marginal failed: GTSAM would throw) drops the measurement instead of poisoning the batch LM
        f = m.score > 0 && isfinite(m.score);
        for (int q = 0; q < 6; ++q) f = f && m.var[q] > 0 && isfinite(m.var[q]);
    }
    flags[g] = f;
}
// edges in ascending target pose id (the reference's loop order), ordered compaction over blocks of 4096 poses
__global__ __launch_bounds__(256) void lc_edge_compact_kernel(const int* __restrict__ flags, const int* __restrict__ bsum, const unsigned long long* __restrict__ slot,
                                                              int total, const int* __restrict__ kp7_off, const double* __restrict__ kp7,
                                                              const dsss_lc* __restrict__ lcs, const int* __restrict__ act_s, const int* __restrict__ frame_off,
                                                              int cap, dsss_lc_edge* __restrict__ edges, int2* __restrict__ ab)
{
    __shared__ int s_w[4];
    __shared__ int s_run;
    const int i0 = blockIdx.x * 4096;
    if (threadIdx.x == 0) s_run = bsum[blockIdx.x];
    __syncthreads();
    for (int c0 = 0; c0 < 4096; c0 += 256) {
        const int g = i0 + c0 + threadIdx.x;
        const int f = g < total ? flags[g] : 0;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        int inc = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        __syncthreads();
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        int base = s_run;
        for (int k = 0; k < w; ++k) base += s_w[k];
        const int pos = base + inc - f;
        if (f && pos < cap) {
            const unsigned long long key = slot[g];
            const int p = (int)(key >> 32) - 1, k = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
            const int i = kp7_off[p] + k;
            dsss_lc_edge ed;
            ed.a = frame_off[act_s[p]] + (int)kp7[(size_t)i * 7 + 0];
            ed.b = g;
            for (int q = 0; q < 12; ++q) ed.rel[q] = lcs[i].rel[q];
            for (int q = 0; q < 6; ++q) ed.var[q] = lcs[i].var[q];
            edges[pos] = ed;
            if (ab) ab[pos] = make_int2(ed.a, ed.b);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

// ends != NULL (dsss_posegraph_solve, one rank): the END POINTS of the edges also come back as packed (a, b) pairs in the context's
// page-locked buffer (*ends): the first passes of the solve (separators, partition boundaries, reduced edges) walk 8 bytes per edge
// instead of 152.  The records themselves are on the host as well when this returns: dv.release() synchronises the stream, and it has
// to -- the arena the device copies sit in is handed back and the solve's uploads into it are blocking copies on the null stream,
// which do not order against this non-blocking stream.  (Round 5 recorded an event behind the record copy and claimed an overlap
// with the analysis; the release-time synchronisation made that event always already satisfied -- advisor, round 5 -- so it is gone.)
#define PG_AB_PREFIX 32768
static int pg_select_impl(dsss_ctx* c, int nframes, dsss_lc_edge* edges, int cap, int* n_edges, const int** ends)
{
    if (!c || nframes <= 0 || nframes > c->max_frames) return DSSS_E_ARG;
    if (!c->has_lc) DSSS_FAIL(c, DSSS_E_STATE, "dsss_lc_solve_all has not run");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int> off(nframes + 1, 0);
    for (int f = 0; f < nframes; ++f) {
        if (!c->frames[f].has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no geometry", f);
        off[f + 1] = off[f] + c->frames[f].N;
    }
    const int total = off[nframes], n = c->total_kp7;
    int ne = 0;
    if (n > 0) {
        // the reference's "last pair wins" ranks pairs in (i<j) loop order == the caller's pair order, provided the
        // active pairs were listed in that order (they are: dsss_match_pairs keeps the caller's order)
        for (int p = 0; p < c->npairs; ++p)
            if (c->pair_s[p] >= nframes || c->pair_t[p] >= nframes) DSSS_FAIL(c, DSSS_E_ARG, "pair %d references a frame >= nframes", p);
        // everything on the device: winner per target pose (atomicMax), score filter, ordered compaction; only the
        // edge records come back.  Scratch comes from the solver arena (reset by the solve that follows).
        pg_dev dv;
        unsigned long long* d_slot; int *d_off, *d_flags, *d_bsum, *d_total; dsss_lc_edge* d_edges; int2* d_ab = nullptr;
        const int nb = (total + 4095) / 4096;
        int rc = dv.alloc(c, &d_slot, (size_t)total); if (rc) return rc;
        rc = dv.alloc(c, &d_off, (size_t)nframes + 1); if (rc) return rc;
        rc = dv.alloc(c, &d_flags, (size_t)total); if (rc) return rc;
        rc = dv.alloc(c, &d_bsum, (size_t)nb); if (rc) return rc;
        rc = dv.alloc(c, &d_total, 1); if (rc) return rc;
        rc = dv.alloc(c, &d_edges, (size_t)cap); if (rc) return rc;
        if (ends) {
            rc = dv.alloc(c, &d_ab, (size_t)cap); if (rc) return rc;
            if (c->pg_ab_cap < (size_t)cap) {
                if (c->pg_ab_host) hipHostFree(c->pg_ab_host);
                c->pg_ab_host = nullptr; c->pg_ab_cap = 0;
                HIPCHK(c, hipHostMalloc((void**)&c->pg_ab_host, (size_t)cap * sizeof(int2), hipHostMallocDefault));
                c->pg_ab_cap = (size_t)cap;
            }
        }
        hipStream_t st = c->stream;
        HIPCHK(c, hipMemsetAsync(d_slot, 0, (size_t)total * sizeof(unsigned long long), st));
        HIPCHK(c, hipMemcpyAsync(d_off, off.data(), (nframes + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lc_select_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->kp7, n, c->kp7_pair, c->kp7_off, c->act_t, d_off, d_slot);
        hipLaunchKernelGGL(lc_edge_flag_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d_slot, total, c->kp7_off, c->lcs, d_flags);
        hipLaunchKernelGGL(pg_flag_blocksum_kernel, dim3(nb), dim3(256), 0, st, d_flags, (long long)total, d_bsum);
        hipLaunchKernelGGL(pg_flag_scan_kernel, dim3(1), dim3(256), 0, st, d_bsum, nb, d_total);
        hipLaunchKernelGGL(lc_edge_compact_kernel, dim3(nb), dim3(256), 0, st, d_flags, d_bsum, d_slot, total, c->kp7_off, c->kp7, c->lcs, c->act_s, d_off, cap, d_edges, d_ab);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&ne, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
        const int pre = std::min(cap, PG_AB_PREFIX);               // the count is not known yet: the first pairs travel with it
        if (ends) HIPCHK(c, hipMemcpyAsync(c->pg_ab_host, d_ab, (size_t)pre * sizeof(int2), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (ne > cap) { dv.release(); DSSS_FAIL(c, DSSS_E_CAPACITY, "more than %d LC edges", cap); }
        if (ends) {
            if (ne > pre) { HIPCHK(c, hipMemcpyAsync(c->pg_ab_host + 2 * (size_t)pre, d_ab + pre, (size_t)(ne - pre) * sizeof(int2), hipMemcpyDeviceToHost, st)); HIPCHK(c, hipStreamSynchronize(st)); }
            if (ne > 0) HIPCHK(c, hipMemcpyAsync(edges, d_edges, (size_t)ne * sizeof(dsss_lc_edge), hipMemcpyDeviceToHost, st));
            *ends = c->pg_ab_host;
        }
        else if (ne > 0) HIPCHK(c, hipMemcpy(edges, d_edges, (size_t)ne * sizeof(dsss_lc_edge), hipMemcpyDeviceToHost));
        dv.release();                                              // synchronises the stream: REQUIRED before the arena is reused (see above)
    }
    else if (ends) *ends = nullptr;
    if (n_edges) *n_edges = ne;
    return DSSS_OK;
}

extern "C" {

int dsss_posegraph_select(dsss_ctx* c, int nframes, dsss_lc_edge* edges, int cap, int* n_edges) { return pg_select_impl(c, nframes, edges, cap, n_edges, nullptr); }

int dsss_posegraph_solve_edges(dsss_ctx* c, const double* dr6, int total, const dsss_lc_edge* edges, int ne, double* poses12, double* stats4)
{
    if (!c || !dr6 || total <= 0 || ne < 0 || (ne > 0 && !edges)) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<double> h_dr((size_t)total * 6);
    HIPCHK(c, hipMemcpy(h_dr.data(), dr6, h_dr.size() * sizeof(double), hipMemcpyDefault));
    std::vector<dsss_lc_edge> h_e(ne);
    if (ne) HIPCHK(c, hipMemcpy(h_e.data(), edges, (size_t)ne * sizeof(dsss_lc_edge), hipMemcpyDefault));
    return pg_solve_impl(c, h_dr.data(), total, h_e.data(), ne, poses12, stats4);
}

int dsss_posegraph_solve(dsss_ctx* c, int nframes, double* poses12, double* rpy6, double* stats4)
{
    if (!c || nframes <= 0 || nframes > c->max_frames) return DSSS_E_ARG;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    size_t total = 0;
    for (int f = 0; f < nframes; ++f) {
        if (!c->frames[f].has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no geometry", f);
        total += (size_t)c->frames[f].N;
    }
    // the selected edges come back into a page-locked buffer the context keeps (a fresh 7 MB vector per solve cost a millisecond of
    // page faults, and a pageable destination halves the copy rate)
    const size_t ecap = (size_t)std::max(c->total_kp7, 1);
    if (c->pg_edges_cap < ecap) {
        if (c->pg_edges_host) hipHostFree(c->pg_edges_host);
        c->pg_edges_host = nullptr; c->pg_edges_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->pg_edges_host, ecap * sizeof(dsss_lc_edge), hipHostMallocDefault));
        c->pg_edges_cap = ecap;
    }
    dsss_lc_edge* edges_p = static_cast<dsss_lc_edge*>(c->pg_edges_host);
    int ne = 0;
    const double t_dr = ms(t0);
    const auto t1 = std::chrono::steady_clock::now();
    const int world = dsss_comm_world(c), rank = dsss_comm_rank(c);
    const int* ends = nullptr;
    int rc = pg_select_impl(c, nframes, edges_p, (int)ecap, &ne, world == 1 ? &ends : nullptr);
    if (rc) return rc;
    if (world > 1) {
        // every rank selected the loop closures of the pairs it matched (pairs go to the owner of the TARGET frame, so a target
        // ping's "last pair wins" choice is rank-local): exchange them with two small all-reduces (counts, then the records in
        // rank order = ascending target pose, the reference's loop order)
        std::vector<double> cnt(world, 0.0); cnt[rank] = ne;
        auto xch = [&](size_t n) -> int {                    // device scratch of the exchange, kept by the context (a hipMalloc / hipFree pair per call cost 0.3 ms)
            if (c->xch_cap >= n) return DSSS_OK;
            HIPCHK(c, hipStreamSynchronize(c->stream));
            hipFree(c->xch_dev); c->xch_dev = nullptr; c->xch_cap = 0;
            const size_t cap = n + n / 2 + 1024;
            HIPCHK(c, hipMalloc(&c->xch_dev, cap * sizeof(double))); c->xch_cap = cap;
            return DSSS_OK;
        };
        rc = xch(world); if (rc) return rc;
        double* d_tmp = c->xch_dev;
        hipError_t e = hipMemcpyAsync(d_tmp, cnt.data(), world * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) { rc = dsss_comm_allreduce(c, d_tmp, world, c->stream); if (rc) return rc; }
        if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(), d_tmp, world * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        HIPCHK(c, e);
        // (round 5) the records travel as they are: ONE all-gather of equal slices (the largest rank's count of 152-byte records), own
        // slice in place.  Ranks own ascending blocks of target frames and every rank's edges ascend in the target pose: the
        // concatenation IS the reference's loop order (checked; sorted only if a caller's pair list broke that).  (Round 6) the slices are
        // closed up ON THE DEVICE and come back as one copy into page-locked memory, which is the edge list: rounds 2 - 5 copied them rank by
        // rank into a fresh vector -- at C5 53 MB of first-touch page faults, 15 of a rank's 20 ms here.  Rounds 2 - 4 packed twenty doubles
        // per edge into a zero-padded vector on the host, summed it over the ranks and sorted the result: 75 of a rank's 355 ms per C5 step.
        size_t tot = 0, maxc = 0;
        for (int r = 0; r < world; ++r) { tot += (size_t)cnt[r]; maxc = std::max(maxc, (size_t)cnt[r]); }
        if (tot > 0) {
            const size_t slice = maxc * sizeof(dsss_lc_edge), all = slice * (size_t)world;
            rc = xch((2 * all + sizeof(double) - 1) / sizeof(double)); if (rc) return rc;
            if (c->xch_host_cap < all) {
                if (c->xch_host) hipHostFree(c->xch_host);
                c->xch_host = nullptr; c->xch_host_cap = 0;
                HIPCHK(c, hipHostMalloc(&c->xch_host, all + all / 2, hipHostMallocDefault));
                c->xch_host_cap = all + all / 2;
            }
            char* d_all = reinterpret_cast<char*>(c->xch_dev);
            char* d_cmp = d_all + all;
            e = hipSuccess;
            if (ne > 0) e = hipMemcpyAsync(d_all + slice * (size_t)rank, edges_p, (size_t)ne * sizeof(dsss_lc_edge), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) { rc = dsss_comm_allgather(c, d_all, slice, c->stream); if (rc) return rc; }
            size_t w = 0;
            for (int r = 0; r < world && e == hipSuccess; ++r) {
                const size_t k = (size_t)cnt[r];
                if (k) e = hipMemcpyAsync(d_cmp + w * sizeof(dsss_lc_edge), d_all + slice * (size_t)r, k * sizeof(dsss_lc_edge), hipMemcpyDeviceToDevice, c->stream);
                w += k;
            }
            if (e == hipSuccess) e = hipMemcpyAsync(c->xch_host, d_cmp, tot * sizeof(dsss_lc_edge), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            HIPCHK(c, e);
            edges_p = static_cast<dsss_lc_edge*>(c->xch_host);
        }
        ne = (int)tot;
        bool ascending = true;
        for (int i = 1; i < ne && ascending; ++i) ascending = edges_p[i - 1].b <= edges_p[i].b;
        if (!ascending) std::stable_sort(edges_p, edges_p + ne, [](const dsss_lc_edge& x, const dsss_lc_edge& y) { return x.b < y.b; });
    }
    const double t_sel = ms(t1);
    const auto t2 = std::chrono::steady_clock::now();
    rc = pg_solve_impl(c, nullptr, (int)total, edges_p, ne, poses12, stats4, rpy6, nframes, ends);
    if (rc) return rc;
    if (getenv("DSSS_PG_VERBOSE")) fprintf(stderr, "[dsss pg] DR rows %.1f ms, LC selection %.1f ms, solve + download %.1f ms\n", t_dr, t_sel, ms(t2));
    return DSSS_OK;
}

/* N3, the online use of the solver (optimizer.cpp:134-139 ISAM2, :262-266 isam.update per ping + calculateEstimate): the
 * graph grows frame by frame.  Each call consumes the LC result set the context holds (if it has not been consumed yet) into
 * the accumulated edge list, and solves frames 0..nframes-1 by the batch LM STARTED FROM THE PREVIOUS ESTIMATE -- no factor is
 * carried over, the analysis and the factorisation are redone (they take milliseconds), but a converged prefix needs one
 * or two trials instead of the cold start's five.                                                                     */
int dsss_posegraph_reset(dsss_ctx* c)
{
    if (!c) return DSSS_E_ARG;
    c->pg_inc_edges.clear(); c->pg_inc_gen = c->lc_gen; c->pg_warm_n = 0;
    return DSSS_OK;
}

int dsss_posegraph_online_edges(dsss_ctx* c) { return c ? (int)c->pg_inc_edges.size() : DSSS_E_ARG; }

/* the panel levels of the last pose-graph solve (instrumentation: bench.py prices the dependent chains of the factorisation with it) */
int dsss_posegraph_schedule_get(dsss_ctx* c, int* levels4, int cap_levels, int* n_levels, int* n_trials)
{
    if (!c || (cap_levels > 0 && !levels4)) return DSSS_E_ARG;
    const int nl = (int)(c->pg_last_levels.size() / 4);
    if (n_levels) *n_levels = nl;
    if (n_trials) *n_trials = c->pg_last_trials;
    for (int l = 0; l < std::min(nl, cap_levels); ++l) for (int k = 0; k < 4; ++k) levels4[4 * l + k] = c->pg_last_levels[4 * (size_t)l + k];
    return DSSS_OK;
}

static int pg_update_impl(dsss_ctx* c, int nframes, int window_frames, double* poses12, double* rpy6, double* stats4)
{
    if (!c || nframes <= 0 || nframes > c->max_frames) return DSSS_E_ARG;
    if (dsss_comm_world(c) > 1) DSSS_FAIL(c, DSSS_E_STATE, "dsss_posegraph_update is a single-rank call (the online use is one vehicle, one GPU)");
    size_t total = 0;
    for (int f = 0; f < nframes; ++f) {
        if (!c->frames[f].has_geom) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no geometry", f);
        total += (size_t)c->frames[f].N;
    }
    if (c->has_lc && c->pg_inc_gen != c->lc_gen && c->total_kp7 > 0) {
        std::vector<dsss_lc_edge> fresh((size_t)c->total_kp7);
        int ne = 0;
        const int rc = dsss_posegraph_select(c, nframes, fresh.data(), (int)fresh.size(), &ne);
        if (rc) return rc;
        // a target ping keeps ONE loop closure, the latest (optimizer.cpp:203-258 within a call; across calls the later set wins)
        // (the accumulated edges are range-checked BEFORE they index anything: nframes may have gone down since the last update)
        for (const dsss_lc_edge& e : c->pg_inc_edges)
            if (e.a >= (int)total || e.b >= (int)total) DSSS_FAIL(c, DSSS_E_ARG, "an accumulated LC edge references ping %d of %zu: nframes went down; dsss_posegraph_reset first", std::max(e.a, e.b), total);
        if (ne > 0 && !c->pg_inc_edges.empty()) {
            std::vector<char> hit(total, 0);
            for (int e = 0; e < ne; ++e) hit[fresh[e].b] = 1;
            size_t w = 0;
            for (size_t e = 0; e < c->pg_inc_edges.size(); ++e) if (!hit[c->pg_inc_edges[e].b]) c->pg_inc_edges[w++] = c->pg_inc_edges[e];
            c->pg_inc_edges.resize(w);
        }
        // the fresh set ascends in the target ping; when it starts behind everything accumulated (a new frame: the usual case) appending keeps the
        // list sorted -- sorting 12 k records of 152 bytes on every update was the part of an update's cost that grew with the survey
        const bool append_only = ne == 0 || c->pg_inc_edges.empty() || c->pg_inc_edges.back().b <= fresh[0].b;
        c->pg_inc_edges.insert(c->pg_inc_edges.end(), fresh.begin(), fresh.begin() + ne);
        if (!append_only) std::stable_sort(c->pg_inc_edges.begin(), c->pg_inc_edges.end(), [](const dsss_lc_edge& x, const dsss_lc_edge& y) { return x.b < y.b; });
    }
    c->pg_inc_gen = c->lc_gen;
    for (const dsss_lc_edge& e : c->pg_inc_edges)
        if (e.a >= (int)total || e.b >= (int)total) DSSS_FAIL(c, DSSS_E_ARG, "an accumulated LC edge references ping %d of %zu: nframes went down; dsss_posegraph_reset first", std::max(e.a, e.b), total);
    // ---- the window (dsss_posegraph_update_window): frames f0 .. nframes - 1 are solved, conditioned on the frozen estimate of everything before
    // them.  Loop closures inside the window keep their form; one from a frozen pose a into the window becomes a BetweenFactor from the
    // window's pinned first pose with the measurement X_0^-1 X_a rel -- exactly the same residual, since X_a = X_0 (X_0^-1 X_a) with both
    // factors frozen (error = Log(rel^-1 X_a^-1 X_b) = Log((X_0^-1 X_a rel)^-1 X_0^-1 X_b)); closures between two frozen poses drop out.
    int f0 = window_frames > 0 ? std::max(0, nframes - window_frames) : 0;
    int p0 = 0;
    for (int f = 0; f < f0; ++f) p0 += c->frames[f].N;
    // the window's first ping anchors it and needs an estimate: a window that starts in frames no update has covered yet (a new frame with
    // window_frames = 1; several new frames at once) is extended backwards to the last frame that has one
    while (f0 > 0 && c->pg_warm_n <= p0) { --f0; p0 -= c->frames[f0].N; }
    if (f0 == 0) {                                       // no frozen part (yet): the whole graph
        c->pg_online = true; c->pg_win_f0 = 0; c->pg_win_p0 = 0;
        const int rc = pg_solve_impl(c, nullptr, (int)total, c->pg_inc_edges.data(), (int)c->pg_inc_edges.size(), poses12, stats4, rpy6, nframes);
        c->pg_online = false;
        return rc;
    }
    std::vector<dsss_lc_edge> we; std::vector<int> frozen;       // window edges; global ids of the frozen end points (in edge order)
    for (const dsss_lc_edge& e : c->pg_inc_edges) {
        if (std::max(e.a, e.b) < p0) continue;
        if (e.a >= p0 && e.b >= p0) { dsss_lc_edge w = e; w.a -= p0; w.b -= p0; we.push_back(w); continue; }
        if (e.a > e.b) DSSS_FAIL(c, DSSS_E_ARG, "window update: loop closure %d -> %d runs from the window into the frozen part (the pipeline's closures end in the later frame)", e.a, e.b);
        dsss_lc_edge w = e; w.b -= p0; w.a = -1 - (int)frozen.size(); frozen.push_back(e.a); we.push_back(w);
    }
    if (!frozen.empty()) {
        HIPCHK(c, hipSetDevice(c->device));
        // X_0 (the window's first pose) and the frozen end points: one gather out of the warm buffer
        // (scratch out of the solver's arena, which is idle between solves: no allocation per update)
        std::vector<int> idx(frozen); idx.push_back(p0);
        pg_dev gv;
        int* d_idx = nullptr; pose_t* d_g = nullptr;
        int rg = gv.alloc(c, &d_idx, idx.size()); if (rg) return rg;
        rg = gv.alloc(c, &d_g, idx.size()); if (rg) { gv.release(); return rg; }
        std::vector<pose_t> g(idx.size());
        hipError_t e = hipMemcpyAsync(d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) { hipLaunchKernelGGL(pg_gather_pose_kernel, dim3(((int)idx.size() + 255) / 256), dim3(256), 0, c->stream, (int)idx.size(), d_idx, static_cast<const pose_t*>(c->pg_warm), d_g); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipMemcpyAsync(g.data(), d_g, g.size() * sizeof(pose_t), hipMemcpyDeviceToHost, c->stream);
        gv.release();                                       // (synchronises the stream: the poses are on the host, the arena is free again)
        HIPCHK(c, e);
        pose_t X0inv; pose_inverse(&g.back(), &X0inv);
        for (dsss_lc_edge& w : we) {
            if (w.a >= 0) continue;
            const pose_t& Xa = g[(size_t)(-1 - w.a)];
            pose_t rel, Y, M;
            for (int k = 0; k < 9; ++k) rel.R[k] = w.rel[k];
            for (int k = 0; k < 3; ++k) rel.t[k] = w.rel[9 + k];
            pose_compose(&X0inv, &Xa, &Y); pose_compose(&Y, &rel, &M);
            for (int k = 0; k < 9; ++k) w.rel[k] = M.R[k];
            for (int k = 0; k < 3; ++k) w.rel[9 + k] = M.t[k];
            w.a = 0;
        }
    }
    // (several closures may now share the end point pair (0, b): the solver sums duplicates in edge order)
    c->pg_online = true; c->pg_win_f0 = f0; c->pg_win_p0 = p0;
    const int rc = pg_solve_impl(c, nullptr, (int)total - p0, we.data(), (int)we.size(), poses12, stats4, rpy6, nframes - f0);
    c->pg_online = false; c->pg_win_f0 = 0; c->pg_win_p0 = 0;
    return rc;
}

int dsss_posegraph_update(dsss_ctx* c, int nframes, double* poses12, double* rpy6, double* stats4) { return pg_update_impl(c, nframes, 0, poses12, rpy6, stats4); }
int dsss_posegraph_update_window(dsss_ctx* c, int nframes, int window_frames, double* poses12, double* rpy6, double* stats4)
{
    if (window_frames < 1) { if (c) c->err = "window_frames must be at least 1"; return DSSS_E_ARG; }
    return pg_update_impl(c, nframes, window_frames, poses12, rpy6, stats4);
}

} // extern "C"
