// diasss_amd/csrc/dsss_quadtree.hip -- K4: ORBextractor::DistributeOctTree on the device, one workgroup per
// (frame, pyramid level).  Restates /root/reference/thirdparty/ORBextractor.cpp:481-763 without the std::list:
//
//   * a node's keypoints are a contiguous segment of a key array; DivideNode is a STABLE 4-way partition of that
//     segment (one wave per node, ballot ranks), children living in the other of two ping-pong key buffers;
//   * the list is an array in list order.  "push_front the non-empty children, erase the parent" over a whole pass
//     (:606-665) gives  new list = reverse(children in creation order) ++ old leaves;  the size-ordered refinement
//     (:676-737) processes the expandable children of the previous round by descending (size, creation order),
//     stops at the first point where the node count reaches the quota (the reference's `break`), and gives
//     new list = reverse(children created this round) ++ (old list minus the divided parents);
//   * ties in the size sort use creation order (the reference compares heap addresses; same rule as the oracle and
//     the host routine in quadtree.cpp);
//   * nIni = max(1, round(w/h)) (the reference divides by zero for tall levels).
// Output: the kept candidate of every node (first maximum response in key order), in list order.
#include "dsss_internal.h"
#include "dsss_quadtree.h"

struct qnode { int x0, y0, x1, y1; int kbeg, kcnt; int buf; int leaf; };   // 32 B

// The workgroup of one (frame, level) instance: sixteen wavefronts.  The passes stream the level's whole key segment several times
// and the first passes have a handful of nodes, so the time of the kernel is the time of ONE workgroup on the largest level: four
// wavefronts took 3 ms per launch.
#define QT_THREADS 1024
#define QT_WAVES (QT_THREADS / 64)
#define QT_RANK_CAP 4096                           // expandable nodes whose (size, id) pairs fit the LDS staging of the rank pass
__device__ inline int qt_block_scan(int v, int* total, int* s_w)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < QT_WAVES; ++k) { const int t = s_w[k]; if (k < w) base += t; tot += t; }
    *total = tot;
    return base + inc - v;
}

// ExtractorNode::DivideNode (:481-537): stable 4-way partition of a node's key segment from its buffer into the other
// one; cnt[0..3] = keys of n1..n4 (top-left, top-right, bottom-left, bottom-right).  The key -> coordinate loads are
// streamed with QT_ILP chunks of 64 keys in flight per wave: a pass over the level costs (keys / (1024 x QT_ILP)) round trips.
#define QT_ILP 8
// A key carries the candidate's coordinates (integers: FAST cell offset + position in the cell) next to its index, so
// DivideNode streams the key segment and never gathers xs / ys.
typedef unsigned long long qkey;
#define QT_NOKEY (~0ull)
// The response rides along (FAST scores are integers below 256, the index keeps 24 bits): the point kept per node is then found
// from the keys alone.   key = y << 48 | x << 32 | response << 24 | index
__device__ inline qkey qt_make_key(int idx, float x, float y, float resp) { return ((qkey)(unsigned)(int)y << 48) | ((qkey)((unsigned)(int)x & 0xffffu) << 32) | ((qkey)((unsigned)(int)resp & 0xffu) << 24) | (qkey)((unsigned)idx & 0xffffffu); }
__device__ inline int qt_key_idx(qkey k) { return (int)(unsigned)(k & 0xffffffull); }
__device__ inline unsigned qt_key_rank(qkey k) { return ((unsigned)(k >> 24) & 0xffu) << 24 | (0xffffffu - (unsigned)(k & 0xffffffull)); }      // larger = higher response, then smaller index
__device__ inline void qt_classify4(const qkey* __restrict__ src, int beg, int end, int b, int lane, float mx, float my, int* c, qkey* k)
{
#pragma unroll
    for (int u = 0; u < QT_ILP; ++u) { const int i = b + u * 64 + lane; k[u] = i < end ? src[beg + i] : QT_NOKEY; }
#pragma unroll
    for (int u = 0; u < QT_ILP; ++u) {
        const float x = (float)(int)((k[u] >> 32) & 0xffffull), y = (float)(int)(k[u] >> 48);
        const bool left = x < mx, top = y < my;
        c[u] = k[u] != QT_NOKEY ? (left ? (top ? 0 : 2) : (top ? 1 : 3)) : -1;
    }
}
// counts of the four classes over keys [lo, hi) of the node (one wave)
__device__ inline void qt_count_range(const qnode& P, const qkey* __restrict__ src, int lo, int hi, float mx, float my, int* cnt)
{
    const int lane = threadIdx.x & 63;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int b = lo; b < hi; b += 64 * QT_ILP) {
        int c[QT_ILP]; qkey k[QT_ILP];
        qt_classify4(src, P.kbeg, hi, b, lane, mx, my, c, k);
#pragma unroll
        for (int u = 0; u < QT_ILP; ++u) {
            c0 += __popcll(__ballot(c[u] == 0)); c1 += __popcll(__ballot(c[u] == 1)); c2 += __popcll(__ballot(c[u] == 2)); c3 += __popcll(__ballot(c[u] == 3));
        }
    }
    cnt[0] = c0; cnt[1] = c1; cnt[2] = c2; cnt[3] = c3;
}
// stable scatter of keys [lo, hi) given the destination offset of each class for this range (one wave)
__device__ inline void qt_scatter_range(const qnode& P, const qkey* __restrict__ src, qkey* __restrict__ dst, int lo, int hi,
                                        float mx, float my, const int* off)
{
    const int lane = threadIdx.x & 63;
    int r0 = off[0], r1 = off[1], r2 = off[2], r3 = off[3];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int b = lo; b < hi; b += 64 * QT_ILP) {
        int c[QT_ILP]; qkey k[QT_ILP];
        qt_classify4(src, P.kbeg, hi, b, lane, mx, my, c, k);
#pragma unroll
        for (int u = 0; u < QT_ILP; ++u) {
            const unsigned long long m0 = __ballot(c[u] == 0), m1 = __ballot(c[u] == 1), m2 = __ballot(c[u] == 2), m3 = __ballot(c[u] == 3);
            if (c[u] == 0) dst[P.kbeg + r0 + __popcll(m0 & below)] = k[u];
            else if (c[u] == 1) dst[P.kbeg + r1 + __popcll(m1 & below)] = k[u];
            else if (c[u] == 2) dst[P.kbeg + r2 + __popcll(m2 & below)] = k[u];
            else if (c[u] == 3) dst[P.kbeg + r3 + __popcll(m3 & below)] = k[u];
            r0 += __popcll(m0); r1 += __popcll(m1); r2 += __popcll(m2); r3 += __popcll(m3);
        }
    }
}
// one node, one wave
__device__ inline void qt_divide_wave(const qnode& P, qkey* __restrict__ keys0, qkey* __restrict__ keys1, int* cnt)
{
    const qkey* src = P.buf ? keys1 : keys0;
    qkey* dst = P.buf ? keys0 : keys1;
    const float mx = (float)(P.x0 + (int)ceilf((float)(P.x1 - P.x0) / 2));
    const float my = (float)(P.y0 + (int)ceilf((float)(P.y1 - P.y0) / 2));
    qt_count_range(P, src, 0, P.kcnt, mx, my, cnt);
    const int off[4] = { 0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2] };
    qt_scatter_range(P, src, dst, 0, P.kcnt, mx, my, off);
}
// one node, the whole workgroup (the first passes have fewer nodes than waves, and those nodes hold most of the keys):
// every wave takes a contiguous share of the segment; s_cnt[wave][class] carries the counts between the two passes
__device__ inline void qt_divide_block(const qnode& P, qkey* __restrict__ keys0, qkey* __restrict__ keys1, int* cnt, int (*s_cnt)[4])
{
    const qkey* src = P.buf ? keys1 : keys0;
    qkey* dst = P.buf ? keys0 : keys1;
    const float mx = (float)(P.x0 + (int)ceilf((float)(P.x1 - P.x0) / 2));
    const float my = (float)(P.y0 + (int)ceilf((float)(P.y1 - P.y0) / 2));
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = ((P.kcnt + QT_WAVES - 1) / QT_WAVES + 63) & ~63;
    const int lo = min(wv * q, P.kcnt), hi = min(lo + q, P.kcnt);
    int mine[4];
    qt_count_range(P, src, lo, hi, mx, my, mine);
    if (lane == 0) { s_cnt[wv][0] = mine[0]; s_cnt[wv][1] = mine[1]; s_cnt[wv][2] = mine[2]; s_cnt[wv][3] = mine[3]; }
    __syncthreads();
    int off[4], run = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < QT_WAVES; ++w) { const int t = s_cnt[w][c]; if (w < wv) before += t; tot += t; }
        off[c] = run + before; cnt[c] = tot; run += tot;
    }
    qt_scatter_range(P, src, dst, lo, hi, mx, my, off);
    __syncthreads();
}

// qnode::leaf: bit 0 = the node holds one key; for the nodes of a PRE-SORTED level (see quadtree_kernel) bits 4..6 = depth and
// bits 8.. = the path code (two bits per level), which is also where the node's keys start in the sorted array
__device__ inline qnode qt_child(const qnode& P, int c, const int* cnt, bool presorted = false)
{
    const int mx = P.x0 + (int)ceilf((float)(P.x1 - P.x0) / 2), my = P.y0 + (int)ceilf((float)(P.y1 - P.y0) / 2);
    qnode q;
    q.x0 = (c & 1) ? mx : P.x0; q.x1 = (c & 1) ? P.x1 : mx;
    q.y0 = (c & 2) ? my : P.y0; q.y1 = (c & 2) ? P.y1 : my;
    int off = 0;
    for (int k = 0; k < c; ++k) off += cnt[k];
    q.kbeg = P.kbeg + off; q.kcnt = cnt[c]; q.buf = presorted ? 1 : P.buf ^ 1; q.leaf = cnt[c] == 1;
    if (presorted) q.leaf |= ((((P.leaf >> 4) & 7) + 1) << 4) | ((((P.leaf >> 8) << 2) | c) << 8);
    return q;
}
// path code of a point after D levels of DivideNode starting from the root box (0, 0, W, H): two bits per level, n1..n4 = 0..3
__device__ inline int qt_code(float x, float y, int W, int H, int D)
{
    int x0 = 0, y0 = 0, x1 = W, y1 = H, code = 0;
    for (int l = 0; l < D; ++l) {
        const int mx = x0 + (int)ceilf((float)(x1 - x0) / 2), my = y0 + (int)ceilf((float)(y1 - y0) / 2);
        const bool left = x < (float)mx, top = y < (float)my;
        code = (code << 2) | (left ? (top ? 0 : 2) : (top ? 1 : 3));
        if (left) x1 = mx; else x0 = mx;
        if (top) y1 = my; else y0 = my;
    }
    return code;
}

// work layout per instance (ints): pool (qnode x pool_cap) | listA listB parents exp order flags (cap each) | pcnt[4*cap]
__global__ __launch_bounds__(QT_THREADS) void quadtree_kernel(const qt_inst* __restrict__ tab)
{
    __shared__ int s_w[QT_WAVES];
    __shared__ int s_S, s_pool, s_nexp, s_done, s_phase2, s_t;
    __shared__ int s_cnt[QT_WAVES][4];
    __shared__ __attribute__((aligned(16))) int s_big[2 * QT_RANK_CAP];           // phase 2: (size, id) pairs
    __shared__ int s_off[QT_RANK_CAP + 8];             // bucket offsets of the D-level path codes (kept to the end: every division above depth D reads its children's sizes here)
    __shared__ unsigned s_best[QT_RANK_CAP];           // per bucket: the best (response, smallest index) rank of its keys; the cursors of the key scatter once keys are needed
    __shared__ int s_have_keys;
    int2* s_rank = reinterpret_cast<int2*>(s_big);
    int* s_cur = reinterpret_cast<int*>(s_best);
    const qt_inst I = tab[blockIdx.x];
    const int base = I.offs[I.cell_begin];
    const int n = min(I.offs[I.cell_end], I.cand_cap) - base;         // never index past the candidate arrays (the overflow itself is flagged by scan_counts_kernel)
    const float* xs = I.xs + base; const float* ys = I.ys + base; const float* rs = I.rs + base;
    int* out = I.out_idx; int* out_n = I.out_n;
    if (n <= 0) { if (threadIdx.x == 0) *out_n = 0; return; }
    if (n >= (1 << 24)) { if (threadIdx.x == 0) { *out_n = 0; *I.err = 5; } return; }      // the key keeps 24 bits of the candidate index
    const int cap = I.list_cap, pool_cap = I.pool_cap, N = I.quota;
    qkey* keys0 = I.keys0 + base; qkey* keys1 = I.keys1 + base;      // frame-wide key arrays, this level's segment
    qnode* pool = reinterpret_cast<qnode*>(I.work);
    int* listA = reinterpret_cast<int*>(pool + pool_cap); int* listB = listA + cap;
    int* parents = listB + cap; int* expv = parents + cap; int* order = expv + cap; int* flags = order + cap; int* pcnt = flags + cap;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    // ---- initial nodes (:543-585)
    int nIni = (int)roundf((float)I.W / (float)I.H);
    if (nIni < 1) nIni = 1;
    if (nIni > 32) nIni = 32;
    const float hX = (float)I.W / nIni;
    if (nIni == 1) {
        if (N <= 1)                                  // (with a quota above one the pre-sort below writes the keys)
        for (int i0 = threadIdx.x; i0 < n; i0 += 8 * QT_THREADS) {      // eight coordinate pairs in flight per thread
            float x8[8], y8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * QT_THREADS; x8[u] = i < n ? xs[i] : 0.f; y8[u] = i < n ? ys[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * QT_THREADS; if (i < n) keys0[i] = qt_make_key(i, x8[u], y8[u], rs[i]); }
        }
    }
    else if (wv == 0) {                              // stable partition of 0..n-1 by root index, one wave
        int off = 0;
        for (int r = 0; r < nIni; ++r) {
            int run = 0;
            for (int b = 0; b < n; b += 64) {
                const int i = b + lane;
                bool in = false;
                if (i < n) { int w = (int)(xs[i] / hX); if (w >= nIni) w = nIni - 1; in = (w == r); }
                const unsigned long long m = __ballot(in);
                if (in) keys0[off + run + __popcll(m & ((1ull << lane) - 1ull))] = qt_make_key(i, xs[i], ys[i], rs[i]);
                run += __popcll(m);
            }
            if (lane == 0) pcnt[r] = run;
            off += run;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int S = 0, off = 0;
        for (int r = 0; r < nIni; ++r) {
            const int c = nIni == 1 ? n : pcnt[r];
            qnode q; q.x0 = (int)(hX * (float)r); q.x1 = (int)(hX * (float)(r + 1)); q.y0 = 0; q.y1 = I.H;
            q.kbeg = off; q.kcnt = c; q.buf = 0; q.leaf = c == 1;
            off += c;
            pool[r] = q;
            if (c > 0) listA[S++] = r;              // empty roots are erased (:581-582)
        }
        s_S = S; s_pool = nIni; s_done = 0; s_phase2 = 0; s_nexp = 0;
    }
    __syncthreads();
    int* L = listA; int* Ln = listB;

    // ---- PRE-SORT.  The whole-list passes below divide every node of the list, so after p of them the keys are bucket-sorted by
    // their p-level path code.  With one root the code of a key follows from its coordinates alone (the boxes of DivideNode are a
    // function of the path), so ONE counting sort by the D-level code replaces the first D passes, each of which streamed the
    // level's keys twice and wrote them once through the single compute unit this workgroup runs on (the kernel's bound).  The
    // passes keep their node bookkeeping and read the children's sizes from the bucket offsets.  Sorting deeper than the passes
    // go is harmless: a node's keys are a contiguous range whatever their order inside it, and the point kept per node is picked
    // by (response, candidate index), not by position.
    int D = 0;
    if (nIni == 1) { int cells = 1; while (cells < N && D < 6) { cells *= 4; ++D; } }
    const bool presorted = D > 0;
    // Round 4: the keys themselves are NOT written any more unless a division below depth D turns up.  Everything the tree needs
    // above depth D is the SIZE of a bucket range, and the point a node keeps is the maximum of (response, smallest index) over its
    // buckets -- both come out of ONE pass over the candidates (LDS histogram + LDS atomicMax).  A level-0 instance of a 2000 x 1024
    // frame (92 k candidates, quota 501: sorted to depth 5, divided to depth 5) read its candidates twice, wrote 8-byte keys, streamed
    // them twice more and wrote them again in the size-ordered round, and read them once more for the kept points -- through the one
    // compute unit it runs on; now it reads them once.  ensure_keys() below is the old second pass, run on demand.
    if (threadIdx.x == 0) s_have_keys = presorted ? 0 : 1;
    if (presorted) {
        const int nb = 1 << (2 * D);
        for (int i = threadIdx.x; i <= nb; i += QT_THREADS) { s_off[i] = 0; if (i < nb) s_best[i] = 0u; }
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < n; i0 += 8 * QT_THREADS) {
            float x8[8], y8[8], r8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * QT_THREADS; x8[u] = i < n ? xs[i] : 0.f; y8[u] = i < n ? ys[i] : 0.f; r8[u] = i < n ? rs[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * QT_THREADS;
                if (i < n) {
                    const int code = qt_code((float)(int)x8[u], (float)(int)y8[u], I.W, I.H, D);
                    atomicAdd(&s_off[code], 1);
                    atomicMax(&s_best[code], (((unsigned)(int)r8[u] & 0xffu) << 24) | (0xffffffu - (unsigned)i));      // = qt_key_rank of the key this candidate would get
                }
            }
        }
        __syncthreads();
        {   // exclusive scan of the nb bucket sizes, four consecutive buckets per thread
            const int b0 = 4 * threadIdx.x;
            int v[4], sum = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { v[u] = b0 + u < nb ? s_off[b0 + u] : 0; sum += v[u]; }
            int tot;
            int run = qt_block_scan(sum, &tot, s_w);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) if (b0 + u < nb) { s_off[b0 + u] = run; run += v[u]; }
            if (threadIdx.x == 0) s_off[nb] = n;
        }
        if (threadIdx.x == 0) pool[0].buf = 1;        // the root's keys are the sorted array (once it exists)
        __syncthreads();
    }
    // the sorted key array, on demand (workgroup-uniform calls only): counting-sort scatter by the D-level code into keys1
    auto ensure_keys = [&]() {
        if (s_have_keys) return;
        const int nb = 1 << (2 * D);
        __syncthreads();
        for (int i = threadIdx.x; i < nb; i += QT_THREADS) s_cur[i] = s_off[i];      // (the bucket maxima are gone from here on: the kept points come from the keys)
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < n; i0 += 8 * QT_THREADS) {
            float x8[8], y8[8], r8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * QT_THREADS; x8[u] = i < n ? xs[i] : 0.f; y8[u] = i < n ? ys[i] : 0.f; r8[u] = i < n ? rs[i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * QT_THREADS;
                if (i < n) keys1[atomicAdd(&s_cur[qt_code((float)(int)x8[u], (float)(int)y8[u], I.W, I.H, D)], 1)] = qt_make_key(i, x8[u], y8[u], r8[u]);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_have_keys = 1;
        __syncthreads();
    };
    int pass = 0;

    // ---- whole-list passes (:594-672)
    while (true) {
        const int S = s_S;
        // non-leaf nodes in list order -> parents[], leaves keep their relative order
        int np = 0, nl = 0;
        for (int b = 0; b < S; b += QT_THREADS) {
            const int i = b + threadIdx.x;
            const int id = i < S ? L[i] : -1;
            const int isp = (id >= 0 && !(pool[id].leaf & 1)) ? 1 : 0, isl = (id >= 0 && (pool[id].leaf & 1)) ? 1 : 0;
            int tp, tl;
            const int rp = qt_block_scan(isp, &tp, s_w), rl = qt_block_scan(isl, &tl, s_w);
            if (isp) parents[np + rp] = id;
            if (isl) flags[nl + rl] = id;            // flags[] doubles as the leaf list here
            np += tp; nl += tl;
        }
        __syncthreads();
        if (np == 0) break;                          // every node holds one point: size == prevSize
        if (s_pool + 4 * np > pool_cap) { if (threadIdx.x == 0) *I.err = 1; break; }
        const bool from_buckets = presorted && pass < D;       // the parents of this pass sit at depth `pass`: their children are bucket ranges
        if (from_buckets) {
            const int sh = 2 * (D - pass - 1);
            for (int r = threadIdx.x; r < np; r += QT_THREADS) {
                const int pf = pool[parents[r]].leaf >> 8;
#pragma unroll
                for (int c = 0; c < 4; ++c) { const int cp = (pf << 2) | c; pcnt[4 * r + c] = s_off[(cp + 1) << sh] - s_off[cp << sh]; }
            }
        } else if (np < QT_WAVES) {
            ensure_keys();
            for (int r = 0; r < np; ++r) {
                int cnt[4];
                qt_divide_block(pool[parents[r]], keys0, keys1, cnt, s_cnt);
                if (threadIdx.x == 0) { pcnt[4 * r] = cnt[0]; pcnt[4 * r + 1] = cnt[1]; pcnt[4 * r + 2] = cnt[2]; pcnt[4 * r + 3] = cnt[3]; }
            }
        } else {
            ensure_keys();
            for (int r = wv; r < np; r += QT_WAVES) {
                int cnt[4];
                qt_divide_wave(pool[parents[r]], keys0, keys1, cnt);
                if (lane == 0) { pcnt[4 * r] = cnt[0]; pcnt[4 * r + 1] = cnt[1]; pcnt[4 * r + 2] = cnt[2]; pcnt[4 * r + 3] = cnt[3]; }
            }
        }
        __syncthreads();
        // children in creation order: parents in list order, n1..n4, empty ones skipped
        int Cn = 0, nexp = 0;
        const int pool0 = s_pool;
        for (int b = 0; b < np; b += QT_THREADS) {
            const int r = b + threadIdx.x;
            int ne = 0, nx = 0, cnt[4] = { 0, 0, 0, 0 };
            if (r < np) for (int c = 0; c < 4; ++c) { cnt[c] = pcnt[4 * r + c]; ne += cnt[c] > 0; nx += cnt[c] > 1; }
            int te, tx;
            const int qe = qt_block_scan(ne, &te, s_w), qx = qt_block_scan(nx, &tx, s_w);
            if (r < np) {
                const qnode P = pool[parents[r]];
                int q = Cn + qe, x = nexp + qx;
                for (int c = 0; c < 4; ++c) if (cnt[c] > 0) {
                    pool[pool0 + q] = qt_child(P, c, cnt, from_buckets);
                    if (cnt[c] > 1) expv[x++] = pool0 + q;
                    ++q;
                }
            }
            Cn += te; nexp += tx;
        }
        ++pass;
        __syncthreads();
        const int Snew = Cn + nl;
        if (Snew > cap) { if (threadIdx.x == 0) *I.err = 2; break; }
        for (int i = threadIdx.x; i < Cn; i += QT_THREADS) Ln[Cn - 1 - i] = pool0 + i;      // push_front order
        for (int i = threadIdx.x; i < nl; i += QT_THREADS) Ln[Cn + i] = flags[i];
        __syncthreads();
        if (threadIdx.x == 0) {
            s_pool = pool0 + Cn; s_nexp = nexp;
            if (Snew >= N || Snew == S) s_done = 1;
            else if (Snew + nexp * 3 > N) s_phase2 = 1;
            s_S = Snew;
        }
        { int* t = L; L = Ln; Ln = t; }
        __syncthreads();
        if (s_done || s_phase2) break;
    }

    // ---- size-ordered refinement (:673-738)
    while (s_phase2 && !s_done) {
        const int S = s_S, m = s_nexp, pool0 = s_pool;
        if (m == 0) break;
        if (pool0 + 4 * m > pool_cap) { if (threadIdx.x == 0) *I.err = 1; break; }
        // processing order: descending (size, creation order).  The m x m comparison runs on (size, id) pairs staged in LDS: straight
        // from the node pool every comparison was two dependent global loads (0.6 ms per round at m = 600)
        if (m <= QT_RANK_CAP) {
            for (int e = threadIdx.x; e < m; e += QT_THREADS) { const int id = expv[e]; s_rank[e] = make_int2(pool[id].kcnt, id); }
            __syncthreads();
            for (int e = threadIdx.x; e < m; e += QT_THREADS) {
                const int2 me = s_rank[e];
                int rank = 0;
                for (int f = 0; f < m; ++f) { const int2 o = s_rank[f]; rank += (o.x > me.x) || (o.x == me.x && o.y > me.y); }
                order[rank] = me.y;
            }
        } else
            for (int e = threadIdx.x; e < m; e += QT_THREADS) {
                const int id = expv[e], sz = pool[id].kcnt;
                int rank = 0;
                for (int f = 0; f < m; ++f) { const int id2 = expv[f], s2 = pool[id2].kcnt; rank += (s2 > sz) || (s2 == sz && id2 > id); }
                order[rank] = id;
            }
        __syncthreads();
        // the nodes of a round were created together: when they sit above depth D (a pre-sorted level) their children are bucket ranges
        // and nothing is streamed; otherwise every one of them is divided on the keys (only the first t + 1 take effect)
        bool r2_buckets = false;
        if (presorted && !s_have_keys) {
            int deep = 0;
            for (int e = threadIdx.x; e < m; e += QT_THREADS) { const int dp = (pool[order[e]].leaf >> 4) & 7; deep |= (dp == 0 || dp >= D); }
            r2_buckets = __syncthreads_or(deep) == 0;
        }
        if (r2_buckets) {
            for (int r = threadIdx.x; r < m; r += QT_THREADS) {
                const int lf = pool[order[r]].leaf, pf = lf >> 8, sh = 2 * (D - ((lf >> 4) & 7) - 1);
#pragma unroll
                for (int c = 0; c < 4; ++c) { const int cp = (pf << 2) | c; pcnt[4 * r + c] = s_off[(cp + 1) << sh] - s_off[cp << sh]; }
            }
        } else {
            ensure_keys();
            for (int r = wv; r < m; r += QT_WAVES) {
                int cnt[4];
                qt_divide_wave(pool[order[r]], keys0, keys1, cnt);
                if (lane == 0) { pcnt[4 * r] = cnt[0]; pcnt[4 * r + 1] = cnt[1]; pcnt[4 * r + 2] = cnt[2]; pcnt[4 * r + 3] = cnt[3]; }
            }
        }
        if (threadIdx.x == 0) s_t = m - 1;
        __syncthreads();
        // running node count; the reference breaks once it reaches the quota (:730-731)
        int run = S;
        for (int b = 0; b < m; b += QT_THREADS) {
            const int r = b + threadIdx.x;
            int d = 0;
            if (r < m) { for (int c = 0; c < 4; ++c) d += pcnt[4 * r + c] > 0; d -= 1; }
            int td;
            const int ex = qt_block_scan(d, &td, s_w);
            if (r < m && run + ex + d >= N) atomicMin(&s_t, r);
            run += td;
            __syncthreads();
        }
        __syncthreads();
        const int t = s_t;                           // parents order[0..t] are divided
        int Cn = 0, nexp = 0;
        for (int b = 0; b <= t; b += QT_THREADS) {
            const int r = b + threadIdx.x;
            int ne = 0, nx = 0, cnt[4] = { 0, 0, 0, 0 };
            if (r <= t) for (int c = 0; c < 4; ++c) { cnt[c] = pcnt[4 * r + c]; ne += cnt[c] > 0; nx += cnt[c] > 1; }
            int te, tx;
            const int qe = qt_block_scan(ne, &te, s_w), qx = qt_block_scan(nx, &tx, s_w);
            if (r <= t) {
                const qnode P = pool[order[r]];
                int q = Cn + qe, x = nexp + qx;
                for (int c = 0; c < 4; ++c) if (cnt[c] > 0) {
                    pool[pool0 + q] = qt_child(P, c, cnt, r2_buckets);
                    if (cnt[c] > 1) parents[x++] = pool0 + q;     // next round's expandables (parents[] is free here)
                    ++q;
                }
                pool[order[r]].kcnt = -1;                          // erased (:728)
            }
            Cn += te; nexp += tx;
        }
        __syncthreads();
        // new list = reverse(children) ++ old list without the divided parents
        int kept = 0;
        for (int b = 0; b < S; b += QT_THREADS) {
            const int i = b + threadIdx.x;
            const int id = i < S ? L[i] : -1;
            const int keep = (id >= 0 && pool[id].kcnt >= 0) ? 1 : 0;
            int tk;
            const int rk = qt_block_scan(keep, &tk, s_w);
            if (keep) { if (Cn + kept + rk < cap) Ln[Cn + kept + rk] = id; }
            kept += tk;
        }
        const int Snew = Cn + kept;
        if (Snew > cap) { if (threadIdx.x == 0) *I.err = 2; break; }
        for (int i = threadIdx.x; i < Cn; i += QT_THREADS) Ln[Cn - 1 - i] = pool0 + i;
        __syncthreads();
        for (int i = threadIdx.x; i < nexp; i += QT_THREADS) expv[i] = parents[i];
        if (threadIdx.x == 0) {
            s_pool = pool0 + Cn; s_nexp = nexp; s_S = Snew;
            if (Snew >= N || Snew == S) s_done = 1;
        }
        { int* tt = L; L = Ln; Ln = tt; }
        __syncthreads();
    }
    __syncthreads();

    // ---- retain the best point of each node, list order (:741-760): first maximum response in key order
    const int S = s_S;
    if (!s_have_keys) {                                // no key was ever written: a node is a range of buckets, its point the best of their maxima
        for (int i = wv; i < S; i += QT_WAVES) {
            const int lf = pool[L[i]].leaf, sh = 2 * (D - ((lf >> 4) & 7)), b0 = (lf >> 8) << sh, b1 = ((lf >> 8) + 1) << sh;
            unsigned best = 0;
            for (int k = b0 + lane; k < b1; k += 64) best = max(best, s_best[k]);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o, 64));
            best = 0xffffffu - (best & 0xffffffu);
            if (lane == 0 && i < I.out_cap) out[i] = base + best;
        }
        if (threadIdx.x == 0) { *out_n = S < I.out_cap ? S : I.out_cap; if (S > I.out_cap) *I.err = 3; }
        return;
    }
    for (int i = wv; i < S; i += QT_WAVES) {           // one wavefront per node: keys read together, first maximum by (response, candidate index)
        const qnode q = pool[L[i]];
        const qkey* kk = (q.buf ? keys1 : keys0) + q.kbeg;
        unsigned best = 0;                                       // keys of a node are in candidate order in the reference (stable partitions): its
        for (int k = lane; k < q.kcnt; k += 64) best = max(best, qt_key_rank(kk[k]));      // "first maximum" = the smallest candidate index among the maxima
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, o, 64));
        best = 0xffffffu - (best & 0xffffffu);
        if (lane == 0 && i < I.out_cap) out[i] = base + best;
    }
    if (threadIdx.x == 0) { *out_n = S < I.out_cap ? S : I.out_cap; if (S > I.out_cap) *I.err = 3; }
}

// concatenate the kept candidates of the levels of one frame into kp_in records (ORBextractor.cpp:840-847,1082-1111)
__global__ __launch_bounds__(256) void quadtree_collect_kernel(const qt_frame* __restrict__ tab)
{
    const qt_frame Fm = tab[blockIdx.x];
    int off = 0;
    for (int l = 0; l < Fm.nlevels; ++l) {
        const int cnt = Fm.out_n[l];
        const int* idx = Fm.out_idx + (size_t)l * Fm.out_cap;
        for (int i = threadIdx.x; i < cnt; i += 256) {
            if (off + i >= Fm.kcap) break;
            const int c = idx[i];
            qt_kp_in q;
            q.x = Fm.xs[c] + (float)Fm.min_border; q.y = Fm.ys[c] + (float)Fm.min_border; q.resp = Fm.rs[c]; q.level = l;
            Fm.kin[off + i] = q;
        }
        off += cnt;
    }
    if (threadIdx.x == 0) { if (off > Fm.kcap) { *Fm.err = 4; off = Fm.kcap; } *Fm.nk = off; }
}

void dsss_launch_quadtree(hipStream_t st, const qt_inst* d_inst, int ninst)
{
    if (ninst > 0) hipLaunchKernelGGL(quadtree_kernel, dim3(ninst), dim3(QT_THREADS), 0, st, d_inst);
}
void dsss_launch_quadtree_collect(hipStream_t st, const qt_frame* d_frames, int nframes)
{
    if (nframes > 0) hipLaunchKernelGGL(quadtree_collect_kernel, dim3(nframes), dim3(256), 0, st, d_frames);
}
