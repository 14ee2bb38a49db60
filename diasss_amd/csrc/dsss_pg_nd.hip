// diasss_amd/csrc/dsss_pg_nd.hip -- the nested dissection of the reduced pose graph ON THE DEVICE (gfx950), one partition.
//
// The ordering is the first thing the analysis of a solve needs and the GPU idles while the host produces it (1.1 ms at C3 behind
// 0.3 ms of adjacency; DESIGN.md section 4 "Round 4").  This file restates nd_order() of dsss_pg_sym.cpp breadth-first: the same
// median cuts in the (coordinate, index) order, the same candidates (the longer extent; both axes from 64 nodes on, the smaller
// separator wins, ties to the longer extent), the same separators (lower-half nodes with a neighbour in the upper half), leaves and
// degenerate cuts in index order -- so the elimination order is THE SAME ARRAY the host produces (checked against it by
// tests/test_gpu_switches.py through DSSS_PG_ND=check).
//   ranks      rank_x[v] = number of nodes before v in the (x, index) order, rank_y likewise: one kernel of n^2 / 1024^2 workgroups
//   adjacency  CSR of the reduced edges by atomics (duplicates stay: a cut only asks whether a neighbour exists)
//   levels     one launch per level of the recursion, one workgroup per node set (heap numbering h, children 2 h and 2 h + 1): bounding
//              box, the median rank of each candidate axis by a two-pass histogram selection, the cut counts, the choice, a stable
//              three-way partition in index order -- lower half without its separator | upper half into the next level's node list,
//              the separator straight to its place at the end of the set's range of the order.
#include "dsss_internal.h"
#include "dsss_pg_nd.h"

namespace {

#define ND_T 1024

__device__ inline double nd_shfl_d(double v, int o) { return __shfl_xor(v, o, 64); }

struct nd_lds { double red[4][ND_T / 64]; int ired[3][ND_T / 64]; unsigned hist2[512]; int sel[4]; int wtot[2][4]; };

__device__ inline int nd_block_sum(int v, int* slot, int which)            // every thread gets the sum over the workgroup
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) slot[which * (ND_T / 64) + w] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int k = 0; k < ND_T / 64; ++k) s += slot[which * (ND_T / 64) + k];
    return s;
}

// the bin that holds rank k of each of the two 256-bin histograms (x: threads 0..255, y: 256..511), and what is left of k inside it:
// S.sel = { bin x, rest x, bin y, rest y }.  (One thread walking the bins was 256 dependent LDS reads: 13 us a pass, the whole
// fixed cost of a level of small sets.)  Every thread of the workgroup calls it; the histograms must be complete (barrier before).
__device__ inline void nd_find_bins(nd_lds& S, int kx, int ky)
{
    const int a = (threadIdx.x >> 8) & 1, t = threadIdx.x & 255, lane = threadIdx.x & 63, wg = (threadIdx.x >> 6) & 3;
    const bool on = threadIdx.x < 512;
    const int val = on ? (int)S.hist2[a * 256 + t] : 0;
    int inc = val;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(inc, o, 64); if (lane >= o) inc += x; }
    if (on && lane == 63) S.wtot[a][wg] = inc;
    __syncthreads();
    if (on) {
        int base = 0;
        for (int k = 0; k < wg; ++k) base += S.wtot[a][k];
        const int incl = base + inc, excl = incl - val, k = a ? ky : kx;
        if (excl <= k && k < incl) { S.sel[2 * a] = t; S.sel[2 * a + 1] = k - excl; }
    }
    __syncthreads();
}

// ---- ranks in the (coordinate, index) order of both axes.  All pairs in one kernel took 153 us at 23 k nodes; instead the nodes are
// thrown into 1024 buckets per axis by their coordinate (a monotone map, so a bucket's nodes come before those of every later bucket) and
// a node's rank is the start of its bucket plus the number of bucket members that come before it.
#define ND_NB 1024
struct nd_rk { double lo[2], scale[2]; };
__device__ inline int nd_bucket(double v, double lo, double scale) { const int b = (int)((v - lo) * scale); return b < 0 ? 0 : (b >= ND_NB ? ND_NB - 1 : b); }
__global__ __launch_bounds__(ND_T) void nd_minmax_kernel(int n, const double* __restrict__ sxy, nd_rk* __restrict__ rk, int* __restrict__ hist)
{
    __shared__ double red[4][ND_T / 64];
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int i = threadIdx.x; i < n; i += ND_T) { const double x = sxy[2 * (size_t)i], y = sxy[2 * (size_t)i + 1]; x0 = fmin(x0, x); x1 = fmax(x1, x); y0 = fmin(y0, y); y1 = fmax(y1, y); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { x0 = fmin(x0, nd_shfl_d(x0, o)); x1 = fmax(x1, nd_shfl_d(x1, o)); y0 = fmin(y0, nd_shfl_d(y0, o)); y1 = fmax(y1, nd_shfl_d(y1, o)); }
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red[0][w] = x0; red[1][w] = x1; red[2][w] = y0; red[3][w] = y1; }
    for (int i = threadIdx.x; i < 4 * ND_NB; i += ND_T) hist[i] = 0;          // [histx | histy | curx | cury]
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < ND_T / 64; ++k) { x0 = fmin(x0, red[0][k]); x1 = fmax(x1, red[1][k]); y0 = fmin(y0, red[2][k]); y1 = fmax(y1, red[3][k]); }
        nd_rk r; r.lo[0] = x0; r.lo[1] = y0;
        r.scale[0] = (x1 > x0 && isfinite(x1 - x0)) ? ND_NB / (x1 - x0) : 0.0; r.scale[1] = (y1 > y0 && isfinite(y1 - y0)) ? ND_NB / (y1 - y0) : 0.0;
        *rk = r;
    }
}
__global__ void nd_bhist_kernel(int n, const double* __restrict__ sxy, const nd_rk* __restrict__ rk, int* __restrict__ hist)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const nd_rk r = *rk;
    atomicAdd(&hist[nd_bucket(sxy[2 * (size_t)v], r.lo[0], r.scale[0])], 1);
    atomicAdd(&hist[ND_NB + nd_bucket(sxy[2 * (size_t)v + 1], r.lo[1], r.scale[1])], 1);
}
__global__ __launch_bounds__(ND_NB) void nd_bscan_kernel(int* __restrict__ hist)      // exclusive scans of the two histograms, in place; copies as fill cursors
{
    __shared__ int slot[2][ND_NB / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int v = hist[a * ND_NB + threadIdx.x];
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) slot[a][w] = inc;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < w; ++k) base += slot[a][k];
        hist[a * ND_NB + threadIdx.x] = base + inc - v; hist[(2 + a) * ND_NB + threadIdx.x] = base + inc - v;
    }
}
__global__ void nd_bfill_kernel(int n, const double* __restrict__ sxy, const nd_rk* __restrict__ rk, int* __restrict__ hist, int* __restrict__ memx, int* __restrict__ memy)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const nd_rk r = *rk;
    memx[atomicAdd(&hist[2 * ND_NB + nd_bucket(sxy[2 * (size_t)v], r.lo[0], r.scale[0])], 1)] = v;
    memy[atomicAdd(&hist[3 * ND_NB + nd_bucket(sxy[2 * (size_t)v + 1], r.lo[1], r.scale[1])], 1)] = v;
}
__global__ void nd_brank_kernel(int n, const double* __restrict__ sxy, const nd_rk* __restrict__ rk, const int* __restrict__ hist, const int* __restrict__ memx, const int* __restrict__ memy,
                                int* __restrict__ rank_x, int* __restrict__ rank_y, unsigned* __restrict__ rank_xy)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const nd_rk r = *rk;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const double key = sxy[2 * (size_t)v + a];
        const int b = nd_bucket(key, r.lo[a], r.scale[a]);
        const int s = hist[a * ND_NB + b], e = hist[(2 + a) * ND_NB + b];      // (the fill cursors ended at the bucket ends)
        const int* __restrict__ mem = a ? memy : memx;
        int before = 0;
        // (coordinate, index) order, made TOTAL: a NaN coordinate sorts behind every number and ties with other NaNs by index, so the
        // ranks are a permutation whatever the caller's dead-reckoning rows hold (with `ku < key || ku == key` alone both tests fail on
        // a NaN, two nodes share a rank, and the level kernels' partitions leave their slices)
        const bool nk = key != key;
        for (int q = s; q < e; ++q) {
            const int u = mem[q]; const double ku = sxy[2 * (size_t)u + a];
            const bool nu = ku != ku;
            const bool less = nu ? false : (nk ? true : ku < key), same = (nu && nk) || ku == key;
            before += (less || (same && u < v)) ? 1 : 0;
        }
        (a ? rank_y : rank_x)[v] = s + before;
    }
    rank_xy[v] = (unsigned)rank_x[v] | ((unsigned)rank_y[v] << 16);      // (n <= 65536)
}

// ---- adjacency
__global__ void nd_deg_kernel(int ne, const int2* __restrict__ e, int* __restrict__ deg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ne) return;
    const int2 p = e[i];
    if (p.x == p.y) return;
    atomicAdd(&deg[p.x], 1); atomicAdd(&deg[p.y], 1);
}
__global__ __launch_bounds__(ND_T) void nd_scan_kernel(int n, const int* __restrict__ deg, int* __restrict__ ptr, int* __restrict__ cur, int* __restrict__ perm, int* __restrict__ setid)
{
    __shared__ int slot[ND_T / 64];
    __shared__ int run;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    for (int b0 = 0; b0 < n; b0 += ND_T) {
        const int i = b0 + threadIdx.x;
        const int v = i < n ? deg[i] : 0;
        int inc = v;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) slot[w] = inc;
        __syncthreads();
        int base = run;
        for (int k = 0; k < w; ++k) base += slot[k];
        if (i < n) { ptr[i] = base + inc - v; cur[i] = base + inc - v; perm[i] = i; setid[i] = 1; }
        __syncthreads();
        if (threadIdx.x == ND_T - 1) run = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) ptr[n] = run;
}
__global__ void nd_fill_kernel(int ne, const int2* __restrict__ e, int* __restrict__ cur, int* __restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ne) return;
    const int2 p = e[i];
    if (p.x == p.y) return;
    idx[atomicAdd(&cur[p.x], 1)] = p.y; idx[atomicAdd(&cur[p.y], 1)] = p.x;
}

// ---- one level of the recursion
struct nd_args {
    int n, leaf, both_axes;
    const double* sxy; const int* rank_x; const int* rank_y; const int* adj_ptr; const int* adj_idx;
    int* perm0; int* perm1; int* setid; unsigned char* cut0; const unsigned* rank_xy; int* order; pg_nd_set* sets; int* fail;
};

// A workgroup walks its set in sixteen contiguous chunks, one per wavefront, 256 nodes at a time: the four node ids of a lane are
// requested together and what hangs on them (rank, coordinates, cut flag) together again -- a pass is two round trips per 4 096 nodes,
// not one per 1 024 (the first version: 60 us per launch on average, 250 us for the 23 k nodes of level 0).
#define ND_U 4
struct nd_chunk { int lo, hi; };
__device__ inline nd_chunk nd_my_chunk(int m)
{
    const int w = threadIdx.x >> 6, per = ((m + (ND_T / 64) - 1) / (ND_T / 64) + 63) & ~63;      // whole rows of 64 per wavefront: lanes stay aligned with positions
    nd_chunk c; c.lo = min(m, w * per); c.hi = min(m, c.lo + per);
    return c;
}

// h2-th smallest rank of the set along BOTH axes (ranks are distinct and below 65536): two histogram passes, each over both axes at once
__device__ inline void nd_select2(const int* __restrict__ P, int m, const int* __restrict__ RX, const int* __restrict__ RY, int h2, nd_lds& S, int* px, int* py)
{
    const nd_chunk ch = nd_my_chunk(m);
    const int lane = threadIdx.x & 63;
    int hx = 0, hy = 0, kx = h2, ky = h2;
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        if (threadIdx.x < 512) S.hist2[threadIdx.x] = 0u;
        __syncthreads();
        for (int b0 = ch.lo; b0 < ch.hi; b0 += 64 * ND_U) {
            int v[ND_U], rx[ND_U], ry[ND_U];
#pragma unroll
            for (int u = 0; u < ND_U; ++u) { const int i = b0 + 64 * u + lane; v[u] = i < ch.hi ? P[i] : -1; }
#pragma unroll
            for (int u = 0; u < ND_U; ++u) { rx[u] = v[u] >= 0 ? RX[v[u]] : -1; ry[u] = v[u] >= 0 ? RY[v[u]] : -1; }
#pragma unroll
            for (int u = 0; u < ND_U; ++u) if (v[u] >= 0) {
                if (pass == 0) { atomicAdd(&S.hist2[rx[u] >> 8], 1u); atomicAdd(&S.hist2[256 + (ry[u] >> 8)], 1u); }
                else { if ((rx[u] >> 8) == hx) atomicAdd(&S.hist2[rx[u] & 255], 1u); if ((ry[u] >> 8) == hy) atomicAdd(&S.hist2[256 + (ry[u] & 255)], 1u); }
            }
        }
        __syncthreads();
        nd_find_bins(S, kx, ky);
        if (pass == 0) { hx = S.sel[0]; kx = S.sel[1]; hy = S.sel[2]; ky = S.sel[3]; }
    }
    *px = (hx << 8) | S.sel[0]; *py = (hy << 8) | S.sel[2];
}

__global__ __launch_bounds__(ND_T) void nd_level_kernel(int L, nd_args A)
{
    __shared__ nd_lds S;
    const int h = (1 << L) + blockIdx.x;
    pg_nd_set d = A.sets[h];
    const int m = d.size;
    if (m <= 0) return;
    const int* __restrict__ P = ((L & 1) ? A.perm1 : A.perm0) + d.lo;
    int* __restrict__ Pn = ((L & 1) ? A.perm0 : A.perm1) + d.lo;
    if (m <= A.leaf) {                                                       // a leaf: its nodes in index order
        for (int i = threadIdx.x; i < m; i += ND_T) A.order[d.out + i] = P[i];
        if (threadIdx.x == 0) { d.kind = 1; A.sets[h] = d; }
        return;
    }
    const nd_chunk ch = nd_my_chunk(m);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // bounding box
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int b0 = ch.lo; b0 < ch.hi; b0 += 64 * ND_U) {
        int v[ND_U]; double2 p[ND_U];
#pragma unroll
        for (int u = 0; u < ND_U; ++u) { const int i = b0 + 64 * u + lane; v[u] = i < ch.hi ? P[i] : -1; }
#pragma unroll
        for (int u = 0; u < ND_U; ++u) p[u] = v[u] >= 0 ? reinterpret_cast<const double2*>(A.sxy)[v[u]] : make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < ND_U; ++u) if (v[u] >= 0) { x0 = fmin(x0, p[u].x); x1 = fmax(x1, p[u].x); y0 = fmin(y0, p[u].y); y1 = fmax(y1, p[u].y); }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { x0 = fmin(x0, nd_shfl_d(x0, o)); x1 = fmax(x1, nd_shfl_d(x1, o)); y0 = fmin(y0, nd_shfl_d(y0, o)); y1 = fmax(y1, nd_shfl_d(y1, o)); }
    if (lane == 0) { S.red[0][wv] = x0; S.red[1][wv] = x1; S.red[2][wv] = y0; S.red[3][wv] = y1; }
    __syncthreads();
    for (int k = 0; k < ND_T / 64; ++k) { x0 = fmin(x0, S.red[0][k]); x1 = fmax(x1, S.red[1][k]); y0 = fmin(y0, S.red[2][k]); y1 = fmax(y1, S.red[3][k]); }
    const bool byx = (x1 - x0) >= (y1 - y0);
    const bool two = m >= A.both_axes;
    const int h2 = m / 2;
    // the median rank along both axes, then both candidates counted in ONE walk over the adjacency: bit 0 of a node's flag says it is in
    // the separator of the cut along x, bit 1 along y
    int pvx, pvy;
    nd_select2(P, m, A.rank_x, A.rank_y, h2, S, &pvx, &pvy);
    int mx = 0, my = 0;
    for (int b0 = ch.lo; b0 < ch.hi; b0 += 64 * ND_U) {
        int v[ND_U], rx[ND_U], ry[ND_U], q0[ND_U], q1[ND_U];
#pragma unroll
        for (int u = 0; u < ND_U; ++u) { const int i = b0 + 64 * u + lane; v[u] = i < ch.hi ? P[i] : -1; }
#pragma unroll
        for (int u = 0; u < ND_U; ++u) { const bool ok = v[u] >= 0; rx[u] = ok ? A.rank_x[v[u]] : 0x7fffffff; ry[u] = ok ? A.rank_y[v[u]] : 0x7fffffff; q0[u] = ok ? A.adj_ptr[v[u]] : 0; q1[u] = ok ? A.adj_ptr[v[u] + 1] : 0; }
#pragma unroll
        for (int u = 0; u < ND_U; ++u) if (v[u] >= 0) {
            const bool lx = rx[u] < pvx, ly = ry[u] < pvy;
            unsigned f = 0;
            if (lx || ly)
                for (int q = q0[u]; q < q1[u] && f != ((lx ? 1u : 0u) | (ly ? 2u : 0u)); ++q) {
                    const int w2 = A.adj_idx[q];
                    if (A.setid[w2] != h) continue;
                    if (lx && A.rank_x[w2] >= pvx) f |= 1u;
                    if (ly && A.rank_y[w2] >= pvy) f |= 2u;
                }
            A.cut0[v[u]] = (unsigned char)f; mx += f & 1u; my += (f >> 1) & 1u;
        }
    }
    const int cx_ = nd_block_sum(mx, &S.ired[0][0], 0), cy_ = nd_block_sum(my, &S.ired[0][0], 1);
    int pivot[2], cnt[2];                                                   // candidate 0: the longer extent
    pivot[0] = byx ? pvx : pvy; cnt[0] = byx ? cx_ : cy_; pivot[1] = byx ? pvy : pvx; cnt[1] = byx ? cy_ : cx_;
    const int win = (two && cnt[1] < cnt[0]) ? 1 : 0;                        // ties: the longer extent
    const bool wbx = win == 0 ? byx : !byx;
    const int* __restrict__ R = wbx ? A.rank_x : A.rank_y;
    const unsigned char* cut = A.cut0; const unsigned cbit = wbx ? 1u : 2u;
    const int pv = pivot[win], nS = cnt[win], nA = h2 - nS, nB = m - h2;
    if (nA <= 0 || nB <= 0) {                                               // degenerate cut: index order
        for (int i = threadIdx.x; i < m; i += ND_T) A.order[d.out + i] = P[i];
        if (threadIdx.x == 0) { d.kind = 1; A.sets[h] = d; }
        return;
    }
    // stable three-way partition in index order.  Pass 1: what each wavefront's chunk holds of A, B, S; pass 2: it writes them behind
    // the earlier chunks' (the class of a node is recomputed: rank and cut flag are one more pair of loads, cheaper than parking 23 k classes)
    int cA = 0, cB = 0, cS = 0;
    for (int b0 = ch.lo; b0 < ch.hi; b0 += 64 * ND_U) {
        int v[ND_U], r[ND_U]; unsigned char f[ND_U];
#pragma unroll
        for (int u = 0; u < ND_U; ++u) { const int i = b0 + 64 * u + lane; v[u] = i < ch.hi ? P[i] : -1; }
#pragma unroll
        for (int u = 0; u < ND_U; ++u) { r[u] = v[u] >= 0 ? R[v[u]] : 0; f[u] = v[u] >= 0 ? (unsigned char)(cut[v[u]] & cbit) : 0; }
#pragma unroll
        for (int u = 0; u < ND_U; ++u) if (v[u] >= 0) { const bool lower = r[u] < pv; cS += lower && f[u]; cA += lower && !f[u]; cB += !lower; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { cA += __shfl_xor(cA, o, 64); cB += __shfl_xor(cB, o, 64); cS += __shfl_xor(cS, o, 64); }
    __syncthreads();                                                         // (every cut flag of the set has been read or written: relabelling may start after the next barrier)
    if (lane == 0) { S.ired[0][wv] = cA; S.ired[1][wv] = cB; S.ired[2][wv] = cS; }
    __syncthreads();
    int bA = 0, bB = 0, bS = 0;
    for (int k = 0; k < wv; ++k) { bA += S.ired[0][k]; bB += S.ired[1][k]; bS += S.ired[2][k]; }
    for (int b0 = ch.lo; b0 < ch.hi; b0 += 64 * ND_U) {
        int v[ND_U], r[ND_U]; unsigned char f[ND_U];
#pragma unroll
        for (int u = 0; u < ND_U; ++u) { const int i = b0 + 64 * u + lane; v[u] = i < ch.hi ? P[i] : -1; }
#pragma unroll
        for (int u = 0; u < ND_U; ++u) { r[u] = v[u] >= 0 ? R[v[u]] : 0; f[u] = v[u] >= 0 ? (unsigned char)(cut[v[u]] & cbit) : 0; }
#pragma unroll
        for (int u = 0; u < ND_U; ++u) {                                     // rows of 64 in position order
            const bool ok = v[u] >= 0, lower = ok && r[u] < pv;
            const bool isS = lower && f[u], isA = lower && !f[u], isB = ok && !lower;
            const unsigned long long mA = __ballot(isA), mB = __ballot(isB), mS = __ballot(isS);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (isA) { Pn[bA + __popcll(mA & below)] = v[u]; A.setid[v[u]] = 2 * h; }
            if (isB) { Pn[nA + bB + __popcll(mB & below)] = v[u]; A.setid[v[u]] = 2 * h + 1; }
            if (isS) { A.order[d.out + nA + nB + bS + __popcll(mS & below)] = v[u]; A.setid[v[u]] = 0; }
            bA += __popcll(mA); bB += __popcll(mB); bS += __popcll(mS);
        }
    }
    if (threadIdx.x == 0) {
        d.kind = 2; d.nA = nA; d.nB = nB; A.sets[h] = d;
        pg_nd_set a; a.lo = d.lo; a.size = nA; a.out = d.out; a.kind = 0; a.nA = 0; a.nB = 0; A.sets[2 * h] = a;
        pg_nd_set b; b.lo = d.lo + nA; b.size = nB; b.out = d.out + nA; b.kind = 0; b.nA = 0; b.nB = 0; A.sets[2 * h + 1] = b;
    }
}

// The same step for sets of at most 4 G nodes, G = 1024, 256 or 64 threads per set (a "group": a whole workgroup, or one of the four
// wavefronts of a 256-thread workgroup).  A thread's (at most four) nodes, their ranks and adjacency ranges are loaded ONCE and stay in
// registers through all phases, and a node's first four neighbours are fetched as a batch: a level of small sets was 24 dependent round
// trips and sixteen workgroup barriers, 40 - 45 us whatever the size of its sets.  In-kernel stamps of the 1024-thread form at 360 nodes:
// 2.5 us loads, 2.8 box, 6.4 medians, 12 neighbours (the tail of a high-degree node, one dependent load at a time), 4 sums, 1.9 partition --
// so the groups shrink with the sets (a wavefront needs no barrier at all) and the tail is walked four neighbours at a time.
#define ND_NBR 4
template <int WPG> struct nd_gl { double red[4][WPG]; int ired[3][WPG]; unsigned hist2[512]; int sel[4]; int wtot[2][4]; };

template <int G> __device__ inline void nd_gsync()
{
    if (G == 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}

// bins of rank kx / ky in the two 256-bin histograms of the group: S.sel = { bin x, rest x, bin y, rest y }
template <int G, int WPG> __device__ inline void nd_find_bins_g(nd_gl<WPG>& S, int tg, int kx, int ky)
{
    constexpr int T = G < 256 ? G : 256, PER = 256 / T;          // threads that search, bins per thread and axis
    const int lane = tg & 63, w = tg >> 6;
    int loc[2][PER], tot[2] = { 0, 0 };
    if (tg < T) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < PER; ++j) { loc[a][j] = (int)S.hist2[a * 256 + tg * PER + j]; tot[a] += loc[a][j]; }
    }
    int inc[2] = { tot[0], tot[1] };
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(inc[0], o, 64), y = __shfl_up(inc[1], o, 64); if (lane >= o) { inc[0] += x; inc[1] += y; } }
    if (T > 64) {
        if (tg < T && lane == 63) { S.wtot[0][w] = inc[0]; S.wtot[1][w] = inc[1]; }
        nd_gsync<G>();
    }
    if (tg < T) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            int base = 0;
            if (T > 64) for (int k = 0; k < w; ++k) base += S.wtot[a][k];
            int excl = base + inc[a] - tot[a];
            const int k = a ? ky : kx;
#pragma unroll
            for (int j = 0; j < PER; ++j) { if (excl <= k && k < excl + loc[a][j]) { S.sel[2 * a] = tg * PER + j; S.sel[2 * a + 1] = k - excl; } excl += loc[a][j]; }
        }
    }
    nd_gsync<G>();
}

template <int G>
__global__ __launch_bounds__(G < 256 ? 256 : G) void nd_level_group_kernel(int L, nd_args A)
{
    constexpr int BT = G < 256 ? 256 : G, GPB = BT / G, WPG = G / 64;
    __shared__ nd_gl<WPG> S_all[GPB];
    const int grp = threadIdx.x / G, tg = threadIdx.x % G, lane = tg & 63, wv = tg >> 6;
    const int kset = blockIdx.x * GPB + grp;
    if (kset >= (1 << L)) return;                                           // (whole groups leave together)
    nd_gl<WPG>& S = S_all[grp];
    const int h = (1 << L) + kset;
    pg_nd_set d = A.sets[h];
    const int m = d.size;
    if (m <= 0) return;
    const int* __restrict__ P = ((L & 1) ? A.perm1 : A.perm0) + d.lo;
    int* __restrict__ Pn = ((L & 1) ? A.perm0 : A.perm1) + d.lo;
    if (m <= A.leaf) {
        for (int i = tg; i < m; i += G) A.order[d.out + i] = P[i];
        if (tg == 0) { d.kind = 1; A.sets[h] = d; }
        return;
    }
    if (m > G * ND_U) { if (tg == 0) *A.fail = 1; return; }                  // (the host sizes the launches so that this cannot happen: never silent)
    const int per = ((m + WPG - 1) / WPG + 63) & ~63;                        // whole rows of 64 per wavefront: lanes stay aligned with positions
    const int clo = min(m, wv * per), chi = min(m, clo + per);
    int v[ND_U], rx[ND_U], ry[ND_U], q0[ND_U], q1[ND_U];
    double2 p[ND_U];
#pragma unroll
    for (int u = 0; u < ND_U; ++u) { const int i = clo + 64 * u + lane; v[u] = i < chi ? P[i] : -1; }
#pragma unroll
    for (int u = 0; u < ND_U; ++u) {
        const bool ok = v[u] >= 0;
        const unsigned r = ok ? A.rank_xy[v[u]] : 0u;
        rx[u] = (int)(r & 0xffffu); ry[u] = (int)(r >> 16);
        q0[u] = ok ? A.adj_ptr[v[u]] : 0; q1[u] = ok ? A.adj_ptr[v[u] + 1] : 0;
        p[u] = ok ? reinterpret_cast<const double2*>(A.sxy)[v[u]] : make_double2(0.0, 0.0);
    }
    // the neighbours' ids are on their way while the box and the medians are worked out
    int nb[ND_U][ND_NBR];
#pragma unroll
    for (int u = 0; u < ND_U; ++u)
#pragma unroll
        for (int j = 0; j < ND_NBR; ++j) nb[u][j] = (v[u] >= 0 && q0[u] + j < q1[u]) ? A.adj_idx[q0[u] + j] : -1;
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
#pragma unroll
    for (int u = 0; u < ND_U; ++u) if (v[u] >= 0) { x0 = fmin(x0, p[u].x); x1 = fmax(x1, p[u].x); y0 = fmin(y0, p[u].y); y1 = fmax(y1, p[u].y); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { x0 = fmin(x0, nd_shfl_d(x0, o)); x1 = fmax(x1, nd_shfl_d(x1, o)); y0 = fmin(y0, nd_shfl_d(y0, o)); y1 = fmax(y1, nd_shfl_d(y1, o)); }
    for (int i = tg; i < 512; i += G) S.hist2[i] = 0u;
    if (WPG > 1) {
        if (lane == 0) { S.red[0][wv] = x0; S.red[1][wv] = x1; S.red[2][wv] = y0; S.red[3][wv] = y1; }
        nd_gsync<G>();
        for (int k = 0; k < WPG; ++k) { x0 = fmin(x0, S.red[0][k]); x1 = fmax(x1, S.red[1][k]); y0 = fmin(y0, S.red[2][k]); y1 = fmax(y1, S.red[3][k]); }
    } else nd_gsync<G>();
    const bool byx = (x1 - x0) >= (y1 - y0);
    const bool two = m >= A.both_axes;
    const int h2 = m / 2;
    int hx = 0, hy = 0, kx = h2, ky = h2;
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int u = 0; u < ND_U; ++u) if (v[u] >= 0) {
            if (pass == 0) { atomicAdd(&S.hist2[rx[u] >> 8], 1u); atomicAdd(&S.hist2[256 + (ry[u] >> 8)], 1u); }
            else { if ((rx[u] >> 8) == hx) atomicAdd(&S.hist2[rx[u] & 255], 1u); if ((ry[u] >> 8) == hy) atomicAdd(&S.hist2[256 + (ry[u] & 255)], 1u); }
        }
        nd_gsync<G>();
        nd_find_bins_g<G, WPG>(S, tg, kx, ky);
        const int sx = S.sel[0], rkx = S.sel[1], sy = S.sel[2], rky = S.sel[3];
        nd_gsync<G>();
        if (pass == 0) { hx = sx; kx = rkx; hy = sy; ky = rky; for (int i = tg; i < 512; i += G) S.hist2[i] = 0u; nd_gsync<G>(); }
        else { hx = (hx << 8) | sx; hy = (hy << 8) | sy; }
    }
    const int pvx = hx, pvy = hy;
    // both candidates in one look at the neighbours, four at a time
    unsigned fl[ND_U]; int mxy = 0;
#pragma unroll
    for (int u = 0; u < ND_U; ++u) fl[u] = 0;
    {
        int sid[ND_U][ND_NBR]; unsigned rk[ND_U][ND_NBR];
#pragma unroll
        for (int u = 0; u < ND_U; ++u)
#pragma unroll
            for (int j = 0; j < ND_NBR; ++j) { const bool ok = nb[u][j] >= 0; sid[u][j] = ok ? A.setid[nb[u][j]] : -1; rk[u][j] = ok ? A.rank_xy[nb[u][j]] : 0u; }
#pragma unroll
        for (int u = 0; u < ND_U; ++u) {
            if (v[u] < 0) continue;
            const bool lx = rx[u] < pvx, ly = ry[u] < pvy;
#pragma unroll
            for (int j = 0; j < ND_NBR; ++j) if (sid[u][j] == h) {
                if (lx && (int)(rk[u][j] & 0xffffu) >= pvx) fl[u] |= 1u;
                if (ly && (int)(rk[u][j] >> 16) >= pvy) fl[u] |= 2u;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < ND_U; ++u) {                                         // the tail of a high-degree node, four neighbours per round trip pair
        if (v[u] < 0) continue;
        const bool lx = rx[u] < pvx, ly = ry[u] < pvy;
        for (int q = q0[u] + ND_NBR; q < q1[u]; q += ND_NBR) {
            int w2[ND_NBR], s2[ND_NBR]; unsigned r2[ND_NBR];
#pragma unroll
            for (int j = 0; j < ND_NBR; ++j) w2[j] = q + j < q1[u] ? A.adj_idx[q + j] : -1;
#pragma unroll
            for (int j = 0; j < ND_NBR; ++j) { s2[j] = w2[j] >= 0 ? A.setid[w2[j]] : -1; r2[j] = w2[j] >= 0 ? A.rank_xy[w2[j]] : 0u; }
#pragma unroll
            for (int j = 0; j < ND_NBR; ++j) if (s2[j] == h) {
                if (lx && (int)(r2[j] & 0xffffu) >= pvx) fl[u] |= 1u;
                if (ly && (int)(r2[j] >> 16) >= pvy) fl[u] |= 2u;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < ND_U; ++u) mxy += (int)(fl[u] & 1u) + ((int)((fl[u] >> 1) & 1u) << 16);      // (both counts in one word: a set holds at most 4 096 nodes)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mxy += __shfl_xor(mxy, o, 64);
    if (WPG > 1) {
        if (lane == 0) S.ired[0][wv] = mxy;
        nd_gsync<G>();
        mxy = 0;
        for (int k = 0; k < WPG; ++k) mxy += S.ired[0][k];
        nd_gsync<G>();
    }
    const int cx_ = mxy & 0xffff, cy_ = mxy >> 16;
    const int c0 = byx ? cx_ : cy_, c1 = byx ? cy_ : cx_;
    const int win = (two && c1 < c0) ? 1 : 0;
    const bool wbx = win == 0 ? byx : !byx;
    const int pv = wbx ? pvx : pvy, nS = win == 0 ? c0 : c1, nA = h2 - nS, nB = m - h2;
    if (nA <= 0 || nB <= 0) {
        for (int i = tg; i < m; i += G) A.order[d.out + i] = P[i];
        if (tg == 0) { d.kind = 1; A.sets[h] = d; }
        return;
    }
    bool isA[ND_U], isB[ND_U], isS[ND_U];
    int cA = 0, cB = 0, cS = 0;
#pragma unroll
    for (int u = 0; u < ND_U; ++u) {
        const bool ok = v[u] >= 0, lower = ok && (wbx ? rx[u] : ry[u]) < pv, cutf = (fl[u] & (wbx ? 1u : 2u)) != 0;
        isS[u] = lower && cutf; isA[u] = lower && !cutf; isB[u] = ok && !lower;
        cA += isA[u]; cB += isB[u]; cS += isS[u];
    }
    int bA = 0, bB = 0, bS = 0;
    if (WPG > 1) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { cA += __shfl_xor(cA, o, 64); cB += __shfl_xor(cB, o, 64); cS += __shfl_xor(cS, o, 64); }
        if (lane == 0) { S.ired[0][wv] = cA; S.ired[1][wv] = cB; S.ired[2][wv] = cS; }
        nd_gsync<G>();                                                       // (every look at setid is over as well)
        for (int k = 0; k < wv; ++k) { bA += S.ired[0][k]; bB += S.ired[1][k]; bS += S.ired[2][k]; }
    } else nd_gsync<G>();
#pragma unroll
    for (int u = 0; u < ND_U; ++u) {
        const unsigned long long mA = __ballot(isA[u]), mB = __ballot(isB[u]), mS = __ballot(isS[u]);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (isA[u]) { Pn[bA + __popcll(mA & below)] = v[u]; A.setid[v[u]] = 2 * h; }
        if (isB[u]) { Pn[nA + bB + __popcll(mB & below)] = v[u]; A.setid[v[u]] = 2 * h + 1; }
        if (isS[u]) { A.order[d.out + nA + nB + bS + __popcll(mS & below)] = v[u]; A.setid[v[u]] = 0; }
        bA += __popcll(mA); bB += __popcll(mB); bS += __popcll(mS);
    }
    if (tg == 0) {
        d.kind = 2; d.nA = nA; d.nB = nB; A.sets[h] = d;
        pg_nd_set a; a.lo = d.lo; a.size = nA; a.out = d.out; a.kind = 0; a.nA = 0; a.nB = 0; A.sets[2 * h] = a;
        pg_nd_set b; b.lo = d.lo + nA; b.size = nB; b.out = d.out + nA; b.kind = 0; b.nA = 0; b.nB = 0; A.sets[2 * h + 1] = b;
    }
}

// ---- the levels whose sets do not fit one workgroup's registers (more than 4 096 nodes: the first three levels of C3).  One workgroup
// walking such a set is bound by what ONE compute unit loads (0.8 MB a pass, six passes: 167 us for the 23 k nodes of level 0), so a set
// is cut into slices of 4 096 nodes, a workgroup each, and the four steps that need the whole set's answer are four launches:
//   A  box (atomic min / max on order-preserving keys), histograms of the high rank bytes      B  histograms of the low rank bytes
//   C  medians, both cut candidates counted, a slice's class counts for either candidate       D  the choice, the stable partition
#define ND_SL (1024 * ND_U)
#define ND_WMAX 16                                   // slices per set: sets of up to 65 536 nodes
struct nd_big { unsigned long long bb[4]; unsigned hist_hi[2][256], hist_lo[2][256]; int cnt[2]; int cls[ND_WMAX][2][3]; int pad[2]; };

__device__ inline unsigned long long nd_key(double v) { const unsigned long long b = (unsigned long long)__double_as_longlong(v); return (b >> 63) ? ~b : (b | 0x8000000000000000ull); }
__device__ inline double nd_unkey(unsigned long long k) { const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k; return __longlong_as_double((long long)b); }

struct nd_slice { int h, m, lo, n; const int* P; };   // set, its size, the slice's first position and node count, the set's node list
__device__ inline bool nd_get_slice(int L, int W, const nd_args& A, nd_slice* s, pg_nd_set* d)
{
    const int kset = blockIdx.x / W, j = blockIdx.x % W;
    s->h = (1 << L) + kset;
    *d = A.sets[s->h];
    s->m = d->size;
    s->lo = j * ND_SL;
    s->n = min(ND_SL, s->m - s->lo);
    s->P = ((L & 1) ? A.perm1 : A.perm0) + d->lo;
    return s->m > A.leaf && s->n > 0;                 // (leaves of a big level -- a tiny DSSS_PG_LEAF aside there are none -- are handled by phase D's slice 0)
}

__global__ __launch_bounds__(1024) void nd_bigA_kernel(int L, int W, nd_args A, nd_big* __restrict__ G)
{
    __shared__ unsigned hist[512];
    __shared__ double red[4][16];
    nd_slice sl; pg_nd_set d;
    if (!nd_get_slice(L, W, A, &sl, &d)) return;
    nd_big& g = G[blockIdx.x / W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 512) hist[threadIdx.x] = 0u;
    int v[ND_U]; unsigned r[ND_U]; double2 p[ND_U];
#pragma unroll
    for (int u = 0; u < ND_U; ++u) { const int i = 256 * wv + 64 * u + lane; v[u] = i < sl.n ? sl.P[sl.lo + i] : -1; }
#pragma unroll
    for (int u = 0; u < ND_U; ++u) { r[u] = v[u] >= 0 ? A.rank_xy[v[u]] : 0u; p[u] = v[u] >= 0 ? reinterpret_cast<const double2*>(A.sxy)[v[u]] : make_double2(0.0, 0.0); }
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
#pragma unroll
    for (int u = 0; u < ND_U; ++u) if (v[u] >= 0) { x0 = fmin(x0, p[u].x); x1 = fmax(x1, p[u].x); y0 = fmin(y0, p[u].y); y1 = fmax(y1, p[u].y); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { x0 = fmin(x0, nd_shfl_d(x0, o)); x1 = fmax(x1, nd_shfl_d(x1, o)); y0 = fmin(y0, nd_shfl_d(y0, o)); y1 = fmax(y1, nd_shfl_d(y1, o)); }
    if (lane == 0) { red[0][wv] = x0; red[1][wv] = x1; red[2][wv] = y0; red[3][wv] = y1; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ND_U; ++u) if (v[u] >= 0) { atomicAdd(&hist[(r[u] & 0xffffu) >> 8], 1u); atomicAdd(&hist[256 + (r[u] >> 24)], 1u); }
    if (threadIdx.x == 0) {
        for (int k = 0; k < 16; ++k) { x0 = fmin(x0, red[0][k]); x1 = fmax(x1, red[1][k]); y0 = fmin(y0, red[2][k]); y1 = fmax(y1, red[3][k]); }
        atomicMin(&g.bb[0], nd_key(x0)); atomicMax(&g.bb[1], nd_key(x1)); atomicMin(&g.bb[2], nd_key(y0)); atomicMax(&g.bb[3], nd_key(y1));
    }
    __syncthreads();
    if (threadIdx.x < 512 && hist[threadIdx.x]) atomicAdd(&g.hist_hi[threadIdx.x >> 8][threadIdx.x & 255], hist[threadIdx.x]);
}

// the set's two high bins (and what is left of h2 inside them) from its global histograms, by every workgroup for itself
__device__ inline void nd_big_hi(const nd_big& g, nd_gl<16>& S, int h2, int* hx, int* kx, int* hy, int* ky)
{
    if (threadIdx.x < 512) S.hist2[threadIdx.x] = g.hist_hi[threadIdx.x >> 8][threadIdx.x & 255];
    __syncthreads();
    nd_find_bins_g<1024, 16>(S, threadIdx.x, h2, h2);
    *hx = S.sel[0]; *kx = S.sel[1]; *hy = S.sel[2]; *ky = S.sel[3];
    __syncthreads();
}

__global__ __launch_bounds__(1024) void nd_bigB_kernel(int L, int W, nd_args A, nd_big* __restrict__ G)
{
    __shared__ nd_gl<16> S;
    __shared__ unsigned hist[512];
    nd_slice sl; pg_nd_set d;
    if (!nd_get_slice(L, W, A, &sl, &d)) return;
    nd_big& g = G[blockIdx.x / W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int v[ND_U]; unsigned r[ND_U];
#pragma unroll
    for (int u = 0; u < ND_U; ++u) { const int i = 256 * wv + 64 * u + lane; v[u] = i < sl.n ? sl.P[sl.lo + i] : -1; }
#pragma unroll
    for (int u = 0; u < ND_U; ++u) r[u] = v[u] >= 0 ? A.rank_xy[v[u]] : 0u;
    if (threadIdx.x < 512) hist[threadIdx.x] = 0u;
    int hx, kx, hy, ky;
    nd_big_hi(g, S, sl.m / 2, &hx, &kx, &hy, &ky);
#pragma unroll
    for (int u = 0; u < ND_U; ++u) if (v[u] >= 0) {
        const int rx = (int)(r[u] & 0xffffu), ry = (int)(r[u] >> 16);
        if ((rx >> 8) == hx) atomicAdd(&hist[rx & 255], 1u);
        if ((ry >> 8) == hy) atomicAdd(&hist[256 + (ry & 255)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 512 && hist[threadIdx.x]) atomicAdd(&g.hist_lo[threadIdx.x >> 8][threadIdx.x & 255], hist[threadIdx.x]);
}

__device__ inline void nd_big_pivots(const nd_big& g, nd_gl<16>& S, int h2, int* pvx, int* pvy)
{
    int hx, kx, hy, ky;
    nd_big_hi(g, S, h2, &hx, &kx, &hy, &ky);
    if (threadIdx.x < 512) S.hist2[threadIdx.x] = g.hist_lo[threadIdx.x >> 8][threadIdx.x & 255];
    __syncthreads();
    nd_find_bins_g<1024, 16>(S, threadIdx.x, kx, ky);
    *pvx = (hx << 8) | S.sel[0]; *pvy = (hy << 8) | S.sel[2];
    __syncthreads();
}

__global__ __launch_bounds__(1024) void nd_bigC_kernel(int L, int W, nd_args A, nd_big* __restrict__ G)
{
    __shared__ nd_gl<16> S;
    nd_slice sl; pg_nd_set d;
    if (!nd_get_slice(L, W, A, &sl, &d)) return;
    nd_big& g = G[blockIdx.x / W];
    const int h = sl.h, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = blockIdx.x % W;
    int v[ND_U], rx[ND_U], ry[ND_U], q0[ND_U], q1[ND_U];
#pragma unroll
    for (int u = 0; u < ND_U; ++u) { const int i = 256 * wv + 64 * u + lane; v[u] = i < sl.n ? sl.P[sl.lo + i] : -1; }
#pragma unroll
    for (int u = 0; u < ND_U; ++u) {
        const bool ok = v[u] >= 0;
        const unsigned r = ok ? A.rank_xy[v[u]] : 0u;
        rx[u] = (int)(r & 0xffffu); ry[u] = (int)(r >> 16);
        q0[u] = ok ? A.adj_ptr[v[u]] : 0; q1[u] = ok ? A.adj_ptr[v[u] + 1] : 0;
    }
    int pvx, pvy;
    nd_big_pivots(g, S, sl.m / 2, &pvx, &pvy);
    int c[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };      // [x-cut, y-cut][A, B, S]
#pragma unroll
    for (int u = 0; u < ND_U; ++u) {
        if (v[u] < 0) continue;
        const bool lx = rx[u] < pvx, ly = ry[u] < pvy;
        unsigned fl = 0;
        for (int q = q0[u]; q < q1[u]; q += ND_NBR) {
            int w2[ND_NBR], s2[ND_NBR]; unsigned r2[ND_NBR];
#pragma unroll
            for (int t = 0; t < ND_NBR; ++t) w2[t] = q + t < q1[u] ? A.adj_idx[q + t] : -1;
#pragma unroll
            for (int t = 0; t < ND_NBR; ++t) { s2[t] = w2[t] >= 0 ? A.setid[w2[t]] : -1; r2[t] = w2[t] >= 0 ? A.rank_xy[w2[t]] : 0u; }
#pragma unroll
            for (int t = 0; t < ND_NBR; ++t) if (s2[t] == h) {
                if (lx && (int)(r2[t] & 0xffffu) >= pvx) fl |= 1u;
                if (ly && (int)(r2[t] >> 16) >= pvy) fl |= 2u;
            }
        }
        A.cut0[v[u]] = (unsigned char)fl;
        c[0][0] += lx && !(fl & 1u); c[0][1] += !lx; c[0][2] += lx && (fl & 1u);
        c[1][0] += ly && !(fl & 2u); c[1][1] += !ly; c[1][2] += ly && (fl & 2u);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int t = c[a][k];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane == 0 && t) atomicAdd(&g.cls[j][a][k], t);
        }
}

__global__ __launch_bounds__(1024) void nd_bigD_kernel(int L, int W, nd_args A, nd_big* __restrict__ G)
{
    __shared__ nd_gl<16> S;
    const int kset = blockIdx.x / W, j = blockIdx.x % W;
    const int h = (1 << L) + kset;
    pg_nd_set d = A.sets[h];
    const int m = d.size;
    if (m <= 0) return;
    const int* __restrict__ P = ((L & 1) ? A.perm1 : A.perm0) + d.lo;
    int* __restrict__ Pn = ((L & 1) ? A.perm0 : A.perm1) + d.lo;
    const int lo = j * ND_SL, n = min(ND_SL, m - lo);
    if (m <= A.leaf) {                                                      // (a leaf on a big level: slice 0 writes it)
        if (j == 0) { for (int i = threadIdx.x; i < m; i += 1024) A.order[d.out + i] = P[i]; if (threadIdx.x == 0) { d.kind = 1; A.sets[h] = d; } }
        return;
    }
    if (n <= 0) return;
    const nd_big& g = G[kset];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int v[ND_U]; unsigned r[ND_U]; unsigned char f[ND_U];
#pragma unroll
    for (int u = 0; u < ND_U; ++u) { const int i = 256 * wv + 64 * u + lane; v[u] = i < n ? P[lo + i] : -1; }
#pragma unroll
    for (int u = 0; u < ND_U; ++u) { r[u] = v[u] >= 0 ? A.rank_xy[v[u]] : 0u; f[u] = v[u] >= 0 ? A.cut0[v[u]] : 0; }
    int pvx, pvy;
    nd_big_pivots(g, S, m / 2, &pvx, &pvy);
    const double x0 = nd_unkey(g.bb[0]), x1 = nd_unkey(g.bb[1]), y0 = nd_unkey(g.bb[2]), y1 = nd_unkey(g.bb[3]);
    const bool byx = (x1 - x0) >= (y1 - y0);
    const bool two = m >= A.both_axes;
    int tot[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } }, bef[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };
    for (int jj = 0; jj < W; ++jj)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int k = 0; k < 3; ++k) { const int t = g.cls[jj][a][k]; tot[a][k] += t; if (jj < j) bef[a][k] += t; }
    const int cx_ = tot[0][2], cy_ = tot[1][2];
    const int c0 = byx ? cx_ : cy_, c1 = byx ? cy_ : cx_;
    const int win = (two && c1 < c0) ? 1 : 0;
    const bool wbx = win == 0 ? byx : !byx;
    const int wa = wbx ? 0 : 1;
    const int pv = wbx ? pvx : pvy, h2 = m / 2, nS = tot[wa][2], nA = h2 - nS, nB = m - h2;
    if (nA <= 0 || nB <= 0) {                                               // degenerate cut: index order (every slice its part)
        for (int i = threadIdx.x; i < n; i += 1024) A.order[d.out + lo + i] = P[lo + i];
        if (j == 0 && threadIdx.x == 0) { d.kind = 1; A.sets[h] = d; }
        return;
    }
    bool isA[ND_U], isB[ND_U], isS[ND_U];
    int cA = 0, cB = 0, cS = 0;
#pragma unroll
    for (int u = 0; u < ND_U; ++u) {
        const bool ok = v[u] >= 0, lower = ok && (int)(wbx ? (r[u] & 0xffffu) : (r[u] >> 16)) < pv, cutf = (f[u] & (wbx ? 1u : 2u)) != 0;
        isS[u] = lower && cutf; isA[u] = lower && !cutf; isB[u] = ok && !lower;
        cA += isA[u]; cB += isB[u]; cS += isS[u];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { cA += __shfl_xor(cA, o, 64); cB += __shfl_xor(cB, o, 64); cS += __shfl_xor(cS, o, 64); }
    if (lane == 0) { S.ired[0][wv] = cA; S.ired[1][wv] = cB; S.ired[2][wv] = cS; }
    __syncthreads();
    int bA = bef[wa][0], bB = bef[wa][1], bS = bef[wa][2];
    for (int k = 0; k < wv; ++k) { bA += S.ired[0][k]; bB += S.ired[1][k]; bS += S.ired[2][k]; }
#pragma unroll
    for (int u = 0; u < ND_U; ++u) {
        const unsigned long long mA = __ballot(isA[u]), mB = __ballot(isB[u]), mS = __ballot(isS[u]);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (isA[u]) { Pn[bA + __popcll(mA & below)] = v[u]; A.setid[v[u]] = 2 * h; }
        if (isB[u]) { Pn[nA + bB + __popcll(mB & below)] = v[u]; A.setid[v[u]] = 2 * h + 1; }
        if (isS[u]) { A.order[d.out + nA + nB + bS + __popcll(mS & below)] = v[u]; A.setid[v[u]] = 0; }
        bA += __popcll(mA); bB += __popcll(mB); bS += __popcll(mS);
    }
    if (j == 0 && threadIdx.x == 0) {
        d.kind = 2; d.nA = nA; d.nB = nB; A.sets[h] = d;
        pg_nd_set a; a.lo = d.lo; a.size = nA; a.out = d.out; a.kind = 0; a.nA = 0; a.nB = 0; A.sets[2 * h] = a;
        pg_nd_set b; b.lo = d.lo + nA; b.size = nB; b.out = d.out + nA; b.kind = 0; b.nA = 0; b.nB = 0; A.sets[2 * h + 1] = b;
    }
}

__global__ void nd_big_init_kernel(nd_big* __restrict__ G, int nsets)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = (int)(sizeof(nd_big) / sizeof(int));
    if (i >= nsets * per) return;
    int* w = reinterpret_cast<int*>(G) + i;
    const int k = i % per;
    // the box keys: minima start at the largest key, maxima at the smallest; everything else zero
    if (k == 0 || k == 1 || k == 4 || k == 5) *w = (int)0xffffffff; else *w = 0;      // bb[0] (min x) and bb[2] (min y): all ones
}

} // namespace

int pg_nd_levels(int n, int leaf)
{
    int L = 1; long long m = n;
    while (m > leaf) { m = (m + 1) / 2; ++L; }
    return L + 1;
}

size_t pg_nd_set_count(int nlev) { return (size_t)1 << (nlev + 1); }      // (the children the last level may still write)

// queue the whole ordering on `st` (asynchronous); sxy must be complete on that stream before
// (called from the ANALYSIS thread of a solve: it reports through its return value only -- the context's error string belongs to the
// thread that called the library, which turns a failure here into "order on the host")
int pg_nd_start(dsss_ctx* c, hipStream_t st, const pg_nd_buffers& B, const int* redges_host, int nedges, int leaf, int both_axes)
{
    (void)c;
#define ND_HIP(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); return DSSS_E_HIP; } } while (0)
    const int n = B.n;
    if (n <= 0 || n > 65536 || nedges < 0) return DSSS_E_ARG;
    ND_HIP(hipMemcpyAsync(B.edges, redges_host, (size_t)nedges * 2 * sizeof(int), hipMemcpyHostToDevice, st));
    ND_HIP(hipMemsetAsync(B.deg, 0, (size_t)n * sizeof(int), st));
    ND_HIP(hipMemsetAsync(B.sets, 0, pg_nd_set_count(B.nlev) * sizeof(pg_nd_set), st));
    {   // ranks: scratch = [4 x 1024 ints: histograms and fill cursors | the parameters of the bucket map] in front of the sets, bucket member lists in perm1 / order
        int* hist = reinterpret_cast<int*>(B.sets + pg_nd_set_count(B.nlev)); nd_rk* rk = reinterpret_cast<nd_rk*>(hist + 4 * ND_NB);
        hipLaunchKernelGGL(nd_minmax_kernel, dim3(1), dim3(ND_T), 0, st, n, B.sxy, rk, hist);
        hipLaunchKernelGGL(nd_bhist_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, B.sxy, rk, hist);
        hipLaunchKernelGGL(nd_bscan_kernel, dim3(1), dim3(ND_NB), 0, st, hist);
        hipLaunchKernelGGL(nd_bfill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, B.sxy, rk, hist, B.perm1, B.order);
        hipLaunchKernelGGL(nd_brank_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, B.sxy, rk, hist, B.perm1, B.order, B.rank_x, B.rank_y, reinterpret_cast<unsigned*>(B.cut1));      // (the packed ranks live in the second flag array: 4 bytes per node)
    }
    if (nedges > 0) hipLaunchKernelGGL(nd_deg_kernel, dim3((nedges + 255) / 256), dim3(256), 0, st, nedges, (const int2*)B.edges, B.deg);
    hipLaunchKernelGGL(nd_scan_kernel, dim3(1), dim3(ND_T), 0, st, n, B.deg, B.adj_ptr, B.adj_cur, B.perm0, B.setid);
    if (nedges > 0) hipLaunchKernelGGL(nd_fill_kernel, dim3((nedges + 255) / 256), dim3(256), 0, st, nedges, (const int2*)B.edges, B.adj_cur, B.adj_idx);
    { pg_nd_set& root = B.h_sets[0]; root.lo = 0; root.size = n; root.out = 0; root.kind = 0; root.nA = 0; root.nB = 0; }      // (page-locked: read by the copy below, overwritten by the download at the end -- same stream)
    ND_HIP(hipMemcpyAsync(B.sets + 1, B.h_sets, sizeof(pg_nd_set), hipMemcpyHostToDevice, st));
    ND_HIP(hipMemsetAsync(B.order, 0xff, (size_t)n * sizeof(int), st));      // (a position nobody fills says the levels did not suffice)
    nd_args A;
    A.n = n; A.leaf = leaf; A.both_axes = both_axes; A.sxy = B.sxy; A.rank_x = B.rank_x; A.rank_y = B.rank_y; A.adj_ptr = B.adj_ptr; A.adj_idx = B.adj_idx;
    A.perm0 = B.perm0; A.perm1 = B.perm1; A.setid = B.setid; A.cut0 = B.cut0; A.rank_xy = reinterpret_cast<const unsigned*>(B.cut1); A.order = B.order; A.sets = B.sets;
    A.fail = reinterpret_cast<int*>(B.sets);                          // (heap node 0 is nobody's: its first word is the failure flag, zeroed with the sets)
    {   // a set of level L holds at most ceil(n / 2^L) nodes (the larger child is the upper half)
        nd_big* Gb = reinterpret_cast<nd_big*>(reinterpret_cast<char*>(B.sets + pg_nd_set_count(B.nlev)) + 4 * ND_NB * sizeof(int) + 64);      // behind the rank kernels' scratch: one record per big set (at most 16 sets of more than 4 096 nodes)
        const bool big_ok = n <= ND_SL * ND_WMAX;
        long long bound = n;
        for (int L = 0; L < B.nlev; ++L) {
            const unsigned nsets = 1u << L;
            if (bound > 1024 * ND_U && big_ok) {
                const int W = (int)((bound + ND_SL - 1) / ND_SL);
                hipLaunchKernelGGL(nd_big_init_kernel, dim3((nsets * (unsigned)(sizeof(nd_big) / sizeof(int)) + 255) / 256), dim3(256), 0, st, Gb, (int)nsets);
                hipLaunchKernelGGL(nd_bigA_kernel, dim3(nsets * W), dim3(1024), 0, st, L, W, A, Gb);
                hipLaunchKernelGGL(nd_bigB_kernel, dim3(nsets * W), dim3(1024), 0, st, L, W, A, Gb);
                hipLaunchKernelGGL(nd_bigC_kernel, dim3(nsets * W), dim3(1024), 0, st, L, W, A, Gb);
                hipLaunchKernelGGL(nd_bigD_kernel, dim3(nsets * W), dim3(1024), 0, st, L, W, A, Gb);
            }
            else if (bound > 1024 * ND_U) hipLaunchKernelGGL(nd_level_kernel, dim3(nsets), dim3(ND_T), 0, st, L, A);
            else if (bound > 256 * ND_U) hipLaunchKernelGGL(nd_level_group_kernel<1024>, dim3(nsets), dim3(1024), 0, st, L, A);
            else if (bound > 64 * ND_U) hipLaunchKernelGGL(nd_level_group_kernel<256>, dim3(nsets), dim3(256), 0, st, L, A);
            else hipLaunchKernelGGL(nd_level_group_kernel<64>, dim3((nsets + 3) / 4), dim3(256), 0, st, L, A);
            bound = (bound + 1) / 2;
        }
    }
    ND_HIP(hipGetLastError());
    ND_HIP(hipMemcpyAsync(B.h_order, B.order, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
    {   // heap nodes 1 .. 63: the top of the tree (the host's parallel column-structure pass follows it).  A small graph has fewer than 64
        // set records: only those that exist come back, the rest of the host copy reads "not reached" (behind them lies scratch)
        const size_t have = std::min<size_t>(64, pg_nd_set_count(B.nlev));
        for (size_t k = have; k < 64; ++k) { pg_nd_set& z = B.h_sets[k]; z.lo = 0; z.size = 0; z.out = 0; z.kind = 0; z.nA = 0; z.nB = 0; }
        ND_HIP(hipMemcpyAsync(B.h_sets, B.sets, have * sizeof(pg_nd_set), hipMemcpyDeviceToHost, st));
    }
    ND_HIP(hipEventRecord(B.done, st));
#undef ND_HIP
    return DSSS_OK;
}
