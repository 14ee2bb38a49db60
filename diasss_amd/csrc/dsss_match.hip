// diasss_amd/csrc/dsss_match.hip -- batched FEAmatcher on the device (gfx950).
// Restates /root/reference/src/core/FEAmatcher.cpp: GeoNearNeighSearch first stage (:79-183) as an all-pairs
// gate + 256-bit Hamming kernel with LDS-staged descriptor tiles, SCC_x (:186-248) with one hypothesis per lane,
// ConsistentCheck (:323-405) + RobustMatching rows (:35-45) + Optimizer::GetKpsPairs (optimizer.cpp:575-639)
// as ordered block compactions.  Integer results are bit-exact against oracle/orc_match.c.
#include "dsss_internal.h"
#include <functional>

#define MT_TILE 256          // threads per block = keypoints of A per block = B entries per LDS tile

// ------------------------------------------------------------------ K9: gate + nearest / second nearest
// grid: (tiles of A, 2 * active pairs).  Thread = one keypoint a of the query frame; the reference frame's
// descriptors (32 B) and geo points (16 B) are staged through LDS in tiles of 256 and read as wave-wide
// broadcasts (all lanes same address: conflict-free).  Order of b is index order, so "first index wins" ties
// (FEAmatcher.cpp:152-161) come out exactly as in the scalar loop.
// MODE 0: Hamming on the 32 ORB bytes; 1: L2 on the same 32 bytes (what the shipped reference feeds its L2 branch); 2: L2 on the 128-element
// rows of DSSS_DESC_SIFT128 (`desc` is then the 128-byte store).  The elements are integers 0..255, so the squared distance is an exact
// integer: |a|^2 + |b|^2 - 2 a.b with the dot product on v_dot4_u32_u8 (32 of them per candidate).
__device__ inline unsigned dot128(const uint4 (&q)[8], const uint4* __restrict__ b)
{
    unsigned acc = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const uint4 v = b[w];
        acc = __builtin_amdgcn_udot4(q[w].x, v.x, acc, false); acc = __builtin_amdgcn_udot4(q[w].y, v.y, acc, false);
        acc = __builtin_amdgcn_udot4(q[w].z, v.z, acc, false); acc = __builtin_amdgcn_udot4(q[w].w, v.w, acc, false);
    }
    return acc;
}
template <int MODE>
__global__ __launch_bounds__(MT_TILE) void match_nn_kernel(
    const int* __restrict__ act_s, const int* __restrict__ act_t, const int* __restrict__ nkp,
    const uint8_t* __restrict__ desc, const double* __restrict__ geo, const double* __restrict__ bbox,
    int kcap, double gate_T, int bound_same, int bound_diff, double l2_bound, double ratio_max,
    int32_t* __restrict__ corres_nn)
{
    constexpr bool L2 = MODE != 0;
    constexpr int DW = MODE == 2 ? 8 : 1;          // uint4 per half-descriptor slot
    __shared__ uint4 s_lo[MT_TILE * DW], s_hi[MODE == 2 ? 1 : MT_TILE];
    __shared__ unsigned s_n2[MODE == 2 ? MT_TILE : 1];
    __shared__ double2 s_geo[MT_TILE];
    const int pd = blockIdx.y, p = pd >> 1, dir = pd & 1;
    const int fa = dir ? act_t[p] : act_s[p];
    const int fb = dir ? act_s[p] : act_t[p];
    const int na = nkp[fa], nb = nkp[fb];
    const int a = blockIdx.x * MT_TILE + threadIdx.x;
    if (blockIdx.x * MT_TILE >= na) return;                       // whole block beyond the query frame
    int32_t* out = corres_nn + (size_t)pd * kcap;
    const double bx0 = bbox[fb * 4 + 0], bx1 = bbox[fb * 4 + 1], by0 = bbox[fb * 4 + 2], by1 = bbox[fb * 4 + 3];
    double ax = 0, ay = 0;
    uint4 alo = make_uint4(0, 0, 0, 0), ahi = alo;
    uint4 q[8]; unsigned qn2 = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) q[w] = make_uint4(0, 0, 0, 0);
    bool live = a < na;
    if (live) {
        const double2 g = reinterpret_cast<const double2*>(geo)[(size_t)fa * kcap + a];
        ax = g.x; ay = g.y;
        live = !(ax < bx0 || ay < by0 || ax > bx1 || ay > by1);   // FEAmatcher.cpp:84
        if (MODE == 2) {
            const uint4* d = reinterpret_cast<const uint4*>(desc + ((size_t)fa * kcap + a) * 128);
#pragma unroll
            for (int w = 0; w < 8; ++w) q[w] = d[w];
            qn2 = dot128(q, d);
        } else {
            const uint4* d = reinterpret_cast<const uint4*>(desc + ((size_t)fa * kcap + a) * 32);
            alo = d[0]; ahi = d[1];
        }
    }
    if (!__syncthreads_or(live)) { if (a < na) out[a] = -1; return; }
    int best = L2 ? 1000000 : 1000, second = best, best_id = -1, nc = 0;
    const uint4* bdesc = reinterpret_cast<const uint4*>(desc + (size_t)fb * kcap * (MODE == 2 ? 128 : 32));
    const double2* bgeo = reinterpret_cast<const double2*>(geo) + (size_t)fb * kcap;
    for (int b0 = 0; b0 < nb; b0 += MT_TILE) {
        const int bj = b0 + threadIdx.x;
        if (bj < nb) {
            if (MODE == 2) {
                uint4 t[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) { t[w] = bdesc[8 * bj + w]; s_lo[threadIdx.x * 8 + w] = t[w]; }
                s_n2[threadIdx.x] = dot128(t, bdesc + 8 * bj);
            } else { s_lo[threadIdx.x] = bdesc[2 * bj]; s_hi[threadIdx.x] = bdesc[2 * bj + 1]; }
            s_geo[threadIdx.x] = bgeo[bj];
        }
        __syncthreads();
        const int cnt = min(MT_TILE, nb - b0);
        if (live) {
            for (int j = 0; j < cnt; ++j) {
                const double2 g = s_geo[j];
                const double dx = ax - g.x, dy = ay - g.y;
                const double d2 = dx * dx + dy * dy;
                if (d2 < gate_T) {                               // sqrt(d2) < radius, FEAmatcher.cpp:92-93
                    int d;
                    if (MODE == 2) {
                        d = (int)(qn2 + s_n2[j] - 2u * dot128(q, s_lo + 8 * j));
                    } else {
                    const uint4 lo = s_lo[j], hi = s_hi[j];
                    if (!L2) {
                        d = __popc(alo.x ^ lo.x) + __popc(alo.y ^ lo.y) + __popc(alo.z ^ lo.z) + __popc(alo.w ^ lo.w)
                          + __popc(ahi.x ^ hi.x) + __popc(ahi.y ^ hi.y) + __popc(ahi.z ^ hi.z) + __popc(ahi.w ^ hi.w);
                    } else {
                        const unsigned wa[8] = { alo.x, alo.y, alo.z, alo.w, ahi.x, ahi.y, ahi.z, ahi.w };
                        const unsigned wb[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
                        d = 0;
#pragma unroll
                        for (int w = 0; w < 8; ++w)
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int e = (int)((wa[w] >> (8 * k)) & 255u) - (int)((wb[w] >> (8 * k)) & 255u);
                                d += e * e;
                            }
                    }
                    }
                    ++nc;
                    if (d < best) { second = best; best = d; best_id = b0 + j; }
                    else if (d < second) second = d;
                }
            }
        }
        __syncthreads();
    }
    if (a >= na) return;
    int res = -1;
    if (live && nc > 0) {
        if (!L2) {
            const int bound = ((fa % 2) != (fb % 2)) ? bound_diff : bound_same;
            const double r = (double)best / (double)second;
            if (best_id != -1 && best <= bound && r <= ratio_max && second != 1000) res = best_id;
            else if (nc == 1 && best <= bound) res = best_id;
        } else {
            const double bd = sqrt((double)best), sd = sqrt((double)second);   // cv::norm(NORM_L2), FEAmatcher.cpp:113
            const double r = bd / sd;
            if (best_id != -1 && bd < l2_bound && r <= ratio_max) res = best_id;
            else if (nc == 1 && bd < l2_bound) res = best_id;
        }
    }
    out[a] = res;
}

// ------------------------------------------------------------------ K9 on a geo grid
// (best, second, lowest index at the minimum, number of candidates) of FEAmatcher.cpp:86-161 do not depend on the order in which the
// candidates inside the search circle are met, so only the keypoints that CAN be inside it are looked at: every frame of an active pair
// gets its keypoints counting-sorted by the cell of a grid over its geo box whose cells are a hair wider than the radius (so that two
// points closer than the radius lie in the same or in adjacent cells whatever the rounding of the cell arithmetic), and a query walks the
// three rows of three cells around its own cell -- three contiguous ranges of the sorted arrays.  The gate itself is the SAME double
// comparison as in the all-pairs kernel, on the same coordinates: the set that passes it is the same set.  At C3 a query meets ~25
// candidates instead of 2 000.
struct mt_grid { int W, H, off, pad; };            // cells per row, rows, first entry of the frame's (W H + 1) cell offsets

__device__ inline int mt_cell(double v, double o, double inv_cs, int n)
{
    const int c = (int)((v - o) * inv_cs);           // monotone in v (saturating conversion; NaN -> 0), and so is the clamp
    return min(max(c, 0), n - 1);
}

// one workgroup per frame: count per cell (global atomics), exclusive scan over the cells, scatter.  The order inside a cell is
// whatever the atomics give; nothing downstream depends on it.
#define MTG_THREADS 1024
__global__ __launch_bounds__(MTG_THREADS) void mt_grid_build_kernel(
    const int* __restrict__ frames, const mt_grid* __restrict__ tab, const int* __restrict__ nkp,
    const uint8_t* __restrict__ desc, const double* __restrict__ geo, const double* __restrict__ bbox, int kcap, double inv_cs,
    int* __restrict__ start_all, int* __restrict__ cur_all, double2* __restrict__ s_geo, uint4* __restrict__ s_desc, int* __restrict__ s_idx)
{
    __shared__ int s_w[MTG_THREADS / 64];
    __shared__ int s_base;
    const int f = frames[blockIdx.x];
    const mt_grid T = tab[f];
    const int n = nkp[f], cells = T.W * T.H;
    int* __restrict__ start = start_all + T.off; int* __restrict__ cur = cur_all + T.off;
    const double ox = bbox[f * 4 + 0], oy = bbox[f * 4 + 2];
    const double2* __restrict__ g = reinterpret_cast<const double2*>(geo) + (size_t)f * kcap;
    for (int i = threadIdx.x; i <= cells; i += MTG_THREADS) cur[i] = 0;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += MTG_THREADS) {
        const double2 p = g[i];
        atomicAdd(&cur[mt_cell(p.y, oy, inv_cs, T.H) * T.W + mt_cell(p.x, ox, inv_cs, T.W)], 1);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c0 = 0; c0 < cells; c0 += MTG_THREADS) {
        const int i = c0 + threadIdx.x;
        const int v = i < cells ? __hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;      // (counted by atomics at L2: not through this CU's vector cache)
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        int base = s_base;
        for (int k = 0; k < wv; ++k) base += s_w[k];
        if (i < cells) { start[i] = base + inc - v; cur[i] = base + inc - v; }
        __syncthreads();
        if (threadIdx.x == MTG_THREADS - 1) s_base = base + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) start[cells] = n;
    const uint4* __restrict__ d = reinterpret_cast<const uint4*>(desc + (size_t)f * kcap * 32);
    const size_t ob = (size_t)f * kcap;
    for (int i = threadIdx.x; i < n; i += MTG_THREADS) {
        const double2 p = g[i];
        const int pos = atomicAdd(&cur[mt_cell(p.y, oy, inv_cs, T.H) * T.W + mt_cell(p.x, ox, inv_cs, T.W)], 1);
        s_geo[ob + pos] = p; s_desc[2 * (ob + pos)] = d[2 * i]; s_desc[2 * (ob + pos) + 1] = d[2 * i + 1]; s_idx[ob + pos] = i;
    }
}

// grid: (tiles of MT_TILE / MT_LPQ queries, 2 * active pairs).  MT_LPQ = 8 ADJACENT LANES PER QUERY: lane s of the eight takes the
// candidates s, s + 8, ... of each of the three row ranges (eight neighbouring geo points are one 128-byte read), and the eight partial
// results are folded by three butterfly steps -- the fold of two (best, second, lowest index, count) is as order-free as the update.  The
// kernel's time is its longest chain of dependent cache round trips: keypoints come in clumps (one landmark at several pyramid levels),
// and with one lane per query a clump of 200 in a cell is 600 round trips for its lanes while the rest of the chip has long finished
// (323 us at C3 against 926 us for the all-pairs kernel; with eight lanes per query and cells of half the radius 162 us + 27 us for the sort).  Queries in THEIR OWN sorted order:
// neighbouring groups are neighbours on the sea floor and walk the same cells.  No LDS, no barrier.
#ifndef MT_LPQ
#define MT_LPQ 8
#endif
#ifndef MT_SUB
#define MT_SUB 2                 // cells per radius: a query walks (2 MT_SUB + 1)^2 cells, 25 cells of 4 m = 400 m^2 against 9 of 8 m = 576 m^2 (the circle: 201 m^2)
#endif
template <int MODE>          // as match_nn_kernel; MODE 2 reads the 128-byte rows of the feature store through the sorted index (s_desc = the store)
__global__ __launch_bounds__(MT_TILE) void match_grid_kernel(
    const int* __restrict__ act_s, const int* __restrict__ act_t, const int* __restrict__ nkp, const double* __restrict__ bbox,
    const mt_grid* __restrict__ tab, const int* __restrict__ start_all, const double2* __restrict__ s_geo, const uint4* __restrict__ s_desc,
    const int* __restrict__ s_idx, int kcap, double inv_cs, double gate_T, int bound_same, int bound_diff, double l2_bound, double ratio_max,
    int32_t* __restrict__ corres_nn)
{
    constexpr bool L2 = MODE != 0;
    const int pd = blockIdx.y, p = pd >> 1, dir = pd & 1;
    const int fa = dir ? act_t[p] : act_s[p];
    const int fb = dir ? act_s[p] : act_t[p];
    const int na = nkp[fa];
    const int t = blockIdx.x * (MT_TILE / MT_LPQ) + (threadIdx.x / MT_LPQ), sub = threadIdx.x & (MT_LPQ - 1);
    if (t >= na) return;                                         // (whole groups of eight lanes leave together: the shuffles below stay inside a group)
    const size_t sa = (size_t)fa * kcap + t;
    const double2 ga = s_geo[sa];
    const int a = s_idx[sa];
    int32_t* out = corres_nn + (size_t)pd * kcap;
    const double bx0 = bbox[fb * 4 + 0], bx1 = bbox[fb * 4 + 1], by0 = bbox[fb * 4 + 2], by1 = bbox[fb * 4 + 3];
    const double ax = ga.x, ay = ga.y;
    if (ax < bx0 || ay < by0 || ax > bx1 || ay > by1) { if (sub == 0) out[a] = -1; return; }      // FEAmatcher.cpp:84
    uint4 alo = make_uint4(0, 0, 0, 0), ahi = alo;
    uint4 q[8]; unsigned qn2 = 0;
    if (MODE == 2) {
        const uint4* d = s_desc + ((size_t)fa * kcap + a) * 8;
#pragma unroll
        for (int w = 0; w < 8; ++w) q[w] = d[w];
        qn2 = dot128(q, d);
    } else { alo = s_desc[2 * sa]; ahi = s_desc[2 * sa + 1]; }
    const mt_grid T = tab[fb];
    const int* __restrict__ start = start_all + T.off;
    const int cx = mt_cell(ax, bx0, inv_cs, T.W), cy = mt_cell(ay, by0, inv_cs, T.H);
    const int c0 = max(cx - MT_SUB, 0), c1 = min(cx + MT_SUB, T.W - 1);
    int jb[2 * MT_SUB + 1], je[2 * MT_SUB + 1];                  // the rows' ranges, fetched together
#pragma unroll
    for (int r = 0; r < 2 * MT_SUB + 1; ++r) {
        const int row = cy - MT_SUB + r;
        const bool ok = row >= 0 && row < T.H;
        jb[r] = ok ? start[row * T.W + c0] : 0; je[r] = ok ? start[row * T.W + c1 + 1] : 0;
    }
    const size_t sb = (size_t)fb * kcap;
    int best = L2 ? 1000000 : 1000, second = best, best_id = -1, nc = 0;
#pragma unroll
    for (int r = 0; r < 2 * MT_SUB + 1; ++r)
        for (int j = jb[r] + sub; j < je[r]; j += MT_LPQ) {
            const double2 g = s_geo[sb + j];
            const double dx = ax - g.x, dy = ay - g.y;
            const double d2 = dx * dx + dy * dy;
            if (d2 < gate_T) {                                   // sqrt(d2) < radius, FEAmatcher.cpp:92-93
                const int id = s_idx[sb + j];
                int d;
                if (MODE == 2) {
                    const uint4* bd = s_desc + (sb + id) * 8;
                    uint4 t[8];
#pragma unroll
                    for (int w = 0; w < 8; ++w) t[w] = bd[w];
                    d = (int)(qn2 + dot128(t, bd) - 2u * dot128(q, bd));
                } else {
                const uint4 lo = s_desc[2 * (sb + j)], hi = s_desc[2 * (sb + j) + 1];
                if (!L2) {
                    d = __popc(alo.x ^ lo.x) + __popc(alo.y ^ lo.y) + __popc(alo.z ^ lo.z) + __popc(alo.w ^ lo.w)
                      + __popc(ahi.x ^ hi.x) + __popc(ahi.y ^ hi.y) + __popc(ahi.z ^ hi.z) + __popc(ahi.w ^ hi.w);
                } else {
                    const unsigned wa[8] = { alo.x, alo.y, alo.z, alo.w, ahi.x, ahi.y, ahi.z, ahi.w };
                    const unsigned wb[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
                    d = 0;
#pragma unroll
                    for (int w = 0; w < 8; ++w)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int e = (int)((wa[w] >> (8 * k)) & 255u) - (int)((wb[w] >> (8 * k)) & 255u);
                            d += e * e;
                        }
                }
                }
                ++nc;
                // the scalar loop's "first index wins" (FEAmatcher.cpp:152-161) without its order: the two smallest distances of the
                // candidates (a tie at the minimum makes the second equal to it) and the lowest index at the minimum
                if (d < best) { second = best; best = d; best_id = id; }
                else if (d == best) { second = d; if (id < best_id) best_id = id; }
                else if (d < second) second = d;
            }
        }
    // fold the eight lanes of the query: the two smallest of the union, the lowest index among the minima, the counts added
#pragma unroll
    for (int o = 1; o < MT_LPQ; o <<= 1) {
        const int ob = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64), oi = __shfl_xor(best_id, o, 64), on = __shfl_xor(nc, o, 64);
        if (ob < best) { second = min(best, os); best = ob; best_id = oi; }
        else if (ob == best) { second = best; best_id = min(best_id, oi); }      // (both -1 when nothing beat the initial value)
        else second = min(second, ob);
        nc += on;
    }
    if (sub != 0) return;
    int res = -1;
    if (nc > 0) {
        if (!L2) {
            const int bound = ((fa % 2) != (fb % 2)) ? bound_diff : bound_same;
            const double r = (double)best / (double)second;
            if (best_id != -1 && best <= bound && r <= ratio_max && second != 1000) res = best_id;
            else if (nc == 1 && best <= bound) res = best_id;
        } else {
            const double bd = sqrt((double)best), sd = sqrt((double)second);   // cv::norm(NORM_L2), FEAmatcher.cpp:113
            const double r = bd / sd;
            if (best_id != -1 && bd < l2_bound && r <= ratio_max) res = best_id;
            else if (nc == 1 && bd < l2_bound) res = best_id;
        }
    }
    out[a] = res;
}

// per-kernel profile only (outside the matcher's timed scope): the gate evaluations match_grid_kernel performs = the sizes of the row
// ranges its live queries walk; bench.py prices the kernel by them
__global__ __launch_bounds__(MT_TILE) void mt_grid_count_kernel(
    const int* __restrict__ act_s, const int* __restrict__ act_t, const int* __restrict__ nkp, const double* __restrict__ bbox,
    const mt_grid* __restrict__ tab, const int* __restrict__ start_all, const double2* __restrict__ s_geo, int kcap, double inv_cs,
    unsigned long long* __restrict__ n_evals)
{
    __shared__ unsigned long long s_tot;
    if (threadIdx.x == 0) s_tot = 0;
    __syncthreads();
    const int pd = blockIdx.y, p = pd >> 1, dir = pd & 1;
    const int fa = dir ? act_t[p] : act_s[p];
    const int fb = dir ? act_s[p] : act_t[p];
    const int t = blockIdx.x * MT_TILE + threadIdx.x;
    if (t < nkp[fa]) {
        const double2 ga = s_geo[(size_t)fa * kcap + t];
        const double bx0 = bbox[fb * 4 + 0], bx1 = bbox[fb * 4 + 1], by0 = bbox[fb * 4 + 2], by1 = bbox[fb * 4 + 3];
        if (!(ga.x < bx0 || ga.y < by0 || ga.x > bx1 || ga.y > by1)) {
            const mt_grid T = tab[fb];
            const int* __restrict__ start = start_all + T.off;
            const int cx = mt_cell(ga.x, bx0, inv_cs, T.W), cy = mt_cell(ga.y, by0, inv_cs, T.H);
            const int c0 = max(cx - MT_SUB, 0), c1 = min(cx + MT_SUB, T.W - 1);
            int tot = 0;
            for (int row = max(cy - MT_SUB, 0); row <= min(cy + MT_SUB, T.H - 1); ++row) tot += start[row * T.W + c1 + 1] - start[row * T.W + c0];
            atomicAdd(&s_tot, (unsigned long long)tot);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_tot) atomicAdd(n_evals, s_tot);
}

// ------------------------------------------------------------------ block scan helper (256 threads = 4 waves)
__device__ inline int block_scan_excl(int v, int* total, int* s_w)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    __syncthreads();                     // protect s_w from the previous call
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { int t = s_w[k]; if (k < w) base += t; tot += t; }
    *total = tot;
    return base + inc - v;
}

// ------------------------------------------------------------------ K10: SCC_x (FEAmatcher.cpp:186-248)
// one block per directed active pair; hypotheses are independent given the fixed cv::RNG stream (the reference
// default-constructs the generator on every call, :59), so each lane evaluates whole hypotheses and the
// "first strictly larger inlier set wins" rule (:237) becomes max count, then lowest iteration index.
__global__ __launch_bounds__(256) void scc_kernel(
    const int* __restrict__ act_s, const int* __restrict__ act_t, const int* __restrict__ nkp,
    const int* __restrict__ frows, const dsss_kp* __restrict__ kps, int kcap,
    const uint32_t* __restrict__ rng_raw, int iters, double pix_err,
    const int32_t* __restrict__ corres_nn, int32_t* __restrict__ corres,
    int* __restrict__ scc_hist, int* __restrict__ scc_count, double* __restrict__ scc_model)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int* id_loc = reinterpret_cast<int*>(smem);                       // [kcap]
    float* xv = reinterpret_cast<float*>(smem + (size_t)kcap * 4);    // [kcap] X_tmp per accepted match, ID_loc order
    int* cnts = reinterpret_cast<int*>(smem + (size_t)kcap * 8);      // [iters]
    __shared__ int s_w[4];
    __shared__ int s_best_it, s_hist, s_nloc;
    __shared__ double s_model;
    const int pd = blockIdx.x, p = pd >> 1, dir = pd & 1;
    const int fa = dir ? act_t[p] : act_s[p];
    const int fb = dir ? act_s[p] : act_t[p];
    const int na = nkp[fa];
    const bool flip = (fa % 2) != (fb % 2);
    const float rows_ref = (float)frows[fb];
    const dsss_kp* ka = kps + (size_t)fa * kcap;
    const dsss_kp* kb = kps + (size_t)fb * kcap;
    const int32_t* nn = corres_nn + (size_t)pd * kcap;
    int32_t* out = corres + (size_t)pd * kcap;
    // ID_loc in index order (:132,169)
    int base = 0;
    for (int c0 = 0; c0 < na; c0 += 256) {
        const int i = c0 + threadIdx.x;
        const int m = (i < na) ? nn[i] : -1;
        int tot;
        const int pos = block_scan_excl(m != -1, &tot, s_w);
        if (m != -1) {
            const float ya = ka[i].y, yb = kb[m].y;
            id_loc[base + pos] = i;
            xv[base + pos] = flip ? fabsf(ya - ((rows_ref - yb) + 1.0f)) : fabsf(ya - yb);   // :209-212,222-227
        }
        base += tot;
    }
    if (threadIdx.x == 0) s_nloc = base;
    __syncthreads();
    const int nloc = s_nloc;
    if (nloc == 0) {
        for (int i = threadIdx.x; i < na; i += 256) out[i] = -1;
        if (threadIdx.x == 0) { scc_hist[pd] = 0; scc_count[pd] = 0; scc_model[pd] = 0.0; }
        return;
    }
    for (int it = threadIdx.x; it < iters; it += 256) {
        const int s0 = (int)(rng_raw[2 * it] % (uint32_t)nloc), s1 = (int)(rng_raw[2 * it + 1] % (uint32_t)nloc);  // :201
        double model = 0.0;
        model = model + xv[s0];
        model = model + xv[s1];
        model = model / 2;
        int cnt = 0;
        for (int k = 0; k < nloc; ++k) cnt += fabs(model - (double)xv[k]) <= pix_err;
        cnts[it] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int fin = 0, best_it = -1, hist = 0;
        for (int it = 0; it < iters; ++it) if (fin < cnts[it]) { fin = cnts[it]; best_it = it; ++hist; }   // :237-242
        s_best_it = best_it; s_hist = hist;
        double model = 0.0;
        if (best_it >= 0) {
            const int s0 = (int)(rng_raw[2 * best_it] % (uint32_t)nloc), s1 = (int)(rng_raw[2 * best_it + 1] % (uint32_t)nloc);
            model = model + xv[s0]; model = model + xv[s1]; model = model / 2;
        }
        s_model = model;
        scc_hist[pd] = hist; scc_count[pd] = fin; scc_model[pd] = model;
    }
    __syncthreads();
    const double model = s_model; const bool any = s_best_it >= 0;
    for (int i = threadIdx.x; i < na; i += 256) out[i] = -1;
    __syncthreads();
    for (int k = threadIdx.x; k < nloc; k += 256) {
        const int i = id_loc[k];
        if (any && fabs(model - (double)xv[k]) <= pix_err) out[i] = nn[i];
    }
}

// ------------------------------------------------------------------ ConsistentCheck + rows + GetKpsPairs
// block per active pair.  WRITE = false counts, WRITE = true fills rows6/kp7 at the scanned offsets, so the
// placement is deterministic (pair order, then the reference's in-pair order).
template <bool WRITE>
__global__ __launch_bounds__(256) void pair_rows_kernel(
    const int* __restrict__ act_s, const int* __restrict__ act_t, const int* __restrict__ nkp,
    const int* __restrict__ frows, const int* __restrict__ fcols, const dsss_kp* __restrict__ kps, int kcap,
    const int32_t* __restrict__ corres, const int* __restrict__ scc_hist, const double* __restrict__ scc_model,
    double merge_thr, const double* const* __restrict__ alt_ptr, const double* const* __restrict__ gr_ptr,
    const double* const* __restrict__ pose_ptr,
    int* __restrict__ row_cnt, int* __restrict__ kp7_cnt, const int* __restrict__ row_off, const int* __restrict__ kp7_off,
    double* __restrict__ rows6, double* __restrict__ kp7, int* __restrict__ kp7_pair, uint8_t* __restrict__ kp7_flip)
{
    __shared__ int s_w[4];
    __shared__ int s_red[2];
    const int p = blockIdx.x;
    const int fs = act_s[p], ft = act_t[p];
    const int n1 = nkp[fs], n2 = nkp[ft];
    const int32_t* c1 = corres + (size_t)(2 * p) * kcap;
    const int32_t* c2 = corres + (size_t)(2 * p + 1) * kcap;
    const dsss_kp* ks = kps + (size_t)fs * kcap;
    const dsss_kp* kt = kps + (size_t)ft * kcap;
    // merge decision (:341-345); an empty scc history falls to the "larger direction" branch
    double img_diff = 0;
    if ((fs % 2) != (ft % 2)) img_diff = (double)abs(frows[fs] - frows[ft]);
    bool merge = false;
    if (scc_hist[2 * p] > 0 && scc_hist[2 * p + 1] > 0)
        merge = fabs(fabs(scc_model[2 * p] - scc_model[2 * p + 1]) - img_diff) <= merge_thr;
    bool use1 = true, use2 = true;
    if (!merge) {
        if (threadIdx.x < 2) s_red[threadIdx.x] = 0;
        __syncthreads();
        int l1 = 0, l2 = 0;
        for (int i = threadIdx.x; i < n1; i += 256) l1 += c1[i] != -1;
        for (int i = threadIdx.x; i < n2; i += 256) l2 += c2[i] != -1;
        atomicAdd(&s_red[0], l1); atomicAdd(&s_red[1], l2);
        __syncthreads();
        use1 = s_red[0] > s_red[1]; use2 = !use1;                  // :378,391 ties -> direction 2
    }
    const int ngr_s = fcols[fs] / 2, ngr_t = fcols[ft] / 2;      // ground_ranges.size()
    const double* alt_s = alt_ptr[fs]; const double* alt_t = alt_ptr[ft];
    const double* gr_s = gr_ptr[fs]; const double* gr_t = gr_ptr[ft];
    const double* pose_s = pose_ptr[fs]; const double* pose_t = pose_ptr[ft];
    const int ro = WRITE ? row_off[p] : 0, ko = WRITE ? kp7_off[p] : 0;
    int rbase = 0, kbase = 0, flip_s_run = 0, flip_t_run = 0;
    for (int phase = 0; phase < 2; ++phase) {
        if (phase == 0 && !use1) continue;
        if (phase == 1 && !use2) continue;
        const int n = phase == 0 ? n1 : n2;
        for (int c0 = 0; c0 < n; c0 += 256) {
            const int i = c0 + threadIdx.x;
            int src = -1, tgt = -1;
            if (i < n) {
                if (phase == 0) { const int m = c1[i]; if (m != -1 && !(merge && c2[m] == i)) { src = i; tgt = m; } }   // :348-360
                else { const int m = c2[i]; if (m != -1) { src = m; tgt = i; } }                                    // :362-371
            }
            const bool valid = src >= 0;
            int tot;
            const int pos = block_scan_excl(valid, &tot, s_w);
            double ys = 0, xs = 0, yt = 0, xt = 0;
            if (valid) {
                ys = (double)ks[src].y; xs = (double)ks[src].x; yt = (double)kt[tgt].y; xt = (double)kt[tgt].x;
                if (WRITE) {
                    double* r = rows6 + (size_t)(ro + rbase + pos) * 6;                                              // :37-40
                    r[0] = fs; r[1] = ft; r[2] = ys; r[3] = xs; r[4] = yt; r[5] = xt;
                }
            }
            rbase += tot;
            // GetKpsPairs (optimizer.cpp:596-631): int truncation, nadir rejection
            bool v7 = false; int ps = 0, bs = 0, pt = 0, bt = 0;
            if (valid) {
                ps = (int)ys; bs = (int)xs; pt = (int)yt; bt = (int)xt;
                v7 = !(abs(bs - ngr_s) < 20 || abs(bt - ngr_t) < 20);
            }
            int tot7;
            const int pos7 = block_scan_excl(v7, &tot7, s_w);
            // sticky yaw compensation flags of LoopClosingTFs (optimizer.cpp:650,698-703): prefix OR in list order
            const double thr = 2 * DSSS_PI_REF / 3;
            const int fs_here = v7 && fabs(pose_s[(size_t)ps * 6 + 2]) > thr;
            const int ft_here = v7 && fabs(pose_t[(size_t)pt * 6 + 2]) > thr;
            int tfs, tft;
            const int pre_s = block_scan_excl(fs_here, &tfs, s_w) + fs_here;
            const int pre_t = block_scan_excl(ft_here, &tft, s_w) + ft_here;
            if (WRITE && v7) {
                const size_t o = (size_t)(ko + kbase + pos7);
                int gis = abs(bs - ngr_s), git = abs(bt - ngr_t);
                gis = gis < ngr_s ? gis : ngr_s - 1; git = git < ngr_t ? git : ngr_t - 1;
                const double as = alt_s[ps], gs = gr_s[gis], at = alt_t[pt], gt = gr_t[git];
                double* q = kp7 + o * 7;
                q[0] = ps; q[1] = bs; q[2] = sqrt(as * as + gs * gs);                                               // :616-619
                q[3] = pt; q[4] = bt; q[5] = sqrt(at * at + gt * gt); q[6] = 0;
                kp7_pair[o] = p;
                kp7_flip[o] = (uint8_t)(((flip_s_run + pre_s) > 0 ? 1 : 0) | ((flip_t_run + pre_t) > 0 ? 2 : 0));
            }
            kbase += tot7; flip_s_run += tfs; flip_t_run += tft;
        }
    }
    if (!WRITE && threadIdx.x == 0) { row_cnt[p] = rbase; kp7_cnt[p] = kbase; }
}

// exclusive scan of two small int arrays (one block)
__global__ void scan2_kernel(const int* __restrict__ a, const int* __restrict__ b, int n, int* __restrict__ oa, int* __restrict__ ob)
{
    __shared__ int s_w[4];
    int ba = 0, bb = 0;
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + threadIdx.x;
        const int va = i < n ? a[i] : 0, vb = i < n ? b[i] : 0;
        int ta, tb;
        const int pa = block_scan_excl(va, &ta, s_w);
        const int pb = block_scan_excl(vb, &tb, s_w);
        if (i < n) { oa[i] = ba + pa; ob[i] = bb + pb; }
        ba += ta; bb += tb;
    }
    if (threadIdx.x == 0) { oa[n] = ba; ob[n] = bb; }
}

__global__ void hamming_one_kernel(const uint8_t* a, const uint8_t* b, int* out)
{
    const uint32_t* pa = reinterpret_cast<const uint32_t*>(a);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(b);
    int d = 0;
    for (int i = 0; i < 8; ++i) d += __popc(pa[i] ^ pb[i]);
    *out = d;
}

#define DSSS_FREE0(p) do { hipFree(p); (p) = nullptr; } while (0)

// threshold T with sqrt(d) < radius  <=>  d < T for correctly rounded sqrt
static double gate_threshold(double radius)
{
    double t = radius * radius;
    if (std::sqrt(t) >= radius) { while (std::sqrt(t) >= radius && t > 0) t = std::nextafter(t, 0.0); t = std::nextafter(t, INFINITY); }
    else { while (std::sqrt(t) < radius) t = std::nextafter(t, INFINITY); }
    return t;
}

extern "C" {

int dsss_match_pairs(dsss_ctx* c, const int* src_ids, const int* tgt_ids, int npairs)
{
    if (!c || npairs < 0 || (npairs > 0 && (!src_ids || !tgt_ids))) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    c->npairs = npairs; c->has_lc = false;
    c->pair_s.assign(src_ids, src_ids + npairs); c->pair_t.assign(tgt_ids, tgt_ids + npairs);
    c->pair_active.assign(npairs, -1);
    { int rc = dsss_sync_bboxes(c); if (rc) return rc; }
    std::vector<int> as, at;
    for (int p = 0; p < npairs; ++p) {
        const int s = src_ids[p], t = tgt_ids[p];
        if (s < 0 || s >= c->max_frames || t < 0 || t >= c->max_frames || s == t) DSSS_FAIL(c, DSSS_E_ARG, "pair %d: bad frame ids (%d,%d)", p, s, t);
        const dsss_frame &a = c->frames[s], &b = c->frames[t];
        if (!a.has_feat || !b.has_feat) DSSS_FAIL(c, DSSS_E_STATE, "pair %d: frames %d/%d have no features", p, s, t);
        if (c->mt.use_l2 == 2 && (!a.has_sift || !b.has_sift)) DSSS_FAIL(c, DSSS_E_STATE, "pair %d: use_l2 = 2 needs the 128-element rows of frames %d/%d (dsss_orb_params.descriptor = DSSS_DESC_SIFT128)", p, s, t);
        // a keypoint outside the other frame's geo box is skipped (FEAmatcher.cpp:84), so disjoint boxes match nothing
        const bool disjoint = a.bbox[1] < b.bbox[0] || b.bbox[1] < a.bbox[0] || a.bbox[3] < b.bbox[2] || b.bbox[3] < a.bbox[2];
        if (!disjoint && a.nkp > 0 && b.nkp > 0) { c->pair_active[p] = (int)as.size(); as.push_back(s); at.push_back(t); }
    }
    const int na = (int)as.size();
    c->nactive = na; c->total_rows = 0; c->total_kp7 = 0;
    c->h_row_off.assign(na + 1, 0); c->h_kp7_off.assign(na + 1, 0);
    if (na == 0) return DSSS_OK;
    const size_t K = c->kcap;
    if ((size_t)na > c->match_cap_pairs) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // free + null first and publish the new capacity only after every allocation succeeded, so that a failed
        // hipMalloc leaves the context consistent (capacity 0, null pointers: the next call starts over, dsss_destroy is safe)
        c->match_cap_pairs = 0;
        DSSS_FREE0(c->act_s); DSSS_FREE0(c->act_t); DSSS_FREE0(c->corres_nn); DSSS_FREE0(c->corres);
        DSSS_FREE0(c->scc_hist); DSSS_FREE0(c->scc_count); DSSS_FREE0(c->scc_model);
        DSSS_FREE0(c->row_cnt); DSSS_FREE0(c->kp7_cnt); DSSS_FREE0(c->row_off); DSSS_FREE0(c->kp7_off);
        HIPCHK(c, hipMalloc(&c->act_s, na * sizeof(int))); HIPCHK(c, hipMalloc(&c->act_t, na * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->corres_nn, 2 * (size_t)na * K * sizeof(int32_t)));
        HIPCHK(c, hipMalloc(&c->corres, 2 * (size_t)na * K * sizeof(int32_t)));
        HIPCHK(c, hipMalloc(&c->scc_hist, 2 * na * sizeof(int))); HIPCHK(c, hipMalloc(&c->scc_count, 2 * na * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->scc_model, 2 * na * sizeof(double)));
        HIPCHK(c, hipMalloc(&c->row_cnt, na * sizeof(int))); HIPCHK(c, hipMalloc(&c->kp7_cnt, na * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->row_off, (na + 1) * sizeof(int))); HIPCHK(c, hipMalloc(&c->kp7_off, (na + 1) * sizeof(int)));
        c->match_cap_pairs = na;
    }
    HIPCHK(c, hipMemcpyAsync(c->act_s, as.data(), na * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->act_t, at.data(), na * sizeof(int), hipMemcpyHostToDevice, c->stream));
    // per-frame pointer tables + RNG stream (small, rebuilt per call)
    const int F = c->max_frames;
    std::vector<const double*> hp(3 * (size_t)F, nullptr);
    for (int f = 0; f < F; ++f) { hp[f] = c->frames[f].alt; hp[F + f] = c->frames[f].gr; hp[2 * F + f] = c->frames[f].pose6; }
    const int iters = c->mt.scc_iters;
    std::vector<uint32_t> raw(2 * (size_t)iters);
    { uint64_t st = 0xffffffffu; for (auto& r : raw) { st = (uint64_t)(uint32_t)st * 4164903690U + (uint32_t)(st >> 32); r = (uint32_t)st; } }
    const size_t aux_bytes = hp.size() * sizeof(double*) + raw.size() * sizeof(uint32_t);
    if (c->mt_aux_bytes < aux_bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream)); hipFree(c->mt_aux);
        c->mt_aux = nullptr; c->mt_aux_bytes = 0;
        HIPCHK(c, hipMalloc(&c->mt_aux, aux_bytes)); c->mt_aux_bytes = aux_bytes;
    }
    const double** d_ptrs = (const double**)c->mt_aux;
    c->d_ptrs = d_ptrs;
    uint32_t* d_raw = (uint32_t*)((char*)c->mt_aux + hp.size() * sizeof(double*));
    HIPCHK(c, hipMemcpyAsync(d_ptrs, hp.data(), hp.size() * sizeof(double*), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_raw, raw.data(), raw.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    int max_nkp = 0;
    for (int f = 0; f < F; ++f) if (c->frames[f].has_feat) max_nkp = std::max(max_nkp, c->frames[f].nkp);
    const dim3 grid((max_nkp + MT_TILE - 1) / MT_TILE, 2 * na);
    const double T = gate_threshold(c->mt.radius);
    // geo grid over the frames of the active pairs (match_grid_kernel); the all-pairs kernel stays for a degenerate geometry (a
    // non-finite box or radius, or a box of more than 2^22 cells) and as the A/B switch DSSS_MT_GRID=0
    bool evals_pending = false;
    std::function<hipError_t()> gcount;                // (profile on) counts the grid's evaluations, launched after the matcher's scope has closed
    bool use_grid = !(getenv("DSSS_MT_GRID") && atoi(getenv("DSSS_MT_GRID")) == 0) && std::isfinite(c->mt.radius) && c->mt.radius > 0;
    const double cs = c->mt.radius / MT_SUB * (1.0 + 1.0 / 1048576.0), inv_cs = 1.0 / cs;      // the hair: 1e-6 of a cell against 1e-13 of rounding
    std::vector<mt_grid> gtab; std::vector<int> gframes; size_t gcells = 0;
    if (use_grid) {
        gtab.assign(F, mt_grid{ 0, 0, 0, 0 });
        std::vector<char> seen(F, 0);
        for (int a2 = 0; a2 < na && use_grid; ++a2)
            for (int f : { as[a2], at[a2] }) {
                if (seen[f]) continue;
                seen[f] = 1;
                const double* b = c->frames[f].bbox;
                const double w = (b[1] - b[0]) * inv_cs, h = (b[3] - b[2]) * inv_cs;
                if (!(std::isfinite(b[0]) && std::isfinite(b[2]) && w >= 0 && h >= 0 && (w + 1) * (h + 1) < 4194304.0)) { use_grid = false; break; }
                mt_grid& g = gtab[f];
                g.W = (int)w + 1; g.H = (int)h + 1; g.off = (int)gcells;
                gcells += (size_t)g.W * g.H + 1;
                gframes.push_back(f);
            }
        if (gcells >= (size_t)1 << 30) use_grid = false;
    }
    if (use_grid) {
        if (c->mt_gs_cap != (size_t)F * K) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->mt_gs_cap = 0;
            DSSS_FREE0(c->mt_gs_geo); DSSS_FREE0(c->mt_gs_desc); DSSS_FREE0(c->mt_gs_idx);
            HIPCHK(c, hipMalloc(&c->mt_gs_geo, (size_t)F * K * 2 * sizeof(double)));
            HIPCHK(c, hipMalloc(&c->mt_gs_desc, (size_t)F * K * 32));
            HIPCHK(c, hipMalloc(&c->mt_gs_idx, (size_t)F * K * sizeof(int)));
            c->mt_gs_cap = (size_t)F * K;
        }
        const size_t need = 2 * gcells * sizeof(int) + (size_t)F * sizeof(mt_grid) + (size_t)F * sizeof(int) + 16;
        if (c->mt_cells_bytes < need) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->mt_cells_bytes = 0;
            DSSS_FREE0(c->mt_cells);
            HIPCHK(c, hipMalloc(&c->mt_cells, need + need / 4));
            c->mt_cells_bytes = need + need / 4;
        }
        char* base = (char*)c->mt_cells;
        mt_grid* d_tab = (mt_grid*)base;                                          // [F] (16-byte entries first: alignment)
        int* d_frames = (int*)(base + (size_t)F * sizeof(mt_grid));               // [frames of the active pairs]
        int* d_start = d_frames + F; int* d_cur = d_start + gcells;
        unsigned long long* d_evals = c->prof.on ? (unsigned long long*)(((uintptr_t)(d_cur + gcells) + 7) & ~(uintptr_t)7) : nullptr;
        if (d_evals) HIPCHK(c, hipMemsetAsync(d_evals, 0, sizeof(unsigned long long), c->stream));
        HIPCHK(c, hipMemcpyAsync(d_tab, gtab.data(), (size_t)F * sizeof(mt_grid), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_frames, gframes.data(), gframes.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        double evals = 0;
        for (int a2 = 0; a2 < na; ++a2) evals += 2.0 * c->frames[as[a2]].nkp * c->frames[at[a2]].nkp;
        dsss_scope sc(c, DSSS_K_MATCH, evals, 2);
        const dim3 ggrid((max_nkp + MT_TILE / MT_LPQ - 1) / (MT_TILE / MT_LPQ), 2 * na);
        hipLaunchKernelGGL(mt_grid_build_kernel, dim3((unsigned)gframes.size()), dim3(MTG_THREADS), 0, c->stream, d_frames, d_tab, c->nkp_dev, c->desc, c->geo,
                           c->bbox_dev, (int)K, inv_cs, d_start, d_cur, (double2*)c->mt_gs_geo, (uint4*)c->mt_gs_desc, c->mt_gs_idx);
        if (c->mt.use_l2 == 2)
            hipLaunchKernelGGL(match_grid_kernel<2>, ggrid, dim3(MT_TILE), 0, c->stream, c->act_s, c->act_t, c->nkp_dev, c->bbox_dev, d_tab, d_start,
                               (const double2*)c->mt_gs_geo, (const uint4*)c->desc128, c->mt_gs_idx, (int)K, inv_cs, T, c->mt.bound_same, c->mt.bound_diff,
                               c->mt.l2_bound, c->mt.ratio, c->corres_nn);
        else if (c->mt.use_l2)
            hipLaunchKernelGGL(match_grid_kernel<1>, ggrid, dim3(MT_TILE), 0, c->stream, c->act_s, c->act_t, c->nkp_dev, c->bbox_dev, d_tab, d_start,
                               (const double2*)c->mt_gs_geo, (const uint4*)c->mt_gs_desc, c->mt_gs_idx, (int)K, inv_cs, T, c->mt.bound_same, c->mt.bound_diff,
                               c->mt.l2_bound, c->mt.ratio, c->corres_nn);
        else
            hipLaunchKernelGGL(match_grid_kernel<0>, ggrid, dim3(MT_TILE), 0, c->stream, c->act_s, c->act_t, c->nkp_dev, c->bbox_dev, d_tab, d_start,
                               (const double2*)c->mt_gs_geo, (const uint4*)c->mt_gs_desc, c->mt_gs_idx, (int)K, inv_cs, T, c->mt.bound_same, c->mt.bound_diff,
                               c->mt.l2_bound, c->mt.ratio, c->corres_nn);
        HIPCHK(c, hipGetLastError());
        evals_pending = d_evals != nullptr;
        if (d_evals) { gcount = [=]() {
            hipLaunchKernelGGL(mt_grid_count_kernel, grid, dim3(MT_TILE), 0, c->stream, c->act_s, c->act_t, c->nkp_dev, c->bbox_dev, d_tab, d_start, (const double2*)c->mt_gs_geo, (int)K, inv_cs, d_evals);
            return hipMemcpyAsync(&c->mt_evals_host, d_evals, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream); }; }
    } else {
        // algorithmic work of the matcher: gate + Hamming evaluations = sum over the directed active pairs of Na x Nb
        double evals = 0;
        for (int a2 = 0; a2 < na; ++a2) evals += 2.0 * c->frames[as[a2]].nkp * c->frames[at[a2]].nkp;
        dsss_scope sc(c, DSSS_K_MATCH, evals);
        if (c->prof.on) c->prof.work[DSSS_K_MATCH_DONE] += evals;      // the all-pairs kernel performs every evaluation it is credited with
        if (c->mt.use_l2 == 2)
            hipLaunchKernelGGL(match_nn_kernel<2>, grid, dim3(MT_TILE), 0, c->stream, c->act_s, c->act_t, c->nkp_dev, c->desc128, c->geo,
                               c->bbox_dev, (int)K, T, c->mt.bound_same, c->mt.bound_diff, c->mt.l2_bound, c->mt.ratio, c->corres_nn);
        else if (c->mt.use_l2)
            hipLaunchKernelGGL(match_nn_kernel<1>, grid, dim3(MT_TILE), 0, c->stream, c->act_s, c->act_t, c->nkp_dev, c->desc, c->geo,
                               c->bbox_dev, (int)K, T, c->mt.bound_same, c->mt.bound_diff, c->mt.l2_bound, c->mt.ratio, c->corres_nn);
        else
            hipLaunchKernelGGL(match_nn_kernel<0>, grid, dim3(MT_TILE), 0, c->stream, c->act_s, c->act_t, c->nkp_dev, c->desc, c->geo,
                               c->bbox_dev, (int)K, T, c->mt.bound_same, c->mt.bound_diff, c->mt.l2_bound, c->mt.ratio, c->corres_nn);
        HIPCHK(c, hipGetLastError());
    }
    if (gcount) HIPCHK(c, gcount());
    {
        dsss_scope sc(c, DSSS_K_SCC);
        const size_t sh = K * 8 + (size_t)iters * 4;
        if (sh > 150 * 1024) DSSS_FAIL(c, DSSS_E_CAPACITY, "SCC kernel needs %zu B of LDS", sh);
        hipLaunchKernelGGL(scc_kernel, dim3(2 * na), dim3(256), sh, c->stream, c->act_s, c->act_t, c->nkp_dev, c->rows_dev, c->kps, (int)K,
                           d_raw, iters, c->mt.pix_err, c->corres_nn, c->corres, c->scc_hist, c->scc_count, c->scc_model);
        HIPCHK(c, hipGetLastError());
    }
    {
        dsss_scope sc(c, DSSS_K_ROWS);
        hipLaunchKernelGGL(pair_rows_kernel<false>, dim3(na), dim3(256), 0, c->stream, c->act_s, c->act_t, c->nkp_dev, c->rows_dev, c->cols_dev,
                           c->kps, (int)K, c->corres, c->scc_hist, c->scc_model, c->mt.merge_thr, d_ptrs, d_ptrs + F, d_ptrs + 2 * F,
                           c->row_cnt, c->kp7_cnt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        HIPCHK(c, hipGetLastError());
        hipLaunchKernelGGL(scan2_kernel, dim3(1), dim3(256), 0, c->stream, c->row_cnt, c->kp7_cnt, na, c->row_off, c->kp7_off);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(c->h_row_off.data(), c->row_off, (na + 1) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_kp7_off.data(), c->kp7_off, (na + 1) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (evals_pending) c->prof.work[DSSS_K_MATCH_DONE] += (double)c->mt_evals_host;
    c->total_rows = c->h_row_off[na]; c->total_kp7 = c->h_kp7_off[na];
    if ((size_t)c->total_rows > c->rows_cap) {
        const size_t want = (size_t)c->total_rows + 1024;
        c->rows_cap = 0;
        DSSS_FREE0(c->rows6); DSSS_FREE0(c->kp7); DSSS_FREE0(c->kp7_pair); DSSS_FREE0(c->kp7_flip);
        HIPCHK(c, hipMalloc(&c->rows6, want * 6 * sizeof(double)));
        HIPCHK(c, hipMalloc(&c->kp7, want * 7 * sizeof(double)));
        HIPCHK(c, hipMalloc(&c->kp7_pair, want * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->kp7_flip, want));
        c->rows_cap = want;
    }
    if (c->total_rows > 0) {
        dsss_scope sc(c, DSSS_K_ROWS);
        hipLaunchKernelGGL(pair_rows_kernel<true>, dim3(na), dim3(256), 0, c->stream, c->act_s, c->act_t, c->nkp_dev, c->rows_dev, c->cols_dev,
                           c->kps, (int)K, c->corres, c->scc_hist, c->scc_model, c->mt.merge_thr, d_ptrs, d_ptrs + F, d_ptrs + 2 * F,
                           c->row_cnt, c->kp7_cnt, c->row_off, c->kp7_off, c->rows6, c->kp7, c->kp7_pair, c->kp7_flip);
        HIPCHK(c, hipGetLastError());
    }
    return DSSS_OK;
}

int dsss_match_get_dir(dsss_ctx* c, int pair, int dir, int32_t* corres_nn, int32_t* corres, int cap, int* hist, int* count, double* model)
{
    if (!c) return DSSS_E_ARG;
    if (pair < 0 || pair >= c->npairs || dir < 0 || dir > 1) DSSS_FAIL(c, DSSS_E_ARG, "pair %d / dir %d out of range", pair, dir);
    const int fa = dir ? c->pair_t[pair] : c->pair_s[pair];
    const int n = c->frames[fa].nkp;
    if (cap < n) DSSS_FAIL(c, DSSS_E_CAPACITY, "caller capacity %d < %d", cap, n);
    const int a = c->pair_active[pair];
    if (a < 0) {
        for (int i = 0; i < n; ++i) { if (corres_nn) corres_nn[i] = -1; if (corres) corres[i] = -1; }
        if (hist) *hist = 0; if (count) *count = 0; if (model) *model = 0;
        return DSSS_OK;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t o = (size_t)(2 * a + dir) * c->kcap;
    if (corres_nn && n) HIPCHK(c, hipMemcpy(corres_nn, c->corres_nn + o, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (corres && n) HIPCHK(c, hipMemcpy(corres, c->corres + o, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (hist) HIPCHK(c, hipMemcpy(hist, c->scc_hist + 2 * a + dir, sizeof(int), hipMemcpyDeviceToHost));
    if (count) HIPCHK(c, hipMemcpy(count, c->scc_count + 2 * a + dir, sizeof(int), hipMemcpyDeviceToHost));
    if (model) HIPCHK(c, hipMemcpy(model, c->scc_model + 2 * a + dir, sizeof(double), hipMemcpyDeviceToHost));
    return DSSS_OK;
}

int dsss_match_get_rows(dsss_ctx* c, int pair, double* rows6, int cap, int* nrows)
{
    if (!c) return DSSS_E_ARG;
    if (pair < 0 || pair >= c->npairs) DSSS_FAIL(c, DSSS_E_ARG, "pair %d out of range", pair);
    const int a = c->pair_active[pair];
    const int n = a < 0 ? 0 : c->h_row_off[a + 1] - c->h_row_off[a];
    if (nrows) *nrows = n;
    if (n == 0 || !rows6) return DSSS_OK;
    if (cap < n) DSSS_FAIL(c, DSSS_E_CAPACITY, "caller capacity %d < %d rows", cap, n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(rows6, c->rows6 + (size_t)c->h_row_off[a] * 6, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost));
    return DSSS_OK;
}

int dsss_match_get_kp7(dsss_ctx* c, int pair, double* kp7, int cap, int* nout)
{
    if (!c) return DSSS_E_ARG;
    if (pair < 0 || pair >= c->npairs) DSSS_FAIL(c, DSSS_E_ARG, "pair %d out of range", pair);
    const int a = c->pair_active[pair];
    const int n = a < 0 ? 0 : c->h_kp7_off[a + 1] - c->h_kp7_off[a];
    if (nout) *nout = n;
    if (n == 0 || !kp7) return DSSS_OK;
    if (cap < n) DSSS_FAIL(c, DSSS_E_CAPACITY, "caller capacity %d < %d", cap, n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(kp7, c->kp7 + (size_t)c->h_kp7_off[a] * 7, (size_t)n * 7 * sizeof(double), hipMemcpyDeviceToHost));
    return DSSS_OK;
}

int dsss_match_pair_active(dsss_ctx* c, int pair, int* active)
{
    if (!c || !active) return DSSS_E_ARG;
    if (pair < 0 || pair >= c->npairs) DSSS_FAIL(c, DSSS_E_ARG, "pair %d out of range", pair);
    *active = c->pair_active[pair] >= 0;
    return DSSS_OK;
}

int dsss_match_total(dsss_ctx* c, int* total_rows, int* total_kp7)
{
    if (!c) return DSSS_E_ARG;
    if (total_rows) *total_rows = c->total_rows;
    if (total_kp7) *total_kp7 = c->total_kp7;
    return DSSS_OK;
}

int dsss_descriptor_distance(dsss_ctx* c, int id_a, int ia, int id_b, int ib, int* dist)
{
    if (!c || !dist) return DSSS_E_ARG;
    if (id_a < 0 || id_a >= c->max_frames || id_b < 0 || id_b >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id out of range");
    if (!c->frames[id_a].has_feat || !c->frames[id_b].has_feat || ia < 0 || ib < 0 || ia >= c->frames[id_a].nkp || ib >= c->frames[id_b].nkp)
        DSSS_FAIL(c, DSSS_E_ARG, "descriptor index out of range");
    if (!c->tmp_dev) HIPCHK(c, hipMalloc(&c->tmp_dev, 64 * sizeof(int)));      // once per context, not per call
    int* d_out = c->tmp_dev;
    hipLaunchKernelGGL(hamming_one_kernel, dim3(1), dim3(1), 0, c->stream, c->desc + ((size_t)id_a * c->kcap + ia) * 32,
                       c->desc + ((size_t)id_b * c->kcap + ib) * 32, d_out);
    hipError_t e = hipMemcpyAsync(dist, d_out, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    HIPCHK(c, e);
    return DSSS_OK;
}

} // extern "C"
