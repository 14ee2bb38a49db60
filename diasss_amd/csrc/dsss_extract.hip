// diasss_amd/csrc/dsss_extract.hip -- per-frame preprocessing and ORB extraction on the device (gfx950).
// Restates Frame::GetNormalizeSSS / GetFilteredMask / DetectFeature (/root/reference/src/core/frame.cpp:57-124,
// 167-203) and ORBextractor::operator() in its ORB-descriptor configuration
// (/root/reference/thirdparty/ORBextractor.cpp:77-147,410-479,765-853,1034-1041,1049-1140).
// Everything runs in HIP kernels (the quadtree cull is dsss_quadtree.hip); all of it is integer / fixed
// point / explicitly ordered floating point, so results are bit-exact against oracle/orc_frame.c + orc_orb.c.
#include "dsss_internal.h"
#include "dsss_quadtree.h"
#include "dsss_extract.h"
#include <algorithm>
#include <map>
#include <memory>
#include <chrono>
#include <cstdlib>

int dsss_quadtree_cull(const float* xs, const float* ys, const float* resp, int n,
                       int minX, int maxX, int minY, int maxY, int quota, std::vector<int>& keep);

#define EDGE_T 19
#define HALF_PATCH 15
#define CELL_MAX 66            // FAST window: cell (ceil(width/nCols) < 60) + 6; 37 once a level is >= 900 px wide
#define CELL_STRIDE 68
#define CELL_CAP 1024          // strict 8-neighbour local maxima in 60 x 60 <= 30 x 30

// ------------------------------------------------------------------ K1: mean / min, normalise, mask
// Row sums in the fixed order of oracle/orc_frame.c:fixed_sum (128 strided partials -> lane pairs -> xor
// butterfly) so that 2.5*mean is reproducible bit for bit.  One wave per ping, 16-byte loads.
__device__ inline double wave_fixed_sum_tail(double p0, double p1)
{
    double v = p0 + p1;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) v = v + __shfl_xor(v, k, 64);
    return v;
}

__global__ __launch_bounds__(256) void row_reduce_kernel(const ex_frame* __restrict__ frs)
{
    const ex_frame& f = frs[blockIdx.y];
    const double* __restrict__ raw = f.raw; const int N = f.N, M = f.M;
    double* __restrict__ rowsum = f.rowsum; double* __restrict__ rowmin = f.rowmin;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const double* r = raw + (size_t)row * M;
    double p0 = 0.0, p1 = 0.0, mn = INFINITY;
    for (int j = 2 * lane; j < M; j += 128) {
        if (j + 1 < M) {
            const double2 v = *reinterpret_cast<const double2*>(r + j);
            p0 += v.x; p1 += v.y;
            mn = v.x < mn ? v.x : mn; mn = v.y < mn ? v.y : mn;
        } else { const double a = r[j]; p0 += a; mn = a < mn ? a : mn; }
    }
    const double s = wave_fixed_sum_tail(p0, p1);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { const double o = __shfl_xor(mn, k, 64); mn = o < mn ? o : mn; }
    if (lane == 0) { rowsum[row] = s; rowmin[row] = mn; }
}

// stats[0] = mean, stats[1] = min, stats[2] = 2.5*mean (normalisation ceiling), stats[3] = mask threshold
__global__ __launch_bounds__(64) void final_reduce_kernel(const ex_frame* __restrict__ frs, double mask_factor)
{
    const ex_frame& f = frs[blockIdx.x];
    const double* __restrict__ rowsum = f.rowsum; const double* __restrict__ rowmin = f.rowmin; double* __restrict__ stats = f.stats;
    const int N = f.N, M = f.M;
    const int lane = threadIdx.x;
    double p0 = 0.0, p1 = 0.0, mn = INFINITY;
    for (int j = 2 * lane; j < N; j += 128) {
        p0 += rowsum[j]; mn = rowmin[j] < mn ? rowmin[j] : mn;
        if (j + 1 < N) { p1 += rowsum[j + 1]; mn = rowmin[j + 1] < mn ? rowmin[j + 1] : mn; }
    }
    const double tot = wave_fixed_sum_tail(p0, p1);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { const double o = __shfl_xor(mn, k, 64); mn = o < mn ? o : mn; }
    if (lane == 0) {
        const double mean = tot / ((double)N * (double)M);
        stats[0] = mean; stats[1] = mn; stats[2] = mean * 2.5; stats[3] = mean * mask_factor;
    }
}

// static part of Frame::GetFilteredMask (frame.cpp:104-112); sixteen bytes per thread, one 16-byte store (the mask buffer is
// 256-byte aligned and the flat index of a group is a multiple of 16; a group may wrap into the next ping)
__global__ __launch_bounds__(256) void mask_init_kernel(const ex_frame* __restrict__ frs, int width, int side, double sidec)
{
    const ex_frame& f = frs[blockIdx.y];
    uint8_t* __restrict__ mask = f.mask; const int N = f.N, M = f.M;
    const size_t total = (size_t)N * M;
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i0 >= total) return;
    int r = (int)(i0 / M), c = (int)(i0 - (size_t)r * M);
    uint32_t w[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const bool off = (c > M / 2 - width && c < M / 2 + width) || (r < side || r > N - side) || ((double)c < sidec || (double)c > (double)M - sidec);
        if (!off) w[u >> 2] |= 255u << (8 * (u & 3));
        if (++c == M) { c = 0; ++r; }
    }
    if (i0 + 16 <= total) *reinterpret_cast<uint4*>(mask + i0) = make_uint4(w[0], w[1], w[2], w[3]);
    else for (size_t u = 0; i0 + u < total; ++u) mask[i0 + u] = (uint8_t)(w[u >> 2] >> (8 * (u & 3)));
}

// Frame::GetNormalizeSSS (frame.cpp:67-78) + the hot-pixel eraser of GetFilteredMask (:98-103).
// 4 pixels per thread: 2 x 16-byte loads, one 4-byte store.  Erasures only ever write 0, so the scatter is race free.
__global__ __launch_bounds__(256) void normalize_kernel(const ex_frame* __restrict__ frs, int er)
{
    const ex_frame& f = frs[blockIdx.y];
    const double* __restrict__ raw = f.raw; const int N = f.N, M = f.M;
    const double* __restrict__ stats = f.stats; uint8_t* __restrict__ norm = f.lvl[0]; uint8_t* __restrict__ mask = f.mask;
    const size_t total = (size_t)N * M;
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= total) return;
    const double mn = stats[1], den = stats[2] - stats[1], thr = stats[3];
    double v[4];
    const int cnt = (int)((total - i0) < 4 ? (total - i0) : 4);
    if (cnt == 4 && ((reinterpret_cast<uintptr_t>(raw + i0) & 15) == 0)) {
        const double2 a = *reinterpret_cast<const double2*>(raw + i0), b = *reinterpret_cast<const double2*>(raw + i0 + 2);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else for (int k = 0; k < 4; ++k) v[k] = k < cnt ? raw[i0 + k] : 0.0;
    uint8_t q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double o = (v[k] - mn) / den * 255.0;
        if (o > 255.0) o = 255.0;
        int r = (int)rint(o);                       // convertTo(CV_8U): saturate_cast<uchar>(cvRound())
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        q[k] = (uint8_t)r;
        if (k < cnt && v[k] > thr) {
            const int pi = (int)((i0 + k) / M), pj = (int)((i0 + k) % M);
            if (pi - er >= 0 && pj - er >= 0)       // the reference's size_t loops do not run otherwise
                for (int x = pi - er; x < pi + er && x < N; ++x)
                    for (int y = pj - er; y < pj + er && y < M; ++y) mask[(size_t)x * M + y] = 0;
        }
    }
    if (cnt == 4 && ((i0 & 3) == 0)) *reinterpret_cast<uchar4*>(norm + i0) = make_uchar4(q[0], q[1], q[2], q[3]);
    else for (int k = 0; k < cnt; ++k) norm[i0 + k] = q[k];
}

// ------------------------------------------------------------------ K2: cv::resize INTER_LINEAR, CV_8UC1
// (ORBextractor.cpp:1128).  11-bit fixed point exactly as OpenCV's scalar path: see oracle/orc_orb.c.
__device__ inline int cvfloorf_dev(float v) { const int i = (int)v; return i - (v < (float)i); }

// cv::resize(INTER_LINEAR) of one pyramid level from the previous one, OpenCV's scalar u8 path: 11-bit fixed-point weights per
// output column / row (tables built once per geometry by resize_tables(), the same float arithmetic OpenCV uses), horizontal pass
// in int, vertical pass with the two 16-bit shifts.  A thread produces FOUR consecutive bytes of the level (flat index, so the
// 4-byte store is always aligned, a group may wrap into the next row) -- the one-byte-per-thread form stored at 7 % of HBM speed.
typedef uint16_t u16_unaligned __attribute__((aligned(1)));
// RS_K groups of four pixels per thread, 1024 pixels apart, written as four straight-line phases (indices, tables, pixels,
// arithmetic): with one group per thread the kernel waited three dependent memory latencies (frame record, tables, pixels)
// for 256 pixels per wavefront and ran at a ninth of the memory rate; the groups of a thread now wait together.  A group that
// runs over the end of a row takes its last pixels from the next one by selects, not by a branch.
#define RS_K 4
#if defined(__HIP_DEVICE_COMPILE__)
#define DSSS_GLOBAL __attribute__((address_space(1)))     // device pass: global_load with a scalar base instead of flat_load
#else
#define DSSS_GLOBAL                                       // host pass of the same source: the qualifier means nothing there
#endif
typedef unsigned short rs_u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t rs_u32x3 __attribute__((ext_vector_type(3)));
__global__ __launch_bounds__(256) void resize_kernel(const ex_frame* __restrict__ frs, int level, int rest_only)
{
    const ex_frame& f = frs[blockIdx.y];
    if (level >= f.nlevels) return;
    if (rest_only && f.xt[level][f.cols[level] + 4].sx != 0) return;      // resize_strip_kernel has done this frame
    // pointers read from the frame record are generic to the compiler (flat loads, 64-bit address arithmetic per access): say
    // they are global, and index them with 32-bit offsets from the wave-uniform base
    const DSSS_GLOBAL uint8_t* __restrict__ src = (const DSSS_GLOBAL uint8_t*)f.lvl[level - 1]; DSSS_GLOBAL uint8_t* __restrict__ dst = (DSSS_GLOBAL uint8_t*)f.lvl[level];
    const int sw = f.cols[level - 1], dh = f.rows[level], dw = f.cols[level];
    const DSSS_GLOBAL resize_xtab* __restrict__ xt = (const DSSS_GLOBAL resize_xtab*)f.xt[level]; const DSSS_GLOBAL resize_ytab* __restrict__ yt = (const DSSS_GLOBAL resize_ytab*)f.yt[level];
    const uint32_t total = (uint32_t)dw * (uint32_t)dh;                   // < 2^31: the host refuses larger frames
    const uint32_t gb = 4u * (blockIdx.x * (256u * RS_K) + threadIdx.x);
    if (gb >= total) return;
    if (dw < 8) {                                                         // degenerate levels: one pixel at a time
        for (int k = 0; k < RS_K; ++k)
            for (uint32_t g = gb + 1024u * k; g < gb + 1024u * k + 4 && g < total; ++g) {
                const int dy = (int)(g / (uint32_t)dw), dx = (int)(g - (uint32_t)dy * dw);
                const resize_xtab X = xt[dx]; const resize_ytab Y = yt[dy];
                const DSSS_GLOBAL uint8_t* S0 = src + (size_t)Y.ya * sw; const DSSS_GLOBAL uint8_t* S1 = src + (size_t)Y.yb * sw;
                int r0, r1;
                if (X.a1) { r0 = S0[X.sx] * X.a0 + S0[X.sx + 1] * X.a1; r1 = S1[X.sx] * X.a0 + S1[X.sx + 1] * X.a1; }
                else { r0 = S0[X.sx] * X.a0; r1 = S1[X.sx] * X.a0; }
                dst[g] = (uint8_t)((((Y.b0 * (r0 >> 4)) >> 16) + ((Y.b1 * (r1 >> 4)) >> 16) + 2) >> 2);
            }
        return;
    }
    const double inv = 1.0 / (double)dw;                                  // quotient by reciprocal (exact to one unit below 2^31), one correction
    uint32_t g0[RS_K], rr[RS_K]; uint32_t q0[RS_K], q1[RS_K];
#pragma unroll
    for (int k = 0; k < RS_K; ++k) {
        const uint32_t g = gb + 1024u * k;
        g0[k] = g;
        const uint32_t gv = g < total ? g : gb;                           // a group past the end recomputes group 0 and stores nothing
        int q = (int)((double)gv * inv), r = (int)gv - (int)__umul24((uint32_t)q, (uint32_t)dw);      // (rows and columns < 65536)
        const int adj = r < 0 ? -1 : (r >= dw ? 1 : 0);
        q += adj; r -= adj < 0 ? -dw : (adj > 0 ? dw : 0);
        rr[k] = (uint32_t)r; q0[k] = (uint32_t)q; q1[k] = (uint32_t)(q + 1 < dh ? q + 1 : dh - 1);
    }
    // tables: xt[r .. r + 3] in two 16-byte loads (the table carries its first entries again behind the last column)
    resize_ytab Y0[RS_K]; uint32_t xs[RS_K][4], xw[RS_K][4];
#pragma unroll
    for (int k = 0; k < RS_K; ++k) {
        Y0[k] = *reinterpret_cast<const DSSS_GLOBAL resize_ytab*>(reinterpret_cast<const DSSS_GLOBAL char*>(yt) + q0[k] * (uint32_t)sizeof(resize_ytab));
        const uint4 xa = *reinterpret_cast<const DSSS_GLOBAL uint4*>(reinterpret_cast<const DSSS_GLOBAL char*>(xt) + rr[k] * 8u);
        const uint4 xb = *reinterpret_cast<const DSSS_GLOBAL uint4*>(reinterpret_cast<const DSSS_GLOBAL char*>(xt) + rr[k] * 8u + 16u);
        xs[k][0] = xa.x; xw[k][0] = xa.y; xs[k][1] = xa.z; xw[k][1] = xa.w; xs[k][2] = xb.x; xw[k][2] = xb.y; xs[k][3] = xb.z; xw[k][3] = xb.w;
    }
    const bool windowed = xt[dw + 4].sx != 0;
    uint32_t outv[RS_K];
    if (windowed) {
        // The 14 loads per group of the first form (8 unaligned 16-bit pixel pairs) made the kernel wait for the texture
        // addresser (one wave-load per ~20 cycles).  The source bytes of the four pixels of a group lie within six consecutive
        // bytes of each of the two source rows: ONE aligned 12-byte load per row, the pairs come out of it by v_perm.
        uint32_t W0[RS_K][3], W1[RS_K][3], m0[RS_K], m1[RS_K];
#pragma unroll
        for (int k = 0; k < RS_K; ++k) {
            const uint32_t b0 = __umul24((uint32_t)Y0[k].ya, (uint32_t)sw) + xs[k][0], b1 = __umul24((uint32_t)Y0[k].yb, (uint32_t)sw) + xs[k][0];
            m0[k] = b0 & 3u; m1[k] = b1 & 3u;                            // the level base is 256-byte aligned
            const rs_u32x3 w0 = *reinterpret_cast<const DSSS_GLOBAL rs_u32x3*>(src + (b0 & ~3u));
            const rs_u32x3 w1 = *reinterpret_cast<const DSSS_GLOBAL rs_u32x3*>(src + (b1 & ~3u));
            W0[k][0] = w0.x; W0[k][1] = w0.y; W0[k][2] = w0.z; W1[k][0] = w1.x; W1[k][1] = w1.y; W1[k][2] = w1.z;
        }
#pragma unroll
        for (int k = 0; k < RS_K; ++k) {
            uint32_t out = 0;
            const uint32_t b0 = (uint32_t)Y0[k].b0, b1 = (uint32_t)Y0[k].b1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t d = xs[k][u] - xs[k][0];                   // 0 .. 4
                const uint32_t o0 = m0[k] + d, o1 = m1[k] + d;            // byte index of the left pixel in the 12-byte window: 0 .. 7
                const uint32_t lo0 = o0 & 4u ? W0[k][1] : W0[k][0], hi0 = o0 & 4u ? W0[k][2] : W0[k][1];
                const uint32_t lo1 = o1 & 4u ? W1[k][1] : W1[k][0], hi1 = o1 & 4u ? W1[k][2] : W1[k][1];
                const rs_u16x2 w = __builtin_bit_cast(rs_u16x2, xw[k][u]);
                const uint32_t r0 = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, __builtin_amdgcn_perm(hi0, lo0, 0x0c010c00u + (o0 & 3u) * 0x00010001u)), w, 0u, false);
                const uint32_t r1 = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, __builtin_amdgcn_perm(hi1, lo1, 0x0c010c00u + (o1 & 3u) * 0x00010001u)), w, 0u, false);
                const uint32_t v = (((__umul24(b0, r0 >> 4) >> 16) + (__umul24(b1, r1 >> 4) >> 16) + 2u) >> 2) & 255u;        // (weights <= 2048, row sums >> 4 < 2^16: 24-bit multiplies, full rate)
                out |= v << (8 * u);
            }
            outv[k] = out;
        }
    }
    // groups that run over the end of a row (one in a row), and every group of a geometry the windows do not fit: pixel by pixel
#pragma unroll
    for (int k = 0; k < RS_K; ++k) {
        if (!windowed || rr[k] + 3u >= (uint32_t)dw) {
            const resize_ytab Y1k = *reinterpret_cast<const DSSS_GLOBAL resize_ytab*>(reinterpret_cast<const DSSS_GLOBAL char*>(yt) + q1[k] * (uint32_t)sizeof(resize_ytab));
            const uint32_t A0 = __umul24((uint32_t)Y0[k].ya, (uint32_t)sw), A1 = __umul24((uint32_t)Y0[k].yb, (uint32_t)sw);      // rows, columns < 65536
            const uint32_t B0 = __umul24((uint32_t)Y1k.ya, (uint32_t)sw), B1 = __umul24((uint32_t)Y1k.yb, (uint32_t)sw);
            uint32_t out = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool wr = rr[k] + (uint32_t)u >= (uint32_t)dw;
                const uint32_t p0 = *reinterpret_cast<const DSSS_GLOBAL u16_unaligned*>(src + ((wr ? B0 : A0) + xs[k][u]));
                const uint32_t p1 = *reinterpret_cast<const DSSS_GLOBAL u16_unaligned*>(src + ((wr ? B1 : A1) + xs[k][u]));
                const uint32_t b0 = (uint32_t)(wr ? Y1k.b0 : Y0[k].b0), b1 = (uint32_t)(wr ? Y1k.b1 : Y0[k].b1);
                const rs_u16x2 w = __builtin_bit_cast(rs_u16x2, xw[k][u]);
                const uint32_t r0 = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, __builtin_amdgcn_perm(0u, p0, 0x0c010c00u)), w, 0u, false);
                const uint32_t r1 = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, __builtin_amdgcn_perm(0u, p1, 0x0c010c00u)), w, 0u, false);
                const uint32_t v = (((__umul24(b0, r0 >> 4) >> 16) + (__umul24(b1, r1 >> 4) >> 16) + 2u) >> 2) & 255u;        // (weights <= 2048, row sums >> 4 < 2^16: 24-bit multiplies, full rate)
                out |= v << (8 * u);
            }
            outv[k] = out;
        }
    }
#pragma unroll
    for (int k = 0; k < RS_K; ++k) {
        if (g0[k] + 3 < total) *reinterpret_cast<DSSS_GLOBAL uint32_t*>(dst + g0[k]) = outv[k];
        else for (int u = 0; u < 4; ++u) if (g0[k] + u < total) dst[g0[k] + u] = (uint8_t)(outv[k] >> (8 * u));
    }
}

// The same resize BY COLUMN STRIPS (round 5), for the geometries whose windows fit (every level of a pyramid with scale < 4/3): a
// thread owns FOUR output columns and walks RS_R output rows.  What depends on the columns alone -- the four table entries, the
// offsets d of the left source pixels from the first one -- is fetched and derived once per thread instead of once per group; per
// row the 12-byte window of each of the two source rows is shifted to the first source pixel by v_alignbyte (the alignment of
// ya * sw + sx is the row's), after which the pixel pairs come out by v_perm with selectors that are constants of the thread.
// 65 vector instructions per four pixels where the flat form (division of the flat index, tables and selectors per group) took
// 147; all loads of the RS_R rows are issued before the arithmetic.  Rows are sw bytes apart, not padded: the four output bytes
// are one store at whatever alignment dy * dw + 4 g has.
#define RS_R 8
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
__global__ __launch_bounds__(256) void resize_strip_kernel(const ex_frame* __restrict__ frs, int level)
{
    const ex_frame& f = frs[blockIdx.y];
    if (level >= f.nlevels) return;
    const DSSS_GLOBAL uint8_t* __restrict__ src = (const DSSS_GLOBAL uint8_t*)f.lvl[level - 1]; DSSS_GLOBAL uint8_t* __restrict__ dst = (DSSS_GLOBAL uint8_t*)f.lvl[level];
    const int sw = f.cols[level - 1], dh = f.rows[level], dw = f.cols[level];
    const DSSS_GLOBAL resize_xtab* __restrict__ xt = (const DSSS_GLOBAL resize_xtab*)f.xt[level]; const DSSS_GLOBAL resize_ytab* __restrict__ yt = (const DSSS_GLOBAL resize_ytab*)f.yt[level];
    if (xt[dw + 4].sx == 0) return;                                       // windows do not fit: resize_kernel(rest_only) does this frame
    const uint32_t G = (uint32_t)(dw + 3) >> 2, nch = (uint32_t)(dh + RS_R - 1) / RS_R;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= G * nch) return;
    uint32_t ch = (uint32_t)((double)t * (1.0 / (double)G));              // quotient by reciprocal, one correction (t < 2^29)
    int g = (int)t - (int)(ch * G);
    if (g < 0) { --ch; g += (int)G; } else if (g >= (int)G) { ++ch; g -= (int)G; }
    // columns: xt[4 g .. 4 g + 3] (the table carries four more entries behind the last column)
    const uint4 xa = *reinterpret_cast<const DSSS_GLOBAL uint4*>(reinterpret_cast<const DSSS_GLOBAL char*>(xt) + (uint32_t)g * 32u);
    const uint4 xb = *reinterpret_cast<const DSSS_GLOBAL uint4*>(reinterpret_cast<const DSSS_GLOBAL char*>(xt) + (uint32_t)g * 32u + 16u);
    const uint32_t xs0 = xa.x;
    const uint32_t xw[4] = { xa.y, xa.w, xb.y, xb.w };
    uint32_t sel[4];
    {
        const uint32_t sx[4] = { xa.x, xa.z, xb.x, xb.z };
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t d = sx[u] - xs0;                                     // 0 .. 4 (a column past the end of the row: anything, never stored)
            d = d > 4u ? 0u : d;
            sel[u] = 0x0c010c00u + d * 0x00010001u;                       // { 0, byte d + 1, 0, byte d } of the shifted window
        }
    }
    const uint32_t dy0 = ch * RS_R;
    resize_ytab Y[RS_R];
#pragma unroll
    for (int r = 0; r < RS_R; ++r) {
        const uint32_t dy = dy0 + r < (uint32_t)dh ? dy0 + r : (uint32_t)dh - 1u;
        Y[r] = *reinterpret_cast<const DSSS_GLOBAL resize_ytab*>(reinterpret_cast<const DSSS_GLOBAL char*>(yt) + dy * (uint32_t)sizeof(resize_ytab));
    }
    rs_u32x3 WA[RS_R], WB[RS_R]; uint32_t ma[RS_R], mb[RS_R];
#pragma unroll
    for (int r = 0; r < RS_R; ++r) {
        const uint32_t ba = __umul24((uint32_t)Y[r].ya, (uint32_t)sw) + xs0, bb = __umul24((uint32_t)Y[r].yb, (uint32_t)sw) + xs0;
        ma[r] = ba & 3u; mb[r] = bb & 3u;                                 // the level base is 256-byte aligned
        WA[r] = *reinterpret_cast<const DSSS_GLOBAL rs_u32x3*>(src + (ba & ~3u));
        WB[r] = *reinterpret_cast<const DSSS_GLOBAL rs_u32x3*>(src + (bb & ~3u));
    }
    const bool full = 4 * g + 3 < dw;
#pragma unroll
    for (int r = 0; r < RS_R; ++r) {
        const uint32_t a0 = __builtin_amdgcn_alignbyte(WA[r].y, WA[r].x, ma[r]), a1 = __builtin_amdgcn_alignbyte(WA[r].z, WA[r].y, ma[r]);
        const uint32_t c0 = __builtin_amdgcn_alignbyte(WB[r].y, WB[r].x, mb[r]), c1 = __builtin_amdgcn_alignbyte(WB[r].z, WB[r].y, mb[r]);
        // (b (r >> 4)) >> 16 = the high word of the 24-bit product (b << 12) (r & ~15): weights <= 2^11, row sums < 2^19
        const uint64_t b0 = (uint64_t)(((uint32_t)Y[r].b0 << 12) & 0xffffffu), b1 = (uint64_t)(((uint32_t)Y[r].b1 << 12) & 0xffffffu);
        uint32_t out = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const rs_u16x2 w = __builtin_bit_cast(rs_u16x2, xw[u]);
            const uint32_t r0 = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, __builtin_amdgcn_perm(a1, a0, sel[u])), w, 0u, false);
            const uint32_t r1 = __builtin_amdgcn_udot2(__builtin_bit_cast(rs_u16x2, __builtin_amdgcn_perm(c1, c0, sel[u])), w, 0u, false);
            const uint32_t h0 = (uint32_t)((b0 * (uint64_t)(r0 & 0xfffff0u)) >> 32), h1 = (uint32_t)((b1 * (uint64_t)(r1 & 0xfffff0u)) >> 32);
            const uint32_t v = ((h0 + h1 + 2u) >> 2) & 255u;
            out |= v << (8 * u);
        }
        if (dy0 + r < (uint32_t)dh) {
            DSSS_GLOBAL uint8_t* o = dst + ((dy0 + r) * (uint32_t)dw + 4u * (uint32_t)g);
            if (full) *reinterpret_cast<DSSS_GLOBAL u32_unaligned*>(o) = out;
            else for (int u = 0; u < 4; ++u) if (4 * g + u < dw) o[u] = (uint8_t)(out >> (8 * u));
        }
    }
}

// ------------------------------------------------------------------ K3: cv::FAST 9/16 per 30-px cell
struct level_tab { const uint8_t* img[DSSS_MAX_LEVELS]; int cols[DSSS_MAX_LEVELS]; };


// arc value A = max over the 16 arcs of 9 contiguous ring pixels of max(min(v - ring), min(ring - v));
// corner at threshold t iff A > t, cornerScore = A - 1.  Values <= tmin are reported as 0 (never a corner, and never
// able to suppress one).  (On speckled sonar imagery four pixels in five fail the usual opposite-pair early exit at
// minThFAST, so there is none.)
// FOUR PIXELS OF A ROW PER LANE (round 3; one pixel per lane with ring positions k and k + 8 sharing a register cost 139 vector
// instructions per pixel, and the kernel sits at the issue ceiling of its half-rate packed-16 instruction mix):
//   * the 7 x 10 patch of the four pixels comes out of LDS as 7 x 3 ALIGNED dwords (the group starts at a window column that is a
//     multiple of four) instead of 4 x 17 byte loads;
//   * two horizontally adjacent pixels share a register (16-bit halves): ring position j of the pair is two ADJACENT bytes of one
//     patch row, spread into the halves by ONE v_perm_b32 with a compile-time selector; all sixteen positions stay separate
//     registers, so an arc that wraps past position 15 is index arithmetic, not a half swap;
//   * the arcs that start at k and k + 1 (k even) share the window [k + 1, k + 8]: eight windows of 8 by doubling over the ODD
//     positions (2 x 24 packed min / max), and  max(min(W, d[k]), min(W, d[k + 9])) = min(W, max(d[k], d[k + 9]))  closes both
//     arcs with three instructions per polarity.
// 65 vector instructions per pixel; min and max are exact, so the value is the same whatever the order: bit-identical output.
typedef short fast_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fast_v2 fast_min3(fast_v2 a, fast_v2 b, fast_v2 c)
{
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, a)), "v"(__builtin_bit_cast(uint32_t, b)), "v"(__builtin_bit_cast(uint32_t, c)));
    return __builtin_bit_cast(fast_v2, r);
}
__device__ __forceinline__ fast_v2 fast_max3(fast_v2 a, fast_v2 b, fast_v2 c)
{
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, a)), "v"(__builtin_bit_cast(uint32_t, b)), "v"(__builtin_bit_cast(uint32_t, c)));
    return __builtin_bit_cast(fast_v2, r);
}
template <int stride>
__device__ inline uint32_t fast_arc4(const uint8_t* __restrict__ w, int xg, int y, int tmin)      // pixels (3 + xg + k, y), k = 0..3; xg a multiple of 4
{
    constexpr int rdx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
    constexpr int rdy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
    const uint32_t* __restrict__ base = reinterpret_cast<const uint32_t*>(w + (y - 3) * stride + xg);      // stride is a multiple of 4, the slice 16-byte aligned
    uint32_t R[7][3];
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) R[r][q] = base[r * (stride / 4) + q];
    uint32_t out = 0;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {                 // pixel pair (2 pr, 2 pr + 1)
        // bytes c0 and c0 + 1 of patch row r, zero-extended into the two halves
        auto pair_at = [&](int r, int c0) -> fast_v2 {
            const bool lowq = c0 + 1 <= 7;
            const uint32_t lo = lowq ? R[r][0] : R[r][1], hi = lowq ? R[r][1] : R[r][2];
            const uint32_t idx = (uint32_t)(lowq ? c0 : c0 - 4);
            return __builtin_bit_cast(fast_v2, __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | idx | ((idx + 1u) << 16)));
        };
        // (the arcs are taken over the ring VALUES: max over arcs of min (v - ring) = v - min over arcs of max ring, and the dark side
        // likewise -- two subtractions per pair instead of sixteen)
        const fast_v2 vv = pair_at(3, 2 * pr + 3);
        fast_v2 d[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = pair_at(rdy[j] + 3, 2 * pr + rdx[j] + 3);
        fast_v2 lo2[8], hi2[8], lo4[8], hi4[8];      // windows starting at the odd positions 2 i + 1
#pragma unroll
        for (int i = 0; i < 8; ++i) { lo2[i] = __builtin_elementwise_min(d[2 * i + 1], d[(2 * i + 2) & 15]); hi2[i] = __builtin_elementwise_max(d[2 * i + 1], d[(2 * i + 2) & 15]); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { lo4[i] = __builtin_elementwise_min(lo2[i], lo2[(i + 1) & 7]); hi4[i] = __builtin_elementwise_max(hi2[i], hi2[(i + 1) & 7]); }
        // (round 5) THREE-operand packed minima / maxima: gfx950 has v_pk_minimum3_f16 / v_pk_maximum3_f16, they issue at the rate of the
        // two-operand packed forms (tools/ubench/pk_f16.hip: 508 - 517 against 489 - 515 G wave-instructions/s), and on halves that hold
        // 0 .. 255 -- f16 denormals n x 2^-24, compared as the integers they are, returned unchanged -- they ARE the integer minimum and
        // maximum (checked on all 2^24 byte triples in both halves).  The window of eight and its closing element fold into one
        // instruction, the eight arc pairs reduce three at a time: 72 packed instructions per pixel pair where there were 96.
        fast_v2 tlo[8], thi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {                // k = 2 i: window [k + 1, k + 8], closed by d[k] and by d[k + 9]
            const fast_v2 e0 = d[2 * i], e1 = d[(2 * i + 9) & 15];
            tlo[i] = fast_min3(lo4[i], lo4[(i + 2) & 7], __builtin_elementwise_max(e0, e1));
            thi[i] = fast_max3(hi4[i], hi4[(i + 2) & 7], __builtin_elementwise_min(e0, e1));
        }
        // largest arc minimum (dark side), smallest arc maximum (bright side) of the ring
        const fast_v2 bmin = __builtin_elementwise_max(fast_max3(fast_max3(tlo[0], tlo[1], tlo[2]), fast_max3(tlo[3], tlo[4], tlo[5]), tlo[6]), tlo[7]);
        const fast_v2 bmax = __builtin_elementwise_min(fast_min3(fast_min3(thi[0], thi[1], thi[2]), fast_min3(thi[3], thi[4], thi[5]), thi[6]), thi[7]);
        const fast_v2 best = __builtin_elementwise_max(vv - bmax, bmin - vv);
        const int b0 = best.x, b1 = best.y;
        out |= (uint32_t)(b0 > tmin ? b0 : 0) << (16 * pr);
        out |= (uint32_t)(b1 > tmin ? b1 : 0) << (16 * pr + 8);
    }
    return out;
}

__device__ inline int block_scan_excl256(int v, int* total, int* s_w)
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
    for (int k = 0; k < 4; ++k) { const int t = s_w[k]; if (k < w) base += t; tot += t; }
    *total = tot;
    return base + inc - v;
}

// ONE WAVEFRONT per cell window (ORBextractor.cpp:789-816), four cells per workgroup: FAST at iniThFAST, retried at
// minThFAST if the cell is empty, non-max suppression inside the window only, keypoints emitted row-major.  The wavefront
// owns a slice of LDS (window + arc values) and nothing of it is shared, so there is no workgroup barrier and no block scan:
// lanes talk through the LDS in program order and the compaction is a ballot + popcount.  (With one workgroup per cell the
// kernel was bound by the per-cell latency chain -- byte loads, three barriers, four block scans -- at 8 ms per 200 frames.)
#define FAST_WAVE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
template <int stride>      // LDS row stride of the window: a compile-time constant, so that the ring offsets are immediates of the LDS loads
__global__ __launch_bounds__(256) void fast_cells_kernel(const ex_frame* __restrict__ frs, int lev_lo, int lev_hi, int ini_th, int min_th, int wave_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t fast_lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const ex_frame& f = frs[blockIdx.y];
    const int cell = f.cell_begin[lev_lo] + blockIdx.x * 4 + wv;     // one launch per group of pyramid levels [lev_lo, lev_hi): a level's quadtree starts while the FAST of the levels above runs
    if (cell >= f.cell_begin[lev_hi]) return;                        // (cell_begin[l] = ncells for l past the frame's levels)
    uint8_t* __restrict__ win = fast_lds + (size_t)wv * wave_bytes;
    uint8_t* __restrict__ A = win + wave_bytes / 2;
    uint32_t* __restrict__ cand = f.cand; const int cell_cap = f.cell_cap;
    const fast_cell c = f.cells[cell];
    const uint8_t* __restrict__ img = f.lvl[c.level];
    const int cols = f.cols[c.level];
    // The window comes in as UNALIGNED dwords, one per (row, four window columns), written to LDS as dwords (the LDS stride is a
    // multiple of four and the slice is 16-byte aligned), six in flight per lane.  The bytes a dword carries beyond c.w land in
    // columns no evaluated pixel reads; only a dword that would run over the end of the level image is assembled from bytes.
    // (Aligned dwords scattered byte by byte cost 320 vector instructions per cell, an eighth of the kernel; one byte per load 840.)
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    {
        for (int t = lane; t < wave_bytes / 8; t += 64) reinterpret_cast<uint32_t*>(A)[t] = 0u;       // the arc values start at zero (borders stay zero)
        const int dpr = (c.w + 3) >> 2;                              // dwords per window row
        const unsigned mg = (65536u + (unsigned)dpr - 1u) / (unsigned)dpr;      // t / dpr = (t * mg) >> 16, exact for t * dpr < 65536
        const int items = c.h * dpr;
        const uint8_t* __restrict__ corner = img + (size_t)c.y0 * cols + c.x0;
        const uint8_t* __restrict__ img_end = img + (size_t)f.rows[c.level] * cols;
        for (int t0 = 0; t0 < items; t0 += 64 * 6) {
            uint32_t v[6]; int off[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int t = t0 + 64 * u + lane;
                const int r = (int)(__umul24((unsigned)t, mg) >> 16), j = t - (int)__umul24((unsigned)dpr, (unsigned)r);      // (24-bit multiplies are full rate, 32-bit ones a quarter: every operand here is below 2^21)
                const uint8_t* src = corner + (size_t)__umul24((unsigned)r, (unsigned)cols) + 4 * j;
                off[u] = t < items ? r * stride + 4 * j : -1;
                v[u] = 0u;
                if (t < items) {
                    if (src + 4 <= img_end) v[u] = *reinterpret_cast<const u32_unaligned*>(src);
                    else for (int b = 0; b < 4; ++b) if (src + b < img_end) v[u] |= (uint32_t)src[b] << (8 * b);
                }
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) if (off[u] >= 0) *reinterpret_cast<uint32_t*>(win + off[u]) = v[u];
        }
    }
    FAST_WAVE_SYNC();
    const int ew = c.w - 6, eh = c.h - 6;
    const int ne = (ew > 0 && eh > 0) ? ew * eh : 0;
    const int tmin = ini_th < min_th ? ini_th : min_th;
    const int G = (ew + 3) >> 2, ngroups = ne > 0 ? eh * G : 0;      // groups of four pixels of a row, the unit of both passes below
    const unsigned mg_g = ((1u << 20) + (unsigned)(G > 0 ? G : 1) - 1u) / (unsigned)(G > 0 ? G : 1);
    for (int g = lane; g < ngroups; g += 64) {
        const int gy = (int)(__umul24((unsigned)g, mg_g) >> 20), xg = 4 * (g - (int)__umul24((unsigned)gy, (unsigned)G));
        const int y = 3 + gy;
        // (a group that runs over the evaluated range reads window bytes nobody wrote -- inside this wavefront's slice -- for pixels whose
        // values are dropped here: the border of A stays zero)
        const uint32_t a4 = fast_arc4<stride>(win, xg, y, tmin);
        uint8_t* __restrict__ ap = A + y * stride + 3 + xg;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (xg + k < ew) ap[k] = (uint8_t)(a4 >> (8 * k));
    }
    FAST_WAVE_SYNC();
    // strict 3x3 maxima (neighbours outside the evaluated range hold 0), FOUR PIXELS OF A ROW PER LANE: the 3 x 6 patch of arc values
    // they need comes in as six aligned dwords (the kernel is bound by its vector issue slots; a pixel per lane spent 50 instructions
    // on index arithmetic, nine byte loads and the emission, four per lane 24).  The survivors' values replace the window, which is not
    // needed any more, four to a dword at column x - 3.  The survivors above iniThFAST are emitted in this same pass: if there is one
    // the cell keeps exactly those, and if there is none nothing was emitted -- only then does a second pass emit the survivors above
    // minThFAST (ORBextractor.cpp:796-810).  Emission order = row-major = (group, pixel of the group): four ballots, one running
    // count of the survivors in lower lanes, plus the lane's own earlier survivors.
    int base = 0;
    DSSS_GLOBAL uint32_t* __restrict__ cell_out = (DSSS_GLOBAL uint32_t*)(cand + (size_t)cell * cell_cap);      // global, not flat, stores
    for (int g0 = 0; g0 < ngroups; g0 += 64) {
        const int g = min(g0 + lane, ngroups - 1);
        const bool gvalid = g0 + lane < ngroups;
        const int gy = (int)(__umul24((unsigned)g, mg_g) >> 20), xg = 4 * (g - (int)__umul24((unsigned)gy, (unsigned)G));
        const int y = 3 + gy;
        const uint8_t* __restrict__ pr = A + (y - 1) * stride + xg;              // dword-aligned: stride and xg are multiples of four
        // bytes 2..7 of (lo, hi) of a row = columns x0 - 1 .. x0 + 4, x0 = 3 + xg
        const uint32_t t_lo = *reinterpret_cast<const uint32_t*>(pr), t_hi = *reinterpret_cast<const uint32_t*>(pr + 4);
        const uint32_t m_lo = *reinterpret_cast<const uint32_t*>(pr + stride), m_hi = *reinterpret_cast<const uint32_t*>(pr + stride + 4);
        const uint32_t b_lo = *reinterpret_cast<const uint32_t*>(pr + 2 * stride), b_hi = *reinterpret_cast<const uint32_t*>(pr + 2 * stride + 4);
        auto by = [](uint32_t lo, uint32_t hi, int k) -> int { return k < 2 ? (int)((lo >> (16 + 8 * k)) & 255u) : (int)((hi >> (8 * (k - 2))) & 255u); };
        int tv[6], mv[6], bv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { tv[k] = by(t_lo, t_hi, k); mv[k] = by(m_lo, m_hi, k); bv[k] = by(b_lo, b_hi, k); }
        int sv[4]; uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int nb = max(max(max(max(tv[k], tv[k + 1]), tv[k + 2]), max(max(bv[k], bv[k + 1]), bv[k + 2])), max(mv[k], mv[k + 2]));
            const int a = mv[k + 1];
            sv[k] = (gvalid && xg + k < ew && a > nb) ? a : 0;                   // a == 0 never passes
            packed |= (uint32_t)sv[k] << (8 * k);
        }
        if (gvalid) *reinterpret_cast<uint32_t*>(win + y * stride + xg) = packed;
        unsigned long long mk[4]; int below = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mk[k] = __builtin_amdgcn_ballot_w64(sv[k] > ini_th);
            below = __builtin_amdgcn_mbcnt_hi((unsigned)(mk[k] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk[k], below));
            total += __builtin_popcountll(mk[k]);
        }
        int pos = base + below;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (sv[k] > ini_th) {
            if (pos < cell_cap) cell_out[pos] = (uint32_t)(3 + xg + k) | ((uint32_t)y << 8) | ((uint32_t)(sv[k] - 1) << 16);
            ++pos;
        }
        base += total;
    }
    if (base == 0 && min_th < ini_th) {              // (with minThFAST >= iniThFAST the second pass could not add anything)
        FAST_WAVE_SYNC();
        for (int g0 = 0; g0 < ngroups; g0 += 64) {
            const int g = min(g0 + lane, ngroups - 1);
            const bool gvalid = g0 + lane < ngroups;
            const int gy = (int)(__umul24((unsigned)g, mg_g) >> 20), xg = 4 * (g - (int)__umul24((unsigned)gy, (unsigned)G));
            const int y = 3 + gy;
            const uint32_t packed = gvalid ? *reinterpret_cast<const uint32_t*>(win + y * stride + xg) : 0u;
            int a4[4]; unsigned long long mk[4]; int below = 0, total = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a4[k] = (int)((packed >> (8 * k)) & 255u);
                mk[k] = __builtin_amdgcn_ballot_w64(a4[k] > min_th);
                below = __builtin_amdgcn_mbcnt_hi((unsigned)(mk[k] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk[k], below));
                total += __builtin_popcountll(mk[k]);
            }
            int pos = base + below;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (a4[k] > min_th) {
                if (pos < cell_cap) cell_out[pos] = (uint32_t)(3 + xg + k) | ((uint32_t)y << 8) | ((uint32_t)(a4[k] - 1) << 16);
                ++pos;
            }
            base += total;
        }
    }
    if (lane == 0) f.counts[cell] = base < cell_cap ? base : cell_cap;
}

// candidate offsets of the cells of the levels [lev_lo, lev_hi); the first offset is where the scan of the levels below ended
// (offs[cell_begin[lev_lo]], written by that launch: the scans of a frame follow each other on one stream)
__global__ __launch_bounds__(256) void scan_counts_kernel(const ex_frame* __restrict__ frs, int lev_lo, int lev_hi)
{
    const ex_frame& f = frs[blockIdx.x];
    if (lev_lo >= f.nlevels) return;
    const int* __restrict__ counts = f.counts; int* __restrict__ offs = f.offs;
    const int lo = f.cell_begin[lev_lo], n = f.cell_begin[lev_hi];
    __shared__ int s_w[4];
    int base = lo == 0 ? 0 : offs[lo];
    __syncthreads();                                 // (offs[lo] is rewritten below)
    for (int c0 = lo; c0 < n; c0 += 256) {
        const int i = c0 + threadIdx.x;
        const int v = i < n ? counts[i] : 0;
        int tot;
        const int p = block_scan_excl256(v, &tot, s_w);
        if (i < n) offs[i] = base + p;
        base += tot;
    }
    if (threadIdx.x == 0) { offs[n] = base; if (base > f.cand_cap) *f.err = 7; }   // cannot happen with the exact bound; never silent
}

// candidates in reference order with the cell offset applied (ORBextractor.cpp:820-825)
#define GC_CELLS 32                                 // cells per workgroup (four wavefronts, eight cells each): one tiny workgroup per cell was launch-rate bound
__global__ __launch_bounds__(256) void gather_cand_kernel(const ex_frame* __restrict__ frs, int lev_lo, int lev_hi)
{
    const ex_frame& f = frs[blockIdx.y];
    const int cell_lo = f.cell_begin[lev_lo], cell_hi = f.cell_begin[lev_hi];
    const uint32_t* __restrict__ cand = f.cand; const int* __restrict__ counts = f.counts; const int* __restrict__ offs = f.offs;
    float* __restrict__ xs = f.xs; float* __restrict__ ys = f.ys; float* __restrict__ rs = f.rs; const int cap = f.cand_cap, cell_cap = f.cell_cap;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < GC_CELLS / 4; ++k) {
        const int cell = cell_lo + blockIdx.x * GC_CELLS + k * 4 + wv;
        if (cell >= cell_hi) return;
        const fast_cell c = f.cells[cell];
        const int n = counts[cell], o = offs[cell];
        for (int q = lane; q < n; q += 64) {
            if (o + q >= cap) break;
            const uint32_t v = cand[(size_t)cell * cell_cap + q];
            xs[o + q] = (float)(v & 255u) + (float)c.offx;
            ys[o + q] = (float)((v >> 8) & 255u) + (float)c.offy;
            rs[o + q] = (float)(v >> 16);
        }
    }
}

// ------------------------------------------------------------------ K5 + K6 + K7: orientation, blur, rBRIEF
typedef qt_kp_in kp_in;

__constant__ signed char c_pattern[1024] = {
#include "orb_pattern_31.inc"
};
// (umax of the patch disc, ORBextractor.cpp:454-469: { 15 15 15 15 14 14 14 13 13 12 11 10 9 8 6 3 }, folded into c_ic below; the 13 blur taps
// 1 2 7 16 31 45 52 45 31 16 7 2 1, oracle/orc_orb.c:orc_gauss13_taps, are the packed constants G0..G2 / T[] of orient_desc_kernel)
// IC_Angle over the radius-15 disc as packed dot products: row v of the 31 x 31 patch is eight aligned dwords of the LDS tile
// (columns 8 .. 39 = u -16 .. 15); wu holds u + 15 inside the disc and 0 outside, w1 holds 1 / 0, so that
// m10 = sum (u + 15) I - 15 sum I and m01 = sum v (row sum) come out of two v_dot4_u32_u8 per dword (the same integers in another order)
struct ic_tab { uint32_t wu[31 * 8], w1[31 * 8]; };
constexpr ic_tab make_ic_tab()
{
    ic_tab t{};
    constexpr int um[16] = { 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3 };
    for (int ri = 0; ri < 31; ++ri)
        for (int j = 0; j < 8; ++j) {
            uint32_t a = 0, o = 0;
            for (int k = 0; k < 4; ++k) {
                const int u = 4 * j + k - 16, v = ri - 15;
                const int au = u < 0 ? -u : u, av = v < 0 ? -v : v;
                if (au <= um[av]) { a |= (uint32_t)(u + 15) << (8 * k); o |= 1u << (8 * k); }
            }
            t.wu[ri * 8 + j] = a; t.w1[ri * 8 + j] = o;
        }
    return t;
}
__constant__ ic_tab c_ic = make_ic_tab();

#define PR 24                  // patch radius: 18 (rotated BRIEF reach) + 6 (blur)
#define PW 49
#define PS 52                  // LDS row stride of the raw patch
#define BR 18
#define BW 37
#define BS 40
#define HS 50                  // row stride (in 16-bit elements) of the transposed horizontal pass: PW rounded up to even
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

// one wave per keypoint, four keypoints per block.  The Gaussian blur of the level clone (ORBextractor.cpp:1091-1092)
// is evaluated only on the 37 x 37 patch the rotated pattern can reach, from a 49 x 49 LDS tile of the level image.
__global__ __launch_bounds__(256) void orient_desc_kernel(const ex_frame* __restrict__ frs)
{
    __shared__ __attribute__((aligned(16))) uint8_t sP[4][PW * PS];
    __shared__ __attribute__((aligned(16))) uint16_t sH[4][BW * HS];      // horizontal pass, TRANSPOSED: H[bx][py], so that the vertical taps are contiguous
    __shared__ uint8_t sB[4][BW * BS];
    const ex_frame& f = frs[blockIdx.y];
    const kp_in* __restrict__ kin = f.kin; const float* __restrict__ lscale = f.lscale;
    dsss_kp* __restrict__ kp_out = f.kptmp; uint8_t* __restrict__ desc_out = f.dtmp;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wv;
    const int n = *f.nk;
    if (blockIdx.x * 4 >= n) return;
    const bool act = k < n;
    kp_in in = act ? kin[k] : kin[0];
    const int L = in.level;
    const uint8_t* img = f.lvl[L];
    const int cols = f.cols[L], rows = f.lrows[L];
    const int cx = __float2int_rn(in.x), cy = __float2int_rn(in.y);
    uint8_t* P = sP[wv]; uint16_t* H = sH[wv]; uint8_t* B = sB[wv];
    // A patch that lies inside the level image (every keypoint more than PR + 3 pixels from the border: all but a thin frame) comes
    // in as 13 unaligned dwords per row and goes to LDS as dwords (PS = 52 = 13 dwords; the three bytes beyond column 48 are never
    // read).  The byte-wise path with its reflect-101 index arithmetic cost 950 of the kernel's 2 800 vector instructions per keypoint.
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    const bool inside = cx - PR >= 0 && cx + PR + 3 < cols && cy - PR >= 0 && cy + PR < rows;      // uniform over the wavefront
    if (act && inside) {
        const uint8_t* __restrict__ corner = img + (size_t)(cy - PR) * cols + (cx - PR);
        for (int t0 = 0; t0 < PW * 13; t0 += 64 * 10) {
            uint32_t v[10]; int o[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int t = t0 + 64 * u + lane;
                const int py = (int)(__umul24((unsigned)t, 5042u) >> 16), j = t - 13 * py;      // t / 13, exact for t < 49 * 13
                o[u] = t < PW * 13 ? py * PS + 4 * j : -1;
                v[u] = t < PW * 13 ? *reinterpret_cast<const u32_unaligned*>(corner + (size_t)__umul24((unsigned)py, (unsigned)cols) + 4 * j) : 0u;      // (py < 49, cols < 65536: a full-rate multiply)
            }
#pragma unroll
            for (int u = 0; u < 10; ++u) if (o[u] >= 0) *reinterpret_cast<uint32_t*>(P + o[u]) = v[u];
        }
    } else if (act)
        for (int t0 = 0; t0 < PW * PW; t0 += 64 * 13) {             // thirteen byte loads in flight per lane: three round trips for the
            uint8_t v[13]; int o[13];                                // 49 x 49 patch (one load per iteration made it 38)
#pragma unroll
            for (int u = 0; u < 13; ++u) {
                const int t = t0 + 64 * u + lane;
                const int py = t / PW, px = t - py * PW;
                o[u] = t < PW * PW ? py * PS + px : -1;
                v[u] = t < PW * PW ? img[(size_t)reflect101_dev(cy + py - PR, rows) * cols + reflect101_dev(cx + px - PR, cols)] : (uint8_t)0;
            }
#pragma unroll
            for (int u = 0; u < 13; ++u) if (o[u] >= 0) P[o[u]] = v[u];
        }
    __syncthreads();
    // IC_Angle (ORBextractor.cpp:77-104): integer moments over the radius-15 disc
    int m10 = 0, m01 = 0, msum = 0;
    if (act)
        for (int t = lane; t < 31 * 8; t += 64) {                    // (row, dword of the row): see c_ic
            const int ri = t >> 3, j = t & 7;
            const uint32_t pix = *reinterpret_cast<const uint32_t*>(P + (ri + PR - HALF_PATCH) * PS + 8 + 4 * j);
            const unsigned rs = __builtin_amdgcn_udot4(pix, c_ic.w1[t], 0u, false);
            m10 = (int)__builtin_amdgcn_udot4(pix, c_ic.wu[t], (unsigned)m10, false);
            msum += (int)rs; m01 += (ri - HALF_PATCH) * (int)rs;
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m10 += __shfl_xor(m10, o, 64); m01 += __shfl_xor(m01, o, 64); msum += __shfl_xor(msum, o, 64); }
    m10 -= HALF_PATCH * msum;
    const float angle = fast_atan2_dev((float)m01, (float)m10);
    // separable 13-tap blur, 8.8 fixed point (taps: oracle/orc_orb.c:orc_gauss13_taps).  The taps go through the packed dot products: a lane
    // reads the 13 bytes of its window as four aligned dwords, shifts them into place (v_alignbyte) and folds them with three
    // v_dot4_u32_u8 + one multiply-add (13 byte loads + 13 multiply-adds before); the vertical pass does the same on 16-bit sums
    // with v_dot2_u32_u16, which is why the horizontal pass stores its result transposed.  Integer arithmetic: same sums.
    const unsigned G0 = 1u | 2u << 8 | 7u << 16 | 16u << 24, G1 = 31u | 45u << 8 | 52u << 16 | 45u << 24, G2 = 31u | 16u << 8 | 7u << 16 | 2u << 24;
    if (act)
        for (int t = lane; t < PW * 10; t += 64) {                   // (row py, four columns bx0 .. bx0 + 3 of the 37-wide band): their windows
            const int py = (int)(__umul24((unsigned)t, 6554u) >> 16), g = t - 10 * py;      // share the sixteen bytes P[py][bx0 .. bx0 + 15]; t / 10 exact for t < 554
            const uint32_t* w = reinterpret_cast<const uint32_t*>(P + py * PS + 4 * g);          // PS and the slice offsets are multiples of 4
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (4 * g + o >= BW) break;                           // the tenth group holds one column
                const uint32_t a0 = o ? __builtin_amdgcn_alignbyte(w1, w0, o) : w0, a1 = o ? __builtin_amdgcn_alignbyte(w2, w1, o) : w1,
                               a2 = o ? __builtin_amdgcn_alignbyte(w3, w2, o) : w2;
                unsigned acc = __builtin_amdgcn_udot4(a0, G0, 0u, false);
                acc = __builtin_amdgcn_udot4(a1, G1, acc, false);
                acc = __builtin_amdgcn_udot4(a2, G2, acc, false);
                acc += (w3 >> (8 * o)) & 255u;                        // tap 12 has weight 1
                H[(4 * g + o) * HS + py] = (uint16_t)acc;
            }
        }
    __syncthreads();
    if (act)
        for (int t = lane; t < BW * 19; t += 64) {                   // (column bx, rows by0 and by0 + 1): taps on H[bx][by0 .. by0 + 13], seven aligned dwords
            const int bx = (int)(__umul24((unsigned)t, 3450u) >> 16), by0 = 2 * (t - 19 * bx);          // t / 19 exact for t < 767
            const uint32_t* w = reinterpret_cast<const uint32_t*>(H + bx * HS + by0);            // HS and by0 are even
            uint32_t v[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) v[k] = w[k];
            const unsigned T[6] = { 1u | 2u << 16, 7u | 16u << 16, 31u | 45u << 16, 52u | 45u << 16, 31u | 16u << 16, 7u | 2u << 16 };
            unsigned acc0 = 0, acc1 = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                acc0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, v[k]), __builtin_bit_cast(u16x2_t, T[k]), acc0, false);
                acc1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, __builtin_amdgcn_alignbyte(v[k + 1], v[k], 2)), __builtin_bit_cast(u16x2_t, T[k]), acc1, false);
            }
            acc0 += v[6] & 65535u; acc1 += v[6] >> 16;                // tap 12 has weight 1
            B[by0 * BS + bx] = (uint8_t)((acc0 + 32768u) >> 16);
            if (by0 + 1 < BW) B[(by0 + 1) * BS + bx] = (uint8_t)((acc1 + 32768u) >> 16);
        }
    __syncthreads();
    // computeOrbDescriptor (ORBextractor.cpp:108-147): 4 tests per lane
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = angle * factorPI;
    double sd, cd;
    dsss_sincos((double)ang, &sd, &cd);
    const float a = (float)cd, b = (float)sd;
    unsigned nib = 0;
    if (act) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const signed char* p = c_pattern + (lane * 4 + q) * 4;
            const float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
            const int r0 = __float2int_rn(x0 * b + y0 * a), c0 = __float2int_rn(x0 * a - y0 * b);
            const int r1 = __float2int_rn(x1 * b + y1 * a), c1 = __float2int_rn(x1 * a - y1 * b);
            const int t0 = B[(r0 + BR) * BS + (c0 + BR)], t1 = B[(r1 + BR) * BS + (c1 + BR)];
            nib |= (unsigned)(t0 < t1) << q;
        }
    }
    const unsigned hi = __shfl_down(nib, 1, 64);
    if (act && (lane & 1) == 0) desc_out[(size_t)k * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
    if (act && lane == 0) {
        dsss_kp o;
        const float sc = lscale[L];
        o.x = L ? in.x * sc : in.x; o.y = L ? in.y * sc : in.y;       // keypoint->pt *= scale (:1103-1109)
        o.size = (float)(int)(31 * sc);                              // scaledPatchSize (:837)
        o.angle = angle; o.response = in.resp; o.octave = L;
        kp_out[k] = o;
    }
}

// Frame::DetectFeature tail (frame.cpp:184-195): keep kp iff mask(int(y), int(x)) != 0, order preserved;
// also samples the geo image for the survivors (FEAmatcher.cpp:81-82).
__global__ __launch_bounds__(256) void mask_filter_kernel(const ex_frame* __restrict__ frs)
{
    __shared__ int s_w[4];
    const ex_frame& f = frs[blockIdx.x];
    const dsss_kp* __restrict__ kin = f.kptmp; const uint8_t* __restrict__ din = f.dtmp;
    const uint8_t* __restrict__ mask = f.mask; const int M = f.M;
    const double* __restrict__ pose6 = f.pose6; const double* __restrict__ gr = f.gr;
    dsss_kp* __restrict__ kout = f.kout; uint8_t* __restrict__ dout = f.dout; double* __restrict__ geo = f.geo; int* __restrict__ count = f.count;
    const int n = *f.nk;
    int base = 0;
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + threadIdx.x;
        dsss_kp kp; bool keep = false;
        if (i < n) { kp = kin[i]; keep = mask[(size_t)(int)kp.y * M + (int)kp.x] != 0; }
        int tot;
        const int pos = block_scan_excl256(keep, &tot, s_w);
        if (keep) {
            const int o = base + pos;
            kout[o] = kp;
            const uint4* s = reinterpret_cast<const uint4*>(din + (size_t)i * 32);
            uint4* d = reinterpret_cast<uint4*>(dout + (size_t)o * 32);
            d[0] = s[0]; d[1] = s[1];
            if (f.d128out) {
                const uint4* s8 = reinterpret_cast<const uint4*>(f.d128tmp + (size_t)i * 128);
                uint4* d8 = reinterpret_cast<uint4*>(f.d128out + (size_t)o * 128);
#pragma unroll
                for (int q = 0; q < 8; ++q) d8[q] = s8[q];
            }
            double x, y;
            dsss_geo_at(pose6, gr, M, (int)kp.y, (int)kp.x, &x, &y);
            geo[2 * o] = x; geo[2 * o + 1] = y;
        }
        base += tot;
    }
    if (threadIdx.x == 0) *count = base;
}

// ------------------------------------------------------------------ host orchestration
namespace {

struct level_geom {
    int nlevels;
    int rows[DSSS_MAX_LEVELS], cols[DSSS_MAX_LEVELS], quota[DSSS_MAX_LEVELS];
    float sf[DSSS_MAX_LEVELS];
    std::vector<fast_cell> cells;
    int cell_begin[DSSS_MAX_LEVELS + 1];
    int cell_cap = 0;                     // strict local maxima possible in the largest cell
    int cell_wmax = 0, cell_hmax = 0;     // largest FAST window
    long long cand_bound = 0;             // sum over the cells of the strict local maxima each can hold: the frame can never exceed it
    fast_cell* d_cells = nullptr;         // device copy (per geometry, cached in the context)
    int* d_lrows = nullptr; float* d_lscale = nullptr;
    resize_xtab* d_xt[DSSS_MAX_LEVELS] = { nullptr }; resize_ytab* d_yt[DSSS_MAX_LEVELS] = { nullptr };
    bool strips[DSSS_MAX_LEVELS] = { false };                  // level l is resized by resize_strip_kernel (the windowed loads fit)
    ~level_geom() { hipFree(d_cells); hipFree(d_lrows); hipFree(d_lscale); for (int l = 0; l < DSSS_MAX_LEVELS; ++l) { hipFree(d_xt[l]); hipFree(d_yt[l]); } }
};

// scale tables, level sizes and quotas of the ORBextractor ctor / ComputePyramid (ORBextractor.cpp:415-446,1119-1120)
void build_geom(const dsss_orb_params& op, int rows, int cols, level_geom& g)
{
    g.nlevels = op.nlevels;
    float inv[DSSS_MAX_LEVELS];
    g.sf[0] = 1.0f;
    for (int i = 1; i < op.nlevels; ++i) g.sf[i] = g.sf[i - 1] * op.scale;
    for (int i = 0; i < op.nlevels; ++i) inv[i] = 1.0f / g.sf[i];
    for (int l = 0; l < DSSS_MAX_LEVELS; ++l) { g.rows[l] = 0; g.cols[l] = 0; g.quota[l] = 0; if (l >= op.nlevels) g.sf[l] = 0; }
    for (int l = 0; l < op.nlevels; ++l) {
        g.cols[l] = (int)lrintf((float)cols * inv[l]);
        g.rows[l] = (int)lrintf((float)rows * inv[l]);
    }
    const float factor = 1.0f / op.scale;
    float nDesired = op.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)op.nlevels));
    int sum = 0;
    for (int l = 0; l < op.nlevels - 1; ++l) { g.quota[l] = (int)lrintf(nDesired); sum += g.quota[l]; nDesired *= factor; }
    g.quota[op.nlevels - 1] = std::max(op.nfeatures - sum, 0);
    // cell windows of ComputeKeyPointsOctTree (ORBextractor.cpp:769-806)
    g.cells.clear(); g.cand_bound = 0; g.cell_wmax = 0; g.cell_hmax = 0;
    int cap = 1;
    for (int l = 0; l < op.nlevels; ++l) {
        g.cell_begin[l] = (int)g.cells.size();
        const float W = 30;
        const int minBX = EDGE_T - 3, minBY = minBX, maxBX = g.cols[l] - EDGE_T + 3, maxBY = g.rows[l] - EDGE_T + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols <= 0 || nRows <= 0) continue;
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        for (int i = 0; i < nRows; ++i) {
            const float iniY = (float)(minBY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; ++j) {
                const float iniX = (float)(minBX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                fast_cell c;
                c.level = l; c.x0 = (int)iniX; c.y0 = (int)iniY; c.w = (int)maxX - (int)iniX; c.h = (int)maxY - (int)iniY;
                c.offx = j * wCell; c.offy = i * hCell; c.pad = 0;
                if (c.w > CELL_MAX) c.w = CELL_MAX;
                if (c.h > CELL_MAX) c.h = CELL_MAX;
                const int ew = std::max(c.w - 6, 0), eh = std::max(c.h - 6, 0);
                cap = std::max(cap, ((ew + 1) / 2) * ((eh + 1) / 2));
                g.cand_bound += std::min(((ew + 1) / 2) * ((eh + 1) / 2), CELL_CAP);   // no two 8-neighbours are both strict maxima
                g.cell_wmax = std::max(g.cell_wmax, c.w); g.cell_hmax = std::max(g.cell_hmax, c.h);
                g.cells.push_back(c);
            }
        }
    }
    g.cell_begin[op.nlevels] = (int)g.cells.size();
    g.cell_cap = std::min(cap, CELL_CAP);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the coefficient tables of cv::resize(INTER_LINEAR), u8 scalar path: scale = src / dst in double,
// fx = (float)((dx + 0.5) scale - 0.5), sx = floor(fx), clamped at both ends, weights cvRound((1 - fx) 2048), cvRound(fx 2048)
inline int cvfloorf_host(float v) { const int i = (int)v; return i - (v < (float)i); }
void resize_tables(int sh, int sw, int dh, int dw, std::vector<resize_xtab>& xt, std::vector<resize_ytab>& yt)
{
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    xt.resize(dw); yt.resize(dh);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvfloorf_host(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }              // dx >= xmax in OpenCV's HResize: the last source column alone
        xt[dx].sx = sx; xt[dx].a0 = (short)lrintf((1.f - fx) * 2048.f); xt[dx].a1 = (short)lrintf(fx * 2048.f);
    }
    // four more entries = the first four again (a group of four pixels that runs over the end of a row reads xt[dx .. dx + 3]
    // as one piece), then one entry whose sx says whether the windowed loads of resize_kernel are valid for this geometry: the
    // source bytes of four consecutive output pixels span at most six
    int span = 0;
    for (int dx = 0; dx + 3 < dw; ++dx) span = std::max(span, xt[dx + 3].sx + 1 - xt[dx].sx);
    for (int u = 0; u < 4; ++u) { const resize_xtab e = xt[u % dw]; xt.push_back(e); }
    resize_xtab flag; flag.sx = (span <= 5 && dw >= 8) ? 1 : 0; flag.a0 = flag.a1 = 0; xt.push_back(flag);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = cvfloorf_host(fy);
        fy -= sy;
        yt[dy].b0 = (short)lrintf((1.f - fy) * 2048.f); yt[dy].b1 = (short)lrintf(fy * 2048.f);
        yt[dy].ya = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        yt[dy].yb = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
    }
}

struct geom_key { int N, M, nf, nl, it, mt; float sc; bool operator<(const geom_key& o) const {
    return std::tie(N, M, nf, nl, it, mt, sc) < std::tie(o.N, o.M, o.nf, o.nl, o.it, o.mt, o.sc); } };
typedef std::map<geom_key, std::unique_ptr<level_geom>> geom_map;    // owned by the context (dsss_ctx::geoms), freed by dsss_destroy
void geom_map_free(void* p) { delete static_cast<geom_map*>(p); }

} // namespace

static int get_geom(dsss_ctx* c, int N, int M, level_geom** out)
{
    geom_key k{ N, M, c->op.nfeatures, c->op.nlevels, c->op.ini_th, c->op.min_th, c->op.scale };
    if (!c->geoms) { c->geoms = new geom_map(); c->geoms_free = geom_map_free; }
    geom_map& geoms = *static_cast<geom_map*>(c->geoms);
    auto it = geoms.find(k);
    if (it == geoms.end()) {
        std::unique_ptr<level_geom> g(new level_geom());
        if (N >= 65536 || M >= 65536) DSSS_FAIL(c, DSSS_E_ARG, "frames of 65536 or more pings / bins are not supported (quadtree keys pack 16-bit coordinates)");
        if ((size_t)N * (size_t)M >= ((size_t)1 << 31)) DSSS_FAIL(c, DSSS_E_ARG, "frames of 2^31 or more samples are not supported (32-bit pixel indices in the pyramid kernel)");
        build_geom(c->op, N, M, *g);
        for (int l = 0; l < g->nlevels; ++l)
            if (g->rows[l] < 2 * EDGE_T + 31 || g->cols[l] < 2 * EDGE_T + 31)
                DSSS_FAIL(c, DSSS_E_ARG, "level %d (%d x %d) too small for 30-px FAST cells", l, g->rows[l], g->cols[l]);
        HIPCHK(c, hipMalloc(&g->d_cells, sizeof(fast_cell) * std::max<size_t>(g->cells.size(), 1)));
        HIPCHK(c, hipMalloc(&g->d_lrows, sizeof(int) * DSSS_MAX_LEVELS));
        HIPCHK(c, hipMalloc(&g->d_lscale, sizeof(float) * DSSS_MAX_LEVELS));
        HIPCHK(c, hipMemcpy(g->d_cells, g->cells.data(), sizeof(fast_cell) * g->cells.size(), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(g->d_lrows, g->rows, sizeof(int) * DSSS_MAX_LEVELS, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(g->d_lscale, g->sf, sizeof(float) * DSSS_MAX_LEVELS, hipMemcpyHostToDevice));
        for (int l = 1; l < g->nlevels; ++l) {                  // cv::resize(INTER_LINEAR) tables of level l from level l - 1
            std::vector<resize_xtab> xt; std::vector<resize_ytab> yt;
            resize_tables(g->rows[l - 1], g->cols[l - 1], g->rows[l], g->cols[l], xt, yt);
            g->strips[l] = xt[g->cols[l] + 4].sx != 0;
            HIPCHK(c, hipMalloc(&g->d_xt[l], xt.size() * sizeof(resize_xtab))); HIPCHK(c, hipMalloc(&g->d_yt[l], yt.size() * sizeof(resize_ytab)));
            HIPCHK(c, hipMemcpy(g->d_xt[l], xt.data(), xt.size() * sizeof(resize_xtab), hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(g->d_yt[l], yt.data(), yt.size() * sizeof(resize_ytab), hipMemcpyHostToDevice));
        }
        it = geoms.emplace(k, std::move(g)).first;
    }
    *out = it->second.get();
    return DSSS_OK;
}

static int ensure_frame_images(dsss_ctx* c, dsss_frame& f, const level_geom& g)
{
    const size_t bytes = (size_t)f.N * f.M;
    if (f.img_cap < bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(f.mask); f.mask = nullptr;
        for (int l = 0; l < DSSS_MAX_LEVELS; ++l) { hipFree(f.lvl[l]); f.lvl[l] = nullptr; }
        HIPCHK(c, hipMalloc(&f.mask, bytes));
        f.img_cap = bytes;
    }
    for (int l = 0; l < g.nlevels; ++l) {
        if (!f.lvl[l] || f.lrows[l] != g.rows[l] || f.lcols[l] != g.cols[l]) {
            if (f.lvl[l]) { HIPCHK(c, hipStreamSynchronize(c->stream)); hipFree(f.lvl[l]); f.lvl[l] = nullptr; }
            HIPCHK(c, hipMalloc(&f.lvl[l], (size_t)g.rows[l] * g.cols[l] + 64));
        }
        f.lrows[l] = g.rows[l]; f.lcols[l] = g.cols[l];
    }
    return DSSS_OK;
}

// per-frame slot of the batch scratch (device)
struct ex_layout {
    size_t rowsum, rowmin, stats, counts, offs, cand, xs, ys, rs, keys0, keys1, work[DSSS_MAX_LEVELS], out_idx, out_n, kin, nk, err, kptmp, dtmp, d128tmp, total;
    int cand_cap, list_cap[DSSS_MAX_LEVELS], pool_cap[DSSS_MAX_LEVELS], out_cap;
};
static ex_layout make_layout(int N, const level_geom& g, int kcap, bool sift)
{
    ex_layout L; size_t o = 0;
    const int ncells = (int)g.cells.size();
    // every candidate array holds the frame's true upper bound (speckled sonar data reaches 60 % of it; a fixed
    // "typical" capacity overflowed on Rayleigh noise), so the compaction can never run past the end
    L.cand_cap = (int)std::min<long long>(g.cand_bound + 64, 0x7fffffff); L.out_cap = kcap;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    L.rowsum = take(sizeof(double) * N); L.rowmin = take(sizeof(double) * N); L.stats = take(sizeof(double) * 4);
    L.counts = take(sizeof(int) * ncells); L.offs = take(sizeof(int) * (ncells + 1));
    L.cand = take(sizeof(uint32_t) * (size_t)ncells * g.cell_cap);
    L.xs = take(sizeof(float) * L.cand_cap); L.ys = take(sizeof(float) * L.cand_cap); L.rs = take(sizeof(float) * L.cand_cap);
    L.keys0 = take(sizeof(unsigned long long) * L.cand_cap); L.keys1 = take(sizeof(unsigned long long) * L.cand_cap);
    for (int l = 0; l < DSSS_MAX_LEVELS; ++l) {
        L.list_cap[l] = 4 * g.quota[l] + 128; L.pool_cap[l] = 16 * g.quota[l] + 1024;
        L.work[l] = l < g.nlevels ? take(sizeof(int) * ((size_t)8 * L.pool_cap[l] + (size_t)10 * L.list_cap[l])) : 0;
    }
    L.out_idx = take(sizeof(int) * (size_t)DSSS_MAX_LEVELS * L.out_cap); L.out_n = take(sizeof(int) * DSSS_MAX_LEVELS);
    L.kin = take(sizeof(kp_in) * kcap); L.nk = take(sizeof(int)); L.err = take(sizeof(int));
    L.kptmp = take(sizeof(dsss_kp) * kcap); L.dtmp = take((size_t)32 * kcap);
    L.d128tmp = sift ? take((size_t)128 * kcap) : 0;
    L.total = align_up(o, 4096);
    return L;
}

#define EX_BATCH 256

// extraction of a list of frames, batched: every stage is one launch for the whole batch (the pyramid: one per level),
// about twenty launches and one host synchronisation per batch of up to EX_BATCH frames.
// phase 0: the whole extraction.  Phases 1 and 2 split it at the one kernel that needs the frame GEOMETRY (mask_filter_kernel: the geo
// sample of the surviving keypoints): phase 1 enqueues everything before it and returns without waiting, phase 2 enqueues the rest and
// finishes -- dsss_frames_set starts phase 1 while it still packs and uploads the geometry (dsss_extract_eager below), as the
// reference's Frame constructor runs DetectFeature itself (frame.cpp:45-52).  Both need the frames to fit one batch, images in HBM.
static int extract_frames_impl(dsss_ctx* c, const int* ids, int n, bool keep_taps, int phase = 0)
{
    if (n <= 0) return DSSS_OK;
    const bool doA = phase != 2, doB = phase != 1;
    const bool sift = c->op.descriptor == DSSS_DESC_SIFT128;
    int rc = sift ? dsss_ensure_sift_store(c) : dsss_ensure_store(c); if (rc) return rc;
    std::vector<level_geom*> G(n);
    size_t slot_bytes = 0;
    std::vector<ex_layout> Ls(n);
    for (int i = 0; i < n; ++i) {
        dsss_frame& f = c->frames[ids[i]];
        if (!f.has_geom || !f.has_raw) DSSS_FAIL(c, DSSS_E_STATE, "frame %d has no raw image (dsss_frame_set with raw != NULL first)", ids[i]);
        rc = get_geom(c, f.N, f.M, &G[i]); if (rc) return rc;
        rc = ensure_frame_images(c, f, *G[i]); if (rc) return rc;
        Ls[i] = make_layout(f.N, *G[i], c->kcap, sift);
        slot_bytes = std::max(slot_bytes, Ls[i].total);
    }
    // slots of one batch share one scratch allocation: bound it (24 GB) instead of the frame count alone
    // (DSSS_EX_SCRATCH_MB: tools/emulate_ranks.py puts the contexts of EIGHT ranks on one device)
    const size_t scratch_mb = getenv("DSSS_EX_SCRATCH_MB") ? (size_t)std::max(1, atoi(getenv("DSSS_EX_SCRATCH_MB"))) : 24576;
    int B = (int)std::max<size_t>(1, std::min<size_t>(std::min(n, EX_BATCH), (scratch_mb << 20) / std::max<size_t>(slot_bytes, 1)));
    // frames whose page-locked host image has not been uploaded yet: smaller batches, the upload of batch k+1 runs on the copy
    // stream (xs[1]) under the kernels of batch k (PCIe: 16 B per pixel against ~6 ns of kernels per pixel, so the copies set the pace)
    bool any_pending = false;
    for (int i = 0; i < n; ++i) any_pending = any_pending || c->frames[ids[i]].raw_pending;
    if (any_pending) { const int sb = getenv("DSSS_EX_UPLOAD_BATCH") ? std::max(1, atoi(getenv("DSSS_EX_UPLOAD_BATCH"))) : 8; B = std::min(B, sb); }
    if (phase != 0 && (any_pending || B < n)) DSSS_FAIL(c, DSSS_E_STATE, "two-phase extraction needs one batch of device-resident images");
    hipEvent_t up_ev[2] = { c->xev[1], c->xev[2] };
    auto upload_batch = [&](int b0, hipEvent_t ev) -> hipError_t {
        hipError_t e = hipSuccess;
        for (int s2 = b0; s2 < std::min(n, b0 + B) && e == hipSuccess; ++s2) {
            dsss_frame& f = c->frames[ids[s2]];
            if (!f.raw_pending) continue;
            e = hipMemcpyAsync(f.raw_owned, f.raw_host, (size_t)f.N * f.M * sizeof(double), hipMemcpyHostToDevice, c->xs[1]);
            f.raw_pending = false;
        }
        if (e == hipSuccess) e = hipEventRecord(ev, c->xs[1]);
        return e;
    };
    if (any_pending) HIPCHK(c, upload_batch(0, up_ev[0]));
    const size_t inst_bytes = align_up(sizeof(qt_inst) * (size_t)B * DSSS_MAX_LEVELS, 256), qfr_bytes = align_up(sizeof(qt_frame) * (size_t)B, 256);
    const size_t exf_bytes = align_up(sizeof(ex_frame) * (size_t)B, 256), err_bytes = align_up(sizeof(int) * (size_t)B, 256);
    const size_t tab_bytes = inst_bytes + qfr_bytes + exf_bytes + err_bytes;
    const size_t need = slot_bytes * B + tab_bytes;
    if (c->ex_scratch_bytes < need) {
        HIPCHK(c, hipStreamSynchronize(c->stream)); hipFree(c->ex_scratch); c->ex_scratch = nullptr; c->ex_scratch_bytes = 0;
        HIPCHK(c, hipMalloc(&c->ex_scratch, need)); c->ex_scratch_bytes = need;
    }
    const size_t pin_need = tab_bytes + sizeof(int) * ((size_t)B + c->max_frames) + 64;
    if (c->ex_pinned_bytes < pin_need) {
        HIPCHK(c, hipStreamSynchronize(c->stream)); if (c->ex_pinned) hipHostFree(c->ex_pinned); c->ex_pinned = nullptr; c->ex_pinned_bytes = 0;
        HIPCHK(c, hipHostMalloc(&c->ex_pinned, pin_need, hipHostMallocDefault)); c->ex_pinned_bytes = pin_need;
    }
    char* S0 = (char*)c->ex_scratch;
    char* T0 = S0 + slot_bytes * B;                            // device tables: quadtree instances, quadtree frames, batch table, error flags
    qt_inst* d_inst = (qt_inst*)T0;
    qt_frame* d_fr = (qt_frame*)(T0 + inst_bytes);
    ex_frame* d_exf = (ex_frame*)(T0 + inst_bytes + qfr_bytes);
    int* d_errs = (int*)(T0 + inst_bytes + qfr_bytes + exf_bytes);
    char* P0 = (char*)c->ex_pinned;                            // the same tables in pinned host memory, one upload per batch
    qt_inst* h_inst = (qt_inst*)P0;
    qt_frame* h_fr = (qt_frame*)(P0 + inst_bytes);
    ex_frame* h_exf = (ex_frame*)(P0 + inst_bytes + qfr_bytes);
    int* h_err = (int*)(P0 + tab_bytes);                       // [B] error flags, then [max_frames] keypoint counts
    int* h_nkp = h_err + B;
    const hipStream_t st = c->stream;

    const bool exv = getenv("DSSS_EX_VERBOSE") != nullptr && any_pending;
    double tv_issue = 0, tv_sync = 0, tv_wait = 0; int tv_n = 0;
    auto tv_now = [] { return std::chrono::steady_clock::now(); };
    for (int b0 = 0, bk = 0; b0 < n; b0 += B, ++bk) {
        const int nb = std::min(B, n - b0);
        const auto tv0 = tv_now();
        if (any_pending) {
            if (b0 + B < n) HIPCHK(c, upload_batch(b0 + B, up_ev[(bk + 1) & 1]));      // next batch's images start moving now
            HIPCHK(c, hipStreamWaitEvent(st, up_ev[bk & 1], 0));                       // this batch's images are in HBM before its kernels read them
        }
        int maxN = 0, maxNM4 = 0, max_cells = 0, max_levels = 0, max_cw = 8, max_ch = 8;
        int lev_cnt[DSSS_MAX_LEVELS] = { 0 };           // frames that have level l
        size_t max_tot = 0;
        int max_rows[DSSS_MAX_LEVELS] = { 0 }, max_cols[DSSS_MAX_LEVELS] = { 0 };
        double w_tot = 0;
        for (int s = 0; s < nb; ++s) {
            const int id = ids[b0 + s];
            dsss_frame& f = c->frames[id];
            const level_geom& g = *G[b0 + s];
            const ex_layout& L = Ls[b0 + s];
            char* S = S0 + slot_bytes * s;
            ex_frame& e = h_exf[s];
            e.raw = f.raw; e.N = f.N; e.M = f.M;
            e.rowsum = (double*)(S + L.rowsum); e.rowmin = (double*)(S + L.rowmin); e.stats = (double*)(S + L.stats);
            e.mask = f.mask; e.nlevels = g.nlevels;
            for (int l = 0; l < DSSS_MAX_LEVELS; ++l) { e.lvl[l] = l < g.nlevels ? f.lvl[l] : nullptr; e.rows[l] = l < g.nlevels ? g.rows[l] : 0; e.cols[l] = l < g.nlevels ? g.cols[l] : 0; }
            e.cells = g.d_cells; e.ncells = (int)g.cells.size(); e.cell_cap = g.cell_cap;
            for (int l = 0; l <= DSSS_MAX_LEVELS; ++l) e.cell_begin[l] = g.cell_begin[std::min(l, g.nlevels)];
            e.cand = (uint32_t*)(S + L.cand); e.counts = (int*)(S + L.counts); e.offs = (int*)(S + L.offs);
            e.xs = (float*)(S + L.xs); e.ys = (float*)(S + L.ys); e.rs = (float*)(S + L.rs); e.cand_cap = L.cand_cap;
            e.kin = (kp_in*)(S + L.kin); e.nk = (int*)(S + L.nk); e.lrows = g.d_lrows; e.lscale = g.d_lscale;
            for (int l = 0; l < DSSS_MAX_LEVELS; ++l) { e.xt[l] = g.d_xt[l]; e.yt[l] = g.d_yt[l]; }
            e.kptmp = (dsss_kp*)(S + L.kptmp); e.dtmp = (uint8_t*)(S + L.dtmp);
            e.d128tmp = sift ? (uint8_t*)(S + L.d128tmp) : nullptr; e.d128out = sift ? c->desc128 + (size_t)id * c->kcap * 128 : nullptr;
            e.pose6 = f.pose6; e.gr = f.gr;
            e.err = d_errs + s; e.kout = c->kps + (size_t)id * c->kcap; e.dout = c->desc + (size_t)id * c->kcap * 32; e.geo = c->geo + (size_t)id * c->kcap * 2; e.count = c->nkp_dev + id;
            maxN = std::max(maxN, f.N); max_tot = std::max(max_tot, (size_t)f.N * f.M); max_cells = std::max(max_cells, e.ncells); max_levels = std::max(max_levels, g.nlevels);
            max_cw = std::max(max_cw, g.cell_wmax); max_ch = std::max(max_ch, g.cell_hmax);
            for (int l = 0; l < g.nlevels; ++l) { max_rows[l] = std::max(max_rows[l], g.rows[l]); max_cols[l] = std::max(max_cols[l], g.cols[l]); }
            w_tot += (double)f.N * f.M;
            // quadtree descriptors, level-major: the instances of level l are one launch
            for (int l = 0; l < g.nlevels; ++l) {
                qt_inst& q = h_inst[(size_t)l * B + lev_cnt[l]++];
                q.offs = e.offs; q.cell_begin = g.cell_begin[l]; q.cell_end = g.cell_begin[l + 1];
                q.xs = e.xs; q.ys = e.ys; q.rs = e.rs;
                q.W = (g.cols[l] - EDGE_T + 3) - (EDGE_T - 3); q.H = (g.rows[l] - EDGE_T + 3) - (EDGE_T - 3); q.quota = g.quota[l];
                q.keys0 = (unsigned long long*)(S + L.keys0); q.keys1 = (unsigned long long*)(S + L.keys1); q.work = (int*)(S + L.work[l]);
                q.list_cap = L.list_cap[l]; q.pool_cap = L.pool_cap[l];
                q.out_idx = (int*)(S + L.out_idx) + (size_t)l * L.out_cap; q.out_n = (int*)(S + L.out_n) + l; q.out_cap = L.out_cap;
                q.err = d_errs + s; q.cand_cap = L.cand_cap;
            }
            qt_frame& qf = h_fr[s];
            qf.nlevels = g.nlevels; qf.out_cap = L.out_cap; qf.kcap = c->kcap; qf.min_border = EDGE_T - 3;
            qf.out_idx = (int*)(S + L.out_idx); qf.out_n = (int*)(S + L.out_n);
            qf.xs = e.xs; qf.ys = e.ys; qf.rs = e.rs;
            qf.kin = (kp_in*)(S + L.kin); qf.nk = (int*)(S + L.nk); qf.err = d_errs + s;
        }
        (void)maxNM4;
        if (doA) {
        HIPCHK(c, hipMemcpyAsync(T0, P0, tab_bytes - err_bytes, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemsetAsync(d_errs, 0, sizeof(int) * nb, st));
        { dsss_scope sc(c, DSSS_K_ROW_REDUCE, 8.0 * w_tot);
          hipLaunchKernelGGL(row_reduce_kernel, dim3((maxN + 3) / 4, nb), dim3(256), 0, st, d_exf); }
        { dsss_scope sc(c, DSSS_K_PRE_MISC, 1.0 * w_tot, 2);
          hipLaunchKernelGGL(final_reduce_kernel, dim3(nb), dim3(64), 0, st, d_exf, (double)(float)c->mp.factor);
          hipLaunchKernelGGL(mask_init_kernel, dim3((unsigned)((max_tot / 16 + 256) / 256), nb), dim3(256), 0, st, d_exf, c->mp.width, c->mp.side, (double)c->mp.side * 0.6); }
        { dsss_scope sc(c, DSSS_K_NORMALIZE, 9.0 * w_tot);
          hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((max_tot / 4 + 256) / 256), nb), dim3(256), 0, st, d_exf, c->mp.r); }
        { dsss_scope sc(c, DSSS_K_PYRAMID, (1.906 + 2.74) * w_tot, std::max(max_levels - 1, 1));
          for (int l = 1; l < max_levels; ++l) {
              bool strips = false, rest = false;
              for (int s2 = 0; s2 < nb; ++s2) { const level_geom& g = *G[b0 + s2]; if (l < g.nlevels) (g.strips[l] ? strips : rest) = true; }
              if (strips) hipLaunchKernelGGL(resize_strip_kernel, dim3((unsigned)(((size_t)((max_cols[l] + 3) / 4) * ((max_rows[l] + RS_R - 1) / RS_R) + 255) / 256), nb), dim3(256), 0, st, d_exf, l);
              if (rest) hipLaunchKernelGGL(resize_kernel, dim3((unsigned)(((size_t)max_cols[l] * max_rows[l] / 4 + 256 * RS_K) / (256 * RS_K)), nb), dim3(256), 0, st, d_exf, l, strips ? 1 : 0);
          } }
        // FAST, candidate compaction and the quadtree run BY GROUPS OF LEVELS.  A quadtree instance is one workgroup whose time is the latency
        // of its own level's candidates (1.4 ms on level 0 of a 2000 x 1024 frame, whatever the number of frames) during which most of the
        // chip idles, so the levels are pipelined over four streams:
        //     main            FAST of group 0, 1, ... back to back (the groups are independent of each other)
        //     xs[2], xs[3]    per group, once its FAST is done: offsets (a chain over the groups: one more event) + candidate gather + the
        //                     quadtrees of its levels; the groups alternate between the two streams
        // and the main stream picks the results up after the last group.  Groups: levels 0 .. 4 alone, the small top levels together
        // (every launch ends with a partly empty chip).  Two side streams, not more: the runtime folds streams onto four hardware queues,
        // and a quadtree that shares its queue with the next group's gather holds it up.
        const int solo_levels = 5;
        const int fstride = max_cw <= 40 ? 40 : CELL_STRIDE, fwave = (2 * max_ch * fstride + 15) & ~15;      // window + arc values of one wavefront
        const hipStream_t s_qt[2] = { c->xs[2], c->xs[3] };
        bool qt_used[2] = { false, false };
        int prev_group = -1;
        int ngroups = 0;
        for (int lo = 0; lo < max_levels; ++ngroups) {
            const int hi = lo < solo_levels ? lo + 1 : max_levels;
            int cells = 0;                               // the most cells a frame has in the group
            for (int s2 = 0; s2 < nb; ++s2) { const level_geom& g = *G[b0 + s2]; cells = std::max(cells, g.cell_begin[std::min(hi, g.nlevels)] - g.cell_begin[std::min(lo, g.nlevels)]); }
            if (cells > 0) {
                { dsss_scope sc(c, DSSS_K_FAST, lo == 0 ? 2.906 * w_tot : 0.0);
                  if (fstride == 40) hipLaunchKernelGGL(fast_cells_kernel<40>, dim3((cells + 3) / 4, nb), dim3(256), 4 * fwave, st, d_exf, lo, hi, c->op.ini_th, c->op.min_th, fwave);
                  else hipLaunchKernelGGL(fast_cells_kernel<CELL_STRIDE>, dim3((cells + 3) / 4, nb), dim3(256), 4 * fwave, st, d_exf, lo, hi, c->op.ini_th, c->op.min_th, fwave); }
                const int k = ngroups & 1;
                const hipStream_t ss = s_qt[k];
                {
                    HIPCHK(c, hipEventRecord(c->ex_lev_ev[ngroups], st)); HIPCHK(c, hipStreamWaitEvent(ss, c->ex_lev_ev[ngroups], 0));
                    if (prev_group >= 0) HIPCHK(c, hipStreamWaitEvent(ss, c->ex_cmp_ev[prev_group], 0));      // this group's offsets start where the previous group's end
                }
                hipLaunchKernelGGL(scan_counts_kernel, dim3(nb), dim3(256), 0, ss, d_exf, lo, hi);
                HIPCHK(c, hipEventRecord(c->ex_cmp_ev[ngroups], ss));
                prev_group = ngroups;
                hipLaunchKernelGGL(gather_cand_kernel, dim3((cells + GC_CELLS - 1) / GC_CELLS, nb), dim3(256), 0, ss, d_exf, lo, hi);
                for (int l = lo; l < hi; ++l) if (lev_cnt[l] > 0) dsss_launch_quadtree(ss, d_inst + (size_t)l * B, lev_cnt[l]);
                qt_used[k] = true;
            }
            lo = hi;
        }
        { dsss_scope sc(c, DSSS_K_QUADTREE, 0, max_levels);        // (with the level pipeline: what compaction and quadtrees leave exposed after the last FAST launch)
          for (int k = 0; k < 2; ++k) if (qt_used[k]) { HIPCHK(c, hipEventRecord(c->ex_side_ev[k], s_qt[k])); HIPCHK(c, hipStreamWaitEvent(st, c->ex_side_ev[k], 0)); }
          dsss_launch_quadtree_collect(st, d_fr, nb); }
        { dsss_scope sc(c, DSSS_K_DESC, (double)nb * c->op.nfeatures * (49.0 * 49.0 + 56.0));
          hipLaunchKernelGGL(orient_desc_kernel, dim3((c->kcap + 3) / 4, nb), dim3(256), 0, st, d_exf); }
        if (sift) { dsss_scope sc(c, DSSS_K_SIFT, (double)nb * c->op.nfeatures * (71.0 * 71.0 + 128.0));
          dsss_launch_sift_desc(c, st, d_exf, c->kcap, nb); }               // N4: the 128-element rows at the same keypoints (dsss_sift.hip)
        HIPCHK(c, hipGetLastError());
        }       // doA
        if (!doB) return DSSS_OK;                    // (phase 1: one batch)
        { dsss_scope sc(c, DSSS_K_FILTER);
          hipLaunchKernelGGL(mask_filter_kernel, dim3(nb), dim3(256), 0, st, d_exf); }
        HIPCHK(c, hipGetLastError());
        if (b0 + B >= n) { const int rb = dsss_bboxes_enqueue(c); if (rb) return rb; }      // the geo boxes the matcher will ask for ride on this batch's synchronisation
        HIPCHK(c, hipMemcpyAsync(h_err, d_errs, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(h_nkp, c->nkp_dev, sizeof(int) * c->max_frames, hipMemcpyDeviceToHost, st));
        const auto tv1 = tv_now();
        if (exv && any_pending) { (void)hipEventSynchronize(up_ev[bk & 1]); }
        const auto tv2 = tv_now();
        HIPCHK(c, hipStreamSynchronize(st));         // one synchronisation per batch of up to EX_BATCH frames
        if (exv) { const auto tv3 = tv_now(); auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
                   tv_issue += us(tv0, tv1); tv_wait += us(tv1, tv2); tv_sync += us(tv2, tv3); ++tv_n; }
        for (int s = 0; s < nb; ++s) {
            dsss_frame& f = c->frames[ids[b0 + s]];
            if (h_err[s]) DSSS_FAIL(c, DSSS_E_CAPACITY, "frame %d: extraction capacity exceeded (code %d; 7 = FAST candidates, else quadtree lists)", ids[b0 + s], h_err[s]);
            f.nkp = h_nkp[ids[b0 + s]]; f.has_feat = true; f.has_norm = true; f.has_sift = sift;
        }
        if (keep_taps) {                             // stage tap for the parity tests: FAST candidates per level
            for (int s = 0; s < nb; ++s) {
                dsss_frame& f = c->frames[ids[b0 + s]];
                const level_geom& g = *G[b0 + s];
                const ex_layout& L = Ls[b0 + s];
                char* S = S0 + slot_bytes * s;
                const int ncells = (int)g.cells.size();
                std::vector<int> offs(ncells + 1);
                HIPCHK(c, hipMemcpy(offs.data(), S + L.offs, sizeof(int) * (ncells + 1), hipMemcpyDeviceToHost));
                const int ncand = std::min(offs[ncells], L.cand_cap);
                std::vector<float> xs(ncand), ys(ncand), rs(ncand);
                if (ncand) {
                    HIPCHK(c, hipMemcpy(xs.data(), S + L.xs, sizeof(float) * ncand, hipMemcpyDeviceToHost));
                    HIPCHK(c, hipMemcpy(ys.data(), S + L.ys, sizeof(float) * ncand, hipMemcpyDeviceToHost));
                    HIPCHK(c, hipMemcpy(rs.data(), S + L.rs, sizeof(float) * ncand, hipMemcpyDeviceToHost));
                }
                for (int l = 0; l < g.nlevels; ++l) {
                    const int b = std::min(offs[g.cell_begin[l]], ncand), e = std::min(offs[g.cell_begin[l + 1]], ncand);
                    f.cand_x[l].assign(xs.begin() + b, xs.begin() + e); f.cand_y[l].assign(ys.begin() + b, ys.begin() + e); f.cand_r[l].assign(rs.begin() + b, rs.begin() + e);
                }
            }
        }
    }
    if (exv && tv_n) fprintf(stderr, "[dsss extract] %d batches of %d: per batch host issue %.0f us, wait for the batch's upload %.0f us, kernels after it %.0f us\n", tv_n, B, tv_issue / tv_n, tv_wait / tv_n, tv_sync / tv_n);
    return DSSS_OK;
}

// Error exits of the batch loop can leave asynchronous uploads FROM CALLER MEMORY queued on the copy stream (the upload of batch
// k + 1 is issued before the kernels of batch k have run): no error is returned before those copies have finished, so the caller's
// page-locked images are free again whenever dsss_extract* returns, with or without an error (include/dsss.h, dsss_frame_set).
static int extract_frames(dsss_ctx* c, const int* ids, int n, bool keep_taps)
{
    // the frames dsss_frames_set already started on (phase 1): only the tail is left.  Anything else drops that start (its kernels
    // are queued on the context's stream ahead of whatever follows; their results are simply overwritten)
    int phase = 0;
    if (c->ex_eager_valid) {
        const bool same = !keep_taps && (int)c->ex_eager_ids.size() == n && std::equal(ids, ids + n, c->ex_eager_ids.begin())
                          && memcmp(&c->ex_eager_op, &c->op, sizeof c->op) == 0 && memcmp(&c->ex_eager_mp, &c->mp, sizeof c->mp) == 0;
        c->ex_eager_valid = false;
        if (same) phase = 2;
    }
    const int rc = extract_frames_impl(c, ids, n, keep_taps, phase);
    if (rc != DSSS_OK) { for (int k = 0; k < 4; ++k) (void)hipStreamSynchronize(c->xs[k]); (void)hipStreamSynchronize(c->stream); (void)hipGetLastError(); }
    return rc;
}

// dsss_frames_set: start the extraction of the frames whose images are in HBM (see extract_frames_impl).  Not an error if it cannot.
void dsss_extract_eager(dsss_ctx* c, const int* ids, int n)
{
    c->ex_eager_valid = false;
    const bool exv = getenv("DSSS_EX_VERBOSE") != nullptr;
    if (n <= 0 || n > EX_BATCH) { if (exv) fprintf(stderr, "[dsss extract] no early start: %d frames in the call\n", n); return; }
    std::vector<int> mine;
    for (int i = 0; i < n; ++i) {
        const dsss_frame& f = c->frames[ids[i]];
        if (!f.has_raw) continue;                    // (a rank sets every frame's geometry and the images of its own)
        if (f.raw_pending) { if (exv) fprintf(stderr, "[dsss extract] no early start: frame %d is still in host memory\n", ids[i]); return; }      // host-resident images are streamed in by the batch loop of the full path
        mine.push_back(ids[i]);
    }
    if (mine.empty()) return;
    if (extract_frames_impl(c, mine.data(), (int)mine.size(), false, 1) != DSSS_OK) { if (exv) fprintf(stderr, "[dsss extract] no early start: %s\n", c->err.c_str()); (void)hipGetLastError(); return; }      // (the full path will report what is wrong)
    c->ex_eager_ids.swap(mine); c->ex_eager_op = c->op; c->ex_eager_mp = c->mp; c->ex_eager_valid = true;
}

extern "C" {

int dsss_extract(dsss_ctx* c, int id, int* n_kp)
{
    if (!c) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range", id);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = extract_frames(c, &id, 1, true);
    if (rc) return rc;
    if (n_kp) *n_kp = c->frames[id].nkp;
    return DSSS_OK;
}

int dsss_extract_many(dsss_ctx* c, const int* ids, int n)
{
    if (!c || (n > 0 && !ids)) return DSSS_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range", ids[i]);
    return extract_frames(c, ids, n, false);
}

int dsss_host_quadtree(const float* x, const float* y, const float* resp, int n, int minX, int maxX, int minY, int maxY,
                       int quota, int32_t* keep_idx, int* n_keep)
{
    if (n < 0 || (n > 0 && (!x || !y || !resp || !keep_idx)) || maxX <= minX || maxY <= minY) return DSSS_E_ARG;
    std::vector<int> keep;
    dsss_quadtree_cull(x, y, resp, n, minX, maxX, minY, maxY, quota, keep);
    for (size_t i = 0; i < keep.size(); ++i) keep_idx[i] = keep[i];
    if (n_keep) *n_keep = (int)keep.size();
    return DSSS_OK;
}

int dsss_frame_get_norm(dsss_ctx* c, int id, uint8_t* norm, uint8_t* mask)
{
    if (!c) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames) DSSS_FAIL(c, DSSS_E_ARG, "frame id %d out of range", id);
    dsss_frame& f = c->frames[id];
    if (!f.has_norm) DSSS_FAIL(c, DSSS_E_STATE, "frame %d not extracted yet", id);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (norm) HIPCHK(c, hipMemcpy(norm, f.lvl[0], (size_t)f.N * f.M, hipMemcpyDeviceToHost));
    if (mask) HIPCHK(c, hipMemcpy(mask, f.mask, (size_t)f.N * f.M, hipMemcpyDeviceToHost));
    return DSSS_OK;
}

int dsss_frame_get_level(dsss_ctx* c, int id, int level, uint8_t* img, int* rows, int* cols)
{
    if (!c) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames || level < 0 || level >= c->op.nlevels) DSSS_FAIL(c, DSSS_E_ARG, "frame/level out of range");
    dsss_frame& f = c->frames[id];
    if (!f.has_norm) DSSS_FAIL(c, DSSS_E_STATE, "frame %d not extracted yet", id);
    if (rows) *rows = f.lrows[level];
    if (cols) *cols = f.lcols[level];
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (img) HIPCHK(c, hipMemcpy(img, f.lvl[level], (size_t)f.lrows[level] * f.lcols[level], hipMemcpyDeviceToHost));
    return DSSS_OK;
}

int dsss_frame_get_candidates(dsss_ctx* c, int id, int level, float* x, float* y, float* r, int cap, int* n)
{
    if (!c) return DSSS_E_ARG;
    if (id < 0 || id >= c->max_frames || level < 0 || level >= c->op.nlevels) DSSS_FAIL(c, DSSS_E_ARG, "frame/level out of range");
    dsss_frame& f = c->frames[id];
    if (!f.has_norm) DSSS_FAIL(c, DSSS_E_STATE, "frame %d not extracted yet", id);
    const int m = (int)f.cand_x[level].size();
    if (n) *n = m;
    if (cap < m) DSSS_FAIL(c, DSSS_E_CAPACITY, "caller capacity %d < %d candidates", cap, m);
    if (x) memcpy(x, f.cand_x[level].data(), sizeof(float) * m);
    if (y) memcpy(y, f.cand_y[level].data(), sizeof(float) * m);
    if (r) memcpy(r, f.cand_r[level].data(), sizeof(float) * m);
    return DSSS_OK;
}

} // extern "C"
