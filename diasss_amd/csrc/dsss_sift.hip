// diasss_amd/csrc/dsss_sift.hip -- the 128-float descriptor of the reference's SIFT call site on the device (gfx950).
// Replaces computeDescriptors(image, keypoints, descriptors) = cv::SIFT::compute on the blurred pyramid level
// (/root/reference/thirdparty/ORBextractor.cpp:1043-1047, called at :1098 behind the GaussianBlur of :1091-1092) with what its author
// evidently meant: Lowe's 4 x 4 x 8 gradient-orientation histogram at the ORB keypoints of the level they were found on, at their IC angle.
// The arithmetic is DEFINED by oracle/orc_sift.c (8-pixel spatial bins, 57 x 57 window, Gaussian weight from a table over i^2 + j^2,
// every trilinear share rounded to 2^-12 fixed point, normalise / clip 0.2 / renormalise on exact integers) and reproduced here bit for bit:
// the per-sample float operations are the oracle's (library built -ffp-contract=off), and the histogram is summed with LDS integer atomics --
// integer sums do not depend on the order in which 256 threads add.  Rows are stored as bytes (the floats are integer-valued 0..255):
// 128 B per keypoint instead of 512 in HBM; dsss_features_get_sift widens them.
//
// One workgroup per keypoint (26 VGPRs, 18.6 KB of LDS: eight workgroups per compute unit).  LDS: the 71 x 71 raw window of the level image
// (57 + 2 for the gradients + 12 for the 13-tap blur; unaligned dword loads inside the image, reflect-101 bytes near its border), the
// horizontal blur pass transposed, the 59 x 59 blurred window, 360 histogram cells.  The separable 8.8 fixed-point blur is the one of
// orient_desc_kernel (oracle/orc_orb.c:orc_blur13: reflect-101 at the borders of the level image, (v + 2^15) >> 16), through v_dot4_u32_u8 /
// v_dot2_u32_u16.  The live samples of the window (inside the rotated square and the image: about half) are compacted by ballot first; a
// thread then walks a contiguous run of that list so that the lanes of a wavefront add to different histogram cells.
// Measured at C3 (400 k keypoints): 13.4 ms (byte loads, scalar blur taps, divergent sample loop) -> 7.3 (dword loads, dot-product blur,
// compaction) -> 5.9 ms (lanes spread over the list); what is left is the 8 LDS atomics per live sample (12.8 k per keypoint).
#include "dsss_extract.h"

#define SRAD 28                // window radius: cvRound(8 sqrt 2 (4 + 1) / 2)
#define SW 57                  // samples per side
#define GW 59                  // blurred window (samples + the gradient's neighbours)
#define GS 60                  // its LDS row stride
#define RW 71                  // raw window (blurred + 6 on every side)
#define RS 72                  // its LDS row stride (dwords: 18)
#define HSS 72                 // row stride (16-bit elements) of the transposed horizontal pass: H[bx][py], py < 71
#define NHIST 360              // (4 + 2) x (4 + 2) x (8 + 2)

typedef unsigned short u16x2_s __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void sift_desc_kernel(const ex_frame* __restrict__ frs, const float* __restrict__ wtab, int* __restrict__ dbg, int dbg_kcap)
{
    __shared__ __attribute__((aligned(16))) uint8_t sP[RW * RS];
    __shared__ __attribute__((aligned(16))) uint16_t sH[GW * HSS];       // (later: the list of the window's live samples)
    __shared__ __attribute__((aligned(16))) uint8_t sB[GW * GS];
    __shared__ int sHist[NHIST];
    __shared__ unsigned long long sSum[4];
    __shared__ int sCount;
    const ex_frame& f = frs[blockIdx.y];
    const int k = blockIdx.x;
    if (k >= *f.nk) return;
    const qt_kp_in in = f.kin[k];
    const int L = in.level;
    const uint8_t* __restrict__ img = f.lvl[L];
    const int cols = f.cols[L], rows = f.rows[L];
    const int cx = __float2int_rn(in.x), cy = __float2int_rn(in.y);
    const float angle = f.kptmp[k].angle;                       // IC_Angle, written by orient_desc_kernel earlier on this stream
    const int tid = threadIdx.x, lane = tid & 63;
    for (int t = tid; t < NHIST; t += 256) sHist[t] = 0;
    if (tid == 0) sCount = 0;
    // raw window.  Inside the level image (all but a frame of 35 pixels): 18 unaligned dwords per row (the 72nd byte is never used; the level
    // buffers carry 64 bytes of slack behind the last row).  Near the border: byte by byte with reflect-101, as the blur of the level clone sees it.
    const int R0 = SRAD + 1 + 6;                                 // 35
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    if (cx - R0 >= 0 && cx + R0 < cols && cy - R0 >= 0 && cy + R0 < rows) {
        const uint8_t* __restrict__ corner = img + (size_t)(cy - R0) * cols + (cx - R0);
        uint32_t v[5]; int o[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int t = tid + 256 * u;
            const int py = (int)(__umul24((unsigned)t, 3641u) >> 16), j = t - 18 * py;      // t / 18, exact for t < 71 * 18 = 1278
            o[u] = t < RW * 18 ? py * RS + 4 * j : -1;
            v[u] = t < RW * 18 ? *reinterpret_cast<const u32_unaligned*>(corner + (size_t)__umul24((unsigned)py, (unsigned)cols) + 4 * j) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) if (o[u] >= 0) *reinterpret_cast<uint32_t*>(sP + o[u]) = v[u];
    } else
        for (int t = tid; t < RW * RW; t += 256) {
            const int py = t / RW, px = t - py * RW;
            sP[py * RS + px] = img[(size_t)reflect101_dev(cy + py - R0, rows) * cols + reflect101_dev(cx + px - R0, cols)];
        }
    __syncthreads();
    // separable 13-tap blur in 8.8 fixed point through the packed dot products, as orient_desc_kernel does it: four outputs of a row share
    // the sixteen bytes P[py][4 g .. 4 g + 15] (four aligned dwords, shifted into place by v_alignbyte, three v_dot4_u32_u8 + the last tap);
    // the result goes to H TRANSPOSED so that the vertical taps are contiguous 16-bit pairs for v_dot2_u32_u16.  Integer arithmetic: same sums.
    const unsigned G0 = 1u | 2u << 8 | 7u << 16 | 16u << 24, G1 = 31u | 45u << 8 | 52u << 16 | 45u << 24, G2 = 31u | 16u << 8 | 7u << 16 | 2u << 24;
    for (int t = tid; t < RW * 15; t += 256) {
        const int py = (int)(__umul24((unsigned)t, 4370u) >> 16), g = t - 15 * py;          // t / 15, exact for t < 71 * 15 = 1065
        const uint32_t* w = reinterpret_cast<const uint32_t*>(sP + py * RS + 4 * g);
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (4 * g + o >= GW) break;                          // the fifteenth group holds three columns
            const uint32_t a0 = o ? __builtin_amdgcn_alignbyte(w1, w0, o) : w0, a1 = o ? __builtin_amdgcn_alignbyte(w2, w1, o) : w1,
                           a2 = o ? __builtin_amdgcn_alignbyte(w3, w2, o) : w2;
            unsigned acc = __builtin_amdgcn_udot4(a0, G0, 0u, false);
            acc = __builtin_amdgcn_udot4(a1, G1, acc, false);
            acc = __builtin_amdgcn_udot4(a2, G2, acc, false);
            acc += (w3 >> (8 * o)) & 255u;                       // tap 12 has weight 1
            sH[(4 * g + o) * HSS + py] = (uint16_t)acc;
        }
    }
    __syncthreads();
    for (int t = tid; t < GW * 30; t += 256) {                   // (column bx, rows by0 and by0 + 1): taps on H[bx][by0 .. by0 + 13], seven aligned dwords
        const int bx = (int)(__umul24((unsigned)t, 2185u) >> 16), by0 = 2 * (t - 30 * bx);       // t / 30, exact for t < 59 * 30 = 1770
        const uint32_t* w = reinterpret_cast<const uint32_t*>(sH + bx * HSS + by0);
        uint32_t v[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) v[q] = w[q];
        const unsigned T[6] = { 1u | 2u << 16, 7u | 16u << 16, 31u | 45u << 16, 52u | 45u << 16, 31u | 16u << 16, 7u | 2u << 16 };
        unsigned acc0 = 0, acc1 = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            acc0 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_s, v[q]), __builtin_bit_cast(u16x2_s, T[q]), acc0, false);
            acc1 = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_s, __builtin_amdgcn_alignbyte(v[q + 1], v[q], 2)), __builtin_bit_cast(u16x2_s, T[q]), acc1, false);
        }
        acc0 += v[6] & 65535u; acc1 += v[6] >> 16;                // tap 12 has weight 1
        sB[by0 * GS + bx] = (uint8_t)((acc0 + 32768u) >> 16);
        if (by0 + 1 < GW) sB[(by0 + 1) * GS + bx] = (uint8_t)((acc1 + 32768u) >> 16);
    }
    __syncthreads();
    // the samples (oracle/orc_sift.c:orc_sift_hist, operation for operation)
    float ori = 360.f - angle;
    if (fabsf(ori - 360.f) < 1.1920928955078125e-7f) ori = 0.f;
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    double sd, cd;
    dsss_sincos((double)(ori * factorPI), &sd, &cd);
    const float cos_t = (float)cd / 8.0f, sin_t = (float)sd / 8.0f;
    const float bins_per_deg = 8 / 360.f;
    // The window's LIVE samples first (inside the rotated 5 x 5-bin square and inside the image: about half of the 57 x 57), compacted into a
    // list by ballot so that the arithmetic below runs on full wavefronts; the order of the list does not matter -- the histogram is a sum of
    // integers.  The list lives where the horizontal pass was.
    uint16_t* sList = sH;
    for (int t0 = 0; t0 < SW * SW; t0 += 256) {
        const int t = t0 + tid;
        bool live = false;
        if (t < SW * SW) {
            const int si = (int)(__umul24((unsigned)t, 1150u) >> 16), sj = t - si * SW;      // t / 57, exact for t < 3249
            const int i = si - SRAD, j = sj - SRAD;
            const float c_rot = (float)j * cos_t - (float)i * sin_t;
            const float r_rot = (float)j * sin_t + (float)i * cos_t;
            const float rbin = r_rot + 1.5f, cbin = c_rot + 1.5f;
            const int r = cy + i, c = cx + j;
            live = rbin > -1 && rbin < 4 && cbin > -1 && cbin < 4 && r > 0 && r < rows - 1 && c > 0 && c < cols - 1;
        }
        const unsigned long long m = __ballot(live);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&sCount, __popcll(m));
        base = __shfl(base, 0, 64);
        if (live) sList[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)t;
    }
    __syncthreads();
    // a thread walks a CONTIGUOUS run of the list, so that at any moment the lanes of a wavefront sit ~25 samples apart -- in different
    // spatial cells: neighbouring samples share their histogram cells, and 64 lanes adding to a handful of LDS words serialise
    const int nlive = sCount, per = (nlive + 255) >> 8;
    const int q0 = (lane * 4 + (tid >> 6)) * per;
    for (int q = q0; q < min(q0 + per, nlive); ++q) {
        const int t = sList[q];
        const int si = (int)(__umul24((unsigned)t, 1150u) >> 16), sj = t - si * SW;
        const int i = si - SRAD, j = sj - SRAD;
        const float c_rot = (float)j * cos_t - (float)i * sin_t;
        const float r_rot = (float)j * sin_t + (float)i * cos_t;
        float rbin = r_rot + 1.5f, cbin = c_rot + 1.5f;
        const uint8_t* b = sB + (si + 1) * GS + (sj + 1);
        const float dx = (float)((int)b[1] - (int)b[-1]);
        const float dy = (float)((int)b[-GS] - (int)b[GS]);
        const float Ori = fast_atan2_dev(dy, dx);
        const float Mag = sqrtf(dx * dx + dy * dy);      // (sqrtf, correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt; __fsqrt_rn is the bare v_sqrt_f32: 1 ulp)
        float obin = (Ori - ori) * bins_per_deg;
        const float mag = Mag * wtab[i * i + j * j];
        const int r0 = __float2int_rd(rbin), c0 = __float2int_rd(cbin);
        int o0 = __float2int_rd(obin);
        rbin -= (float)r0; cbin -= (float)c0; obin -= (float)o0;
        if (o0 < 0) o0 += 8;
        if (o0 >= 8) o0 -= 8;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
        const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
        const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
        const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
        const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
        const int idx = ((r0 + 1) * 6 + c0 + 1) * 10 + o0;
        atomicAdd(&sHist[idx], __float2int_rn(v000 * 4096.0f));
        atomicAdd(&sHist[idx + 1], __float2int_rn(v001 * 4096.0f));
        atomicAdd(&sHist[idx + 10], __float2int_rn(v010 * 4096.0f));
        atomicAdd(&sHist[idx + 11], __float2int_rn(v011 * 4096.0f));
        atomicAdd(&sHist[idx + 60], __float2int_rn(v100 * 4096.0f));
        atomicAdd(&sHist[idx + 61], __float2int_rn(v101 * 4096.0f));
        atomicAdd(&sHist[idx + 70], __float2int_rn(v110 * 4096.0f));
        atomicAdd(&sHist[idx + 71], __float2int_rn(v111 * 4096.0f));
    }
    __syncthreads();
    // circular orientation bins, then normalise / clip / renormalise on the integers (orc_sift_finalize); element e on thread e < 128
    long long h = 0;
    if (tid < 128) {
        const int cell = tid >> 3, o = tid & 7, ci = cell >> 2, cj = cell & 3;
        const int idx = ((ci + 1) * 6 + (cj + 1)) * 10;
        h = sHist[idx + o];
        if (o < 2) h += sHist[idx + 8 + o];
    }
    if (dbg && tid < 128) dbg[((size_t)blockIdx.y * dbg_kcap + k) * 128 + tid] = (int)h;
    unsigned long long s = (unsigned long long)(h * h);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) sSum[tid >> 6] = s;
    __syncthreads();
    const unsigned long long s_all = sSum[0] + sSum[1];
    const double nrm = sqrt((double)s_all);
    const long long thr = (long long)(nrm * 0.2);
    const long long v = h < thr ? h : thr;
    __syncthreads();
    unsigned long long s2 = tid < 128 ? (unsigned long long)(v * v) : 0ull;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    if ((tid & 63) == 0) sSum[tid >> 6] = s2;
    __syncthreads();
    double den = sqrt((double)(sSum[0] + sSum[1]));
    if (den < 1.1920928955078125e-7) den = 1.1920928955078125e-7;
    const double scale = 512.0 / den;
    if (tid < 128) {
        const int rr = __double2int_rn((double)v * scale);
        f.d128tmp[(size_t)k * 128 + tid] = (uint8_t)(rr < 0 ? 0 : rr > 255 ? 255 : rr);
    }
}

void dsss_launch_sift_desc(dsss_ctx* c, hipStream_t st, const ex_frame* d_exf, int kcap, int nb)
{
    // DSSS_SIFT_HIST_DUMP=<file> (diagnostic, tests/test_gpu_sift.py): the raw 2^-12 fixed-point histograms of every pre-filter keypoint of the
    // batch, int32[nb][kcap][128], written after the launch -- the parity bar UNDER the rounded bytes (a one-unit difference in one of a
    // keypoint's 13 000 rounded shares moves an output byte only once in a few thousand keypoints: the bare v_sqrt_f32 behind __fsqrt_rn did)
    int* dbg = nullptr;
    const char* dump = getenv("DSSS_SIFT_HIST_DUMP");
    if (dump && hipMalloc(&dbg, (size_t)nb * kcap * 128 * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); dbg = nullptr; }
    if (dbg) (void)hipMemsetAsync(dbg, 0, (size_t)nb * kcap * 128 * sizeof(int), st);
    hipLaunchKernelGGL(sift_desc_kernel, dim3(kcap, nb), dim3(256), 0, st, d_exf, c->sift_w, dbg, kcap);
    if (dbg) {
        std::vector<int> h((size_t)nb * kcap * 128);
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), dbg, h.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
            FILE* fp = fopen(dump, "wb");
            if (fp) { fwrite(h.data(), sizeof(int), h.size(), fp); fclose(fp); }
        }
        hipFree(dbg);
    }
}
