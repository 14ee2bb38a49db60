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
// One workgroup per keypoint.  LDS: the 71 x 71 raw window of the level image (57 + 2 for the gradients + 12 for the 13-tap blur), the
// horizontal blur pass transposed, the 59 x 59 blurred window, 360 histogram cells.  The separable 8.8 fixed-point blur is the one of
// orient_desc_kernel (oracle/orc_orb.c:orc_blur13: reflect-101 at the borders of the level image, (v + 2^15) >> 16).
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

__global__ __launch_bounds__(256) void sift_desc_kernel(const ex_frame* __restrict__ frs, const float* __restrict__ wtab)
{
    __shared__ __attribute__((aligned(16))) uint8_t sP[RW * RS];
    __shared__ __attribute__((aligned(16))) uint16_t sH[GW * HSS];
    __shared__ __attribute__((aligned(16))) uint8_t sB[GW * GS];
    __shared__ int sHist[NHIST];
    __shared__ unsigned long long sSum[4];
    const ex_frame& f = frs[blockIdx.y];
    const int k = blockIdx.x;
    if (k >= *f.nk) return;
    const qt_kp_in in = f.kin[k];
    const int L = in.level;
    const uint8_t* __restrict__ img = f.lvl[L];
    const int cols = f.cols[L], rows = f.rows[L];
    const int cx = __float2int_rn(in.x), cy = __float2int_rn(in.y);
    const float angle = f.kptmp[k].angle;                       // IC_Angle, written by orient_desc_kernel earlier on this stream
    const int tid = threadIdx.x;
    for (int t = tid; t < NHIST; t += 256) sHist[t] = 0;
    // raw window, reflect-101 at the borders of the level image
    const int R0 = SRAD + 1 + 6;                                 // 35
    for (int t = tid; t < RW * RW; t += 256) {
        const int py = t / RW, px = t - py * RW;
        sP[py * RS + px] = img[(size_t)reflect101_dev(cy + py - R0, rows) * cols + reflect101_dev(cx + px - R0, cols)];
    }
    __syncthreads();
    // horizontal 13-tap pass on rows 0 .. 70, columns 6 .. 64 of the raw window -> H[bx][py], bx < 59
    const int T13[13] = { 1, 2, 7, 16, 31, 45, 52, 45, 31, 16, 7, 2, 1 };
    for (int t = tid; t < RW * GW; t += 256) {
        const int py = t / GW, bx = t - py * GW;
        const uint8_t* p = sP + py * RS + bx;
        unsigned acc = 0;
#pragma unroll
        for (int q = 0; q < 13; ++q) acc += (unsigned)T13[q] * p[q];
        sH[bx * HSS + py] = (uint16_t)acc;
    }
    __syncthreads();
    for (int t = tid; t < GW * GW; t += 256) {
        const int bx = t / GW, by = t - bx * GW;
        const uint16_t* p = sH + bx * HSS + by;
        unsigned acc = 0;
#pragma unroll
        for (int q = 0; q < 13; ++q) acc += (unsigned)T13[q] * p[q];
        sB[by * GS + bx] = (uint8_t)((acc + 32768u) >> 16);
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
    for (int t = tid; t < SW * SW; t += 256) {
        const int si = t / SW, sj = t - si * SW;
        const int i = si - SRAD, j = sj - SRAD;
        const float c_rot = (float)j * cos_t - (float)i * sin_t;
        const float r_rot = (float)j * sin_t + (float)i * cos_t;
        float rbin = r_rot + 1.5f, cbin = c_rot + 1.5f;
        const int r = cy + i, c = cx + j;
        if (!(rbin > -1 && rbin < 4 && cbin > -1 && cbin < 4 && r > 0 && r < rows - 1 && c > 0 && c < cols - 1)) continue;
        const uint8_t* b = sB + (si + 1) * GS + (sj + 1);
        const float dx = (float)((int)b[1] - (int)b[-1]);
        const float dy = (float)((int)b[-GS] - (int)b[GS]);
        const float Ori = fast_atan2_dev(dy, dx);
        const float Mag = __fsqrt_rn(dx * dx + dy * dy);
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
    hipLaunchKernelGGL(sift_desc_kernel, dim3(kcap, nb), dim3(256), 0, st, d_exf, c->sift_w);
}
