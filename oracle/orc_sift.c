/* oracle/orc_sift.c -- the 128-float descriptor of the reference's SIFT call site, as its author evidently meant it
 * (SURVEY.md section 8f, N4).  Test infrastructure, see orc.h.
 *
 * What the reference does: ORBextractor::operator() hands every blurred pyramid level and the ORB keypoints found on it to
 * cv::SIFT::compute (/root/reference/thirdparty/ORBextractor.cpp:1043-1047, live call :1098) and the matcher compares the rows
 * of Frame::dst by cv::norm(NORM_L2) with bound 350 and ratio 0.35 (/root/reference/src/core/FEAmatcher.cpp:106-139).  As
 * shipped the result never reaches the caller's buffer and the ORB level is misread as a power-of-two SIFT octave.
 * The intent restated here: Lowe's 4 x 4 x 8 gradient-orientation histogram (OpenCV's calcSIFTDescriptor, restated from its
 * published source: trilinear interpolation, Gaussian weight over the window, normalise / clip at 0.2 / renormalise, x 512
 * saturated to 0..255) on the blurred image of THE LEVEL THE KEYPOINT WAS FOUND ON, in that level's pixels, at the keypoint's
 * IC angle (ori = 360 - kpt.angle as SIFT::compute does).
 *
 * Definitions of this build (OpenCV is not in the image; parity unpinned at its boundary):
 *   - one spatial bin is 8 level pixels wide (hist_width = 3 * scl with scl = 8 / 3): the 4 x 4 bins span 32 px = the 31-px ORB
 *     patch, and dividing by 8 is exact; radius = cvRound(8 * sqrt 2 * 5 / 2) = 28: a 57 x 57 window;
 *   - the Gaussian weight exp(-(c_rot^2 + r_rot^2) / 8) = exp(-(i^2 + j^2) / 512) comes from a table over the integer i^2 + j^2,
 *     built by repeated multiplication with exp(-1/512) in double (the same loop in oracle and product: same bits);
 *   - the histogram is accumulated in 2^-12 FIXED POINT (every trilinear share rounded to an integer, cvRound): integer sums do
 *     not depend on the order of accumulation, which is what lets the device kernel add with LDS atomics and still be
 *     bit-identical;
 *   - normalisation on those integers: thr = trunc(0.2 sqrt(sum h^2)), scale = 512 / max(sqrt(sum min(h, thr)^2), FLT_EPSILON),
 *     out = saturate_u8(cvRound(min(h, thr) * scale)) -- integer-valued floats 0..255, as OpenCV stores them.  Distances between
 *     two such rows are square roots of exact integers (FEAmatcher.cpp:113), see orc_match.c.                                  */
#include "orc.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SD 4
#define SN 8
#define SRAD 28
#define SHW 8.0f
#define SFIX 4096.0f

void orc_sift_weights(float* w, int n)      /* w[k] = exp(-k / 512), k < n */
{
    const double q = 0.9980487811074755;    /* exp(-1/512) rounded to double */
    double v = 1.0;
    for (int k = 0; k < n; ++k) { w[k] = (float)v; v = v * q; }
}

void orc_sift_finalize(const int32_t h[128], uint8_t out[128])
{
    uint64_t s = 0;
    for (int k = 0; k < 128; ++k) s += (uint64_t)((int64_t)h[k] * h[k]);
    const double nrm = sqrt((double)s);
    const int64_t thr = (int64_t)(nrm * 0.2);
    uint64_t s2 = 0;
    for (int k = 0; k < 128; ++k) { const int64_t v = h[k] < thr ? h[k] : thr; s2 += (uint64_t)(v * v); }
    double den = sqrt((double)s2);
    if (den < (double)FLT_EPSILON) den = (double)FLT_EPSILON;
    const double scale = 512.0 / den;
    for (int k = 0; k < 128; ++k) {
        const int64_t v = h[k] < thr ? h[k] : thr;
        const int r = orc_cvround((double)v * scale);
        out[k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
    }
}

void orc_sift_hist(const uint8_t* img, int rows, int cols, int x, int y, float angle_deg, int32_t h[128])
{
    float wtab[2 * SRAD * SRAD + 1];                 /* (per call: the tests run frames on a thread pool; 1 569 multiplications next to 3 249 samples) */
    orc_sift_weights(wtab, 2 * SRAD * SRAD + 1);
    float ori = 360.f - angle_deg;
    if (fabsf(ori - 360.f) < FLT_EPSILON) ori = 0.f;
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    double sd, cd;
    orc_sincos((double)(ori * factorPI), &sd, &cd);
    const float cos_t = (float)cd / SHW, sin_t = (float)sd / SHW;
    const float bins_per_deg = SN / 360.f;
    int32_t hist[(SD + 2) * (SD + 2) * (SN + 2)];
    memset(hist, 0, sizeof hist);
    for (int i = -SRAD; i <= SRAD; ++i)
        for (int j = -SRAD; j <= SRAD; ++j) {
            const float c_rot = (float)j * cos_t - (float)i * sin_t;
            const float r_rot = (float)j * sin_t + (float)i * cos_t;
            float rbin = r_rot + (SD / 2 - 0.5f), cbin = c_rot + (SD / 2 - 0.5f);
            const int r = y + i, c = x + j;
            if (!(rbin > -1 && rbin < SD && cbin > -1 && cbin < SD && r > 0 && r < rows - 1 && c > 0 && c < cols - 1)) continue;
            const float dx = (float)((int)img[(size_t)r * cols + c + 1] - (int)img[(size_t)r * cols + c - 1]);
            const float dy = (float)((int)img[(size_t)(r - 1) * cols + c] - (int)img[(size_t)(r + 1) * cols + c]);
            const float Ori = orc_fast_atan2(dy, dx);
            const float Mag = sqrtf(dx * dx + dy * dy);
            float obin = (Ori - ori) * bins_per_deg;
            const float mag = Mag * wtab[i * i + j * j];
            const int r0 = orc_cvfloorf(rbin), c0 = orc_cvfloorf(cbin);
            int o0 = orc_cvfloorf(obin);
            rbin -= (float)r0; cbin -= (float)c0; obin -= (float)o0;
            if (o0 < 0) o0 += SN;
            if (o0 >= SN) o0 -= SN;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
            const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
            const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
            const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
            const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
            const int idx = ((r0 + 1) * (SD + 2) + c0 + 1) * (SN + 2) + o0;
            hist[idx] += orc_cvroundf(v000 * SFIX);
            hist[idx + 1] += orc_cvroundf(v001 * SFIX);
            hist[idx + (SN + 2)] += orc_cvroundf(v010 * SFIX);
            hist[idx + (SN + 3)] += orc_cvroundf(v011 * SFIX);
            hist[idx + (SD + 2) * (SN + 2)] += orc_cvroundf(v100 * SFIX);
            hist[idx + (SD + 2) * (SN + 2) + 1] += orc_cvroundf(v101 * SFIX);
            hist[idx + (SD + 3) * (SN + 2)] += orc_cvroundf(v110 * SFIX);
            hist[idx + (SD + 3) * (SN + 2) + 1] += orc_cvroundf(v111 * SFIX);
        }
    /* the orientation histogram is circular */
    for (int i = 0; i < SD; ++i)
        for (int j = 0; j < SD; ++j) {
            const int idx = ((i + 1) * (SD + 2) + (j + 1)) * (SN + 2);
            hist[idx] += hist[idx + SN];
            hist[idx + 1] += hist[idx + SN + 1];
            for (int k = 0; k < SN; ++k) h[(i * SD + j) * SN + k] = hist[idx + k];
        }
}

void orc_sift128(const uint8_t* blurred, int rows, int cols, int x, int y, float angle_deg, uint8_t out[128])
{
    int32_t h[128];
    orc_sift_hist(blurred, rows, cols, x, y, angle_deg, h);
    orc_sift_finalize(h, out);
}
