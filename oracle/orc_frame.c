/* oracle/orc_frame.c -- Frame preprocessing restated from /root/reference/src/core/frame.cpp
 * and Util::ComputeIntersection (src/util/util.cpp:13-43).  Test infrastructure, see orc.h. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_mask_params_default(orc_mask_params* p) { p->factor = 2.5; p->width = 10; p->r = 6; p->side = 150; }

/* Sum of n doubles in the FIXED order shared with the HIP reduction kernel:
 * 128 strided partials (element j goes to partial j mod 128, accumulated in index order), lane l takes
 * partial 2l + partial 2l+1, then a 64-lane xor butterfly (32,16,8,4,2,1). cv::mean (frame.cpp:62,90)
 * leaves the order unspecified; fixing it makes the 2.5*mean thresholds reproducible bit for bit. */
static double fixed_sum(const double* v, int n)
{
    double part[128];
    for (int j = 0; j < 128; ++j) part[j] = 0.0;
    for (int i = 0; i < n; ++i) part[i & 127] += v[i];
    double lane[64], nxt[64];
    for (int l = 0; l < 64; ++l) lane[l] = part[2 * l] + part[2 * l + 1];
    for (int k = 32; k >= 1; k >>= 1) {
        for (int l = 0; l < 64; ++l) nxt[l] = lane[l] + lane[l ^ k];
        memcpy(lane, nxt, sizeof lane);
    }
    return lane[0];
}

double orc_mean(const double* raw, int N, int M)
{
    double* rows = (double*)malloc(sizeof(double) * (size_t)N);
    for (int i = 0; i < N; ++i) rows[i] = fixed_sum(raw + (size_t)i * M, M);
    double tot = fixed_sum(rows, N);
    free(rows);
    return tot / ((double)N * (double)M);
}

/* Frame::GetNormalizeSSS (frame.cpp:57-81) incl. convertTo(CV_8U) = saturate(cvRound()) */
void orc_normalize(const double* raw, int N, int M, uint8_t* out)
{
    double mean = orc_mean(raw, N, M);
    double max_used = mean * 2.5;
    double mn = raw[0];
    for (size_t i = 1; i < (size_t)N * M; ++i) if (raw[i] < mn) mn = raw[i];
    for (size_t i = 0; i < (size_t)N * M; ++i) {
        double v = (raw[i] - mn) / (max_used - mn) * 255.0;
        if (v > 255.0) v = 255.0;
        int q = orc_cvround(v);
        out[i] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
}

/* Frame::GetFilteredMask (frame.cpp:83-124); eraser clipped to the image (orc.h deviations) */
void orc_mask(const double* raw, int N, int M, const orc_mask_params* p, uint8_t* mask)
{
    double mean = orc_mean(raw, N, M);
    double thr = mean * (double)(float)p->factor;
    memset(mask, 255, (size_t)N * M);
    int r = p->r;
    double sidec = (double)p->side * 0.6;
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < M; ++j) {
            if (raw[(size_t)i * M + j] > thr && i - r >= 0 && j - r >= 0) { /* size_t loops of :100-101 never run for i<r or j<r */
                for (int x = i - r; x < i + r && x < N; ++x)
                    for (int y = j - r; y < j + r && y < M; ++y)
                        mask[(size_t)x * M + y] = 0;
            }
            if (j > M / 2 - p->width && j < M / 2 + p->width) mask[(size_t)i * M + j] = 0;
            if (i < p->side || i > N - p->side) mask[(size_t)i * M + j] = 0;
            if ((double)j < sidec || (double)j > (double)M - sidec) mask[(size_t)i * M + j] = 0;
        }
    }
}

/* Frame::GetGeoImg (frame.cpp:126-165) for one bin. Starboard = cols >= M/2 at yaw+PI/2 with
 * g_range[col-M/2]; port = cols < M/2 at yaw-PI/2 with g_range[M/2-col] (clamped at M/2-1). tf = 0 (:38-39). */
void orc_geo_at(const double* pose6, const double* gr, int N, int M, int row, int col, double* x, double* y)
{
    (void)N;
    const double* P = pose6 + (size_t)row * 6;
    int half = M / 2;
    int idx; double ang;
    if (col >= half) { idx = col - half; ang = P[2] + ORC_PI_REF / 2; }
    else { idx = half - col; if (idx > half - 1) idx = half - 1; ang = P[2] - ORC_PI_REF / 2; }
    double s, c;
    orc_sincos(ang, &s, &c);
    *x = (P[3] - 0.0) + gr[idx] * c;
    *y = (P[4] - 0.0) + gr[idx] * s;
}

void orc_geo_img(const double* pose6, const double* gr, int N, int M, double* gx, double* gy)
{
    int half = M / 2;
    for (int i = 0; i < N; ++i) {
        const double* P = pose6 + (size_t)i * 6;
        double ss, cs, sp, cp;
        orc_sincos(P[2] + ORC_PI_REF / 2, &ss, &cs);
        orc_sincos(P[2] - ORC_PI_REF / 2, &sp, &cp);
        for (int j = 0; j < M; ++j) {
            int idx; double s, c;
            if (j >= half) { idx = j - half; s = ss; c = cs; }
            else { idx = half - j; if (idx > half - 1) idx = half - 1; s = sp; c = cp; }
            gx[(size_t)i * M + j] = (P[3] - 0.0) + gr[idx] * c;
            gy[(size_t)i * M + j] = (P[4] - 0.0) + gr[idx] * s;
        }
    }
}

/* the four cv::minMaxLoc calls over the geo image (FEAmatcher.cpp:71-72, util.cpp:21-26): full scan */
void orc_geo_bbox(const double* pose6, const double* gr, int N, int M, double bbox[4])
{
    double* gx = (double*)malloc(sizeof(double) * (size_t)N * M);
    double* gy = (double*)malloc(sizeof(double) * (size_t)N * M);
    orc_geo_img(pose6, gr, N, M, gx, gy);
    double xmin = gx[0], xmax = gx[0], ymin = gy[0], ymax = gy[0];
    for (size_t i = 1; i < (size_t)N * M; ++i) {
        if (gx[i] < xmin) xmin = gx[i];
        if (gx[i] > xmax) xmax = gx[i];
        if (gy[i] < ymin) ymin = gy[i];
        if (gy[i] > ymax) ymax = gy[i];
    }
    bbox[0] = xmin; bbox[1] = xmax; bbox[2] = ymin; bbox[3] = ymax;
    free(gx); free(gy);
}

/* Util::ComputeIntersection (util.cpp:13-43): float arithmetic on double extrema */
float orc_overlap(const double bs[4], const double bt[4])
{
    float output = 0.0f;
    double sx_min = bs[0], sx_max = bs[1], sy_min = bs[2], sy_max = bs[3];
    double tx_min = bt[0], tx_max = bt[1], ty_min = bt[2], ty_max = bt[3];
    float x_dist_ol = (float)((sx_max < tx_max ? sx_max : tx_max) - (sx_min > tx_min ? sx_min : tx_min));
    float y_dist_ol = (float)((ty_max < sy_max ? ty_max : sy_max) - (sy_min > ty_min ? sy_min : ty_min));
    if (x_dist_ol > 0 && y_dist_ol > 0) {
        float area_ol = x_dist_ol * y_dist_ol;
        float area_s = (float)(fabs(sx_max - sx_min) * fabs(sy_max - sy_min));
        float area_t = (float)(fabs(tx_max - tx_min) * fabs(ty_max - ty_min));
        output = area_ol / (area_s + area_t - area_ol);
    }
    return output;
}
