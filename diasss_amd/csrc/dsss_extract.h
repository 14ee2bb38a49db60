// diasss_amd/csrc/dsss_extract.h -- the batch table of the extraction stages (dsss_extract.hip, dsss_sift.hip) and two device helpers they share.
#pragma once
#include "dsss_internal.h"
#include "dsss_quadtree.h"

// ------------------------------------------------------------------ batch table
// Every stage is ONE launch for a whole batch of frames: blockIdx.y (or .z) is the slot, and this per-slot record
// (uploaded once per batch) carries the frame's buffers and sizes.  Frames of different sizes share a launch; the
// grid is sized for the largest and the rest exit early.
struct fast_cell { int level, x0, y0, w, h, offx, offy, pad; };
struct resize_xtab { int sx; short a0, a1; };     // cv::resize tables: source column (sx + 1 is read only when a1 != 0), 11-bit weights
struct resize_ytab { int ya, yb; short b0, b1; };
struct ex_frame {
    const double* raw; int N, M;
    double* rowsum; double* rowmin; double* stats;
    uint8_t* mask; uint8_t* lvl[DSSS_MAX_LEVELS]; int rows[DSSS_MAX_LEVELS], cols[DSSS_MAX_LEVELS]; int nlevels;
    const fast_cell* cells; int ncells, cell_cap;
    int cell_begin[DSSS_MAX_LEVELS + 1];      // cells of level l: [cell_begin[l], cell_begin[l + 1])
    uint32_t* cand; int* counts; int* offs; float* xs; float* ys; float* rs; int cand_cap;
    const qt_kp_in* kin; const int* nk; const int* lrows; const float* lscale; dsss_kp* kptmp; uint8_t* dtmp;
    uint8_t* d128tmp; uint8_t* d128out;      // DSSS_DESC_SIFT128: pre-filter rows / the frame's rows of the store (null otherwise)
    const double* pose6; const double* gr;
    dsss_kp* kout; uint8_t* dout; double* geo; int* count;
    int* err;                      // per-slot error flag (shared with the quadtree descriptors)
    const resize_xtab* xt[DSSS_MAX_LEVELS]; const resize_ytab* yt[DSSS_MAX_LEVELS];      // per level l >= 1: tables of the resize from l - 1
};

__device__ inline float fast_atan2_dev(float y, float x)        // cv::fastAtan2 (ORBextractor.cpp:103)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + eps); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + eps); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__device__ inline int reflect101_dev(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

