// diasss_amd/csrc/dsss_quadtree.h -- launch descriptors of the device quadtree (dsss_quadtree.hip)
#pragma once
#include <hip/hip_runtime.h>

struct qt_kp_in { float x, y, resp; int level; };

struct qt_inst {                 // one (frame, level)
    const int* offs;             // candidate offsets per FAST cell of the frame (scan_counts_kernel output)
    int cell_begin, cell_end;    // cells of this level
    const float *xs, *ys, *rs;   // candidates of the frame, reference order
    int W, H, quota;             // maxBorderX - minBorderX, maxBorderY - minBorderY, mnFeaturesPerLevel[level]
    unsigned long long *keys0, *keys1;   // frame-wide ping-pong key arrays (candidate capacity each); key = y << 48 | x << 32 | candidate index
    int* work;                   // 8*pool_cap + 10*list_cap ints
    int list_cap, pool_cap;
    int* out_idx; int* out_n; int out_cap;
    int* err;
    int cand_cap;                // capacity of xs / ys / rs / keys0 / keys1 (candidates of the whole frame)
};

struct qt_frame {                // one frame: gathers its levels into kp_in records
    int nlevels, out_cap, kcap, min_border;
    const int* out_idx; const int* out_n;
    const float *xs, *ys, *rs;
    qt_kp_in* kin; int* nk; int* err;
};

void dsss_launch_quadtree(hipStream_t st, const qt_inst* d_inst, int ninst);                      // one workgroup per instance
void dsss_launch_quadtree_collect(hipStream_t st, const qt_frame* d_frames, int nframes);         // after every level of the frames has run
