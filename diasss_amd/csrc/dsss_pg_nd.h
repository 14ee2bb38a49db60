// diasss_amd/csrc/dsss_pg_nd.h -- nested dissection of the reduced pose graph on the device (dsss_pg_nd.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

struct dsss_ctx;

// a node set of the recursion, heap numbering (root 1, children 2 h and 2 h + 1): its slice [lo, lo + size) of the level's node list,
// the first position `out` of its range of the elimination order; kind 0 = not reached, 1 = leaf or degenerate cut (index order),
// 2 = split into nA | nB | separator
struct pg_nd_set { int lo, size, out, kind, nA, nB; };

struct pg_nd_buffers {
    int n = 0, nlev = 0;
    const double* sxy = nullptr;                       // [n][2] separator coordinates (device)
    int* edges = nullptr;                              // [nedges][2]
    int *deg = nullptr, *adj_ptr = nullptr, *adj_cur = nullptr, *adj_idx = nullptr;      // [n], [n + 1], [n], [2 nedges]
    int *rank_x = nullptr, *rank_y = nullptr, *perm0 = nullptr, *perm1 = nullptr, *setid = nullptr, *order = nullptr;      // [n] each
    unsigned char *cut0 = nullptr, *cut1 = nullptr;    // [n]
    pg_nd_set* sets = nullptr;                         // [2^(nlev + 1)]
    int* h_order = nullptr; pg_nd_set* h_sets = nullptr;      // page-locked: [n], [64]
    hipEvent_t done = nullptr;
};

int pg_nd_levels(int n, int leaf);                     // launches needed for n nodes (one spare)
size_t pg_nd_set_count(int nlev);
int pg_nd_start(dsss_ctx* c, hipStream_t st, const pg_nd_buffers& B, const int* redges_host, int nedges, int leaf, int both_axes);
