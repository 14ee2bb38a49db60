// diasss_amd/csrc/dsss_pg_kernels.h -- what the files of the pose-graph solve share: the records the driver fills for the kernels, the
// constants both sides size things with, and the kernels' declarations (a kernel is defined in ONE file and launched from dsss_pg.hip:
// the host-side stub of a __global__ function is an ordinary external symbol, no relocatable device code is involved).
//     dsss_pg_chain.hip    factors and linearisation, per-pose blocks, chain condensation and its back-substitution, scatter into the
//                          reduced system, rank packing, initial values, trajectory rows
//     dsss_pg_bins.hip     the bottom of the elimination tree: index lists built on the device, left-looking block columns per bin
//     dsss_pg_fronts.hip   the multifrontal top: extend-add, panel Cholesky, row solve, trailing update, back-substitution
//     dsss_pg.hip          the solve itself (host): analysis hand-over, LM loop, loop-closure selection, the C ABI
#pragma once
#include "dsss_internal.h"
#include "dsss_pose.h"
#include "dsss_pg_sym.h"

struct pg_weights { double prior[6], odo[6]; };

// update matrices that cross from a rank's interior into the interface (pg_comm_pack_kernel)
struct pg_pack { const double* U; const double* g; double* dst; int cld, cb; };
// a front of the multifrontal top and a child of one (dsss_pg_fronts.hip)
struct pg_front {
    long long off, roff;            // front / right-hand-side arena offsets (doubles)
    int ld, n6, s6, c0;             // leading dimension, scalar rows, own scalar columns, first column (elimination index)
    int rowptr, pan0;               // offset of its block-row list, first global panel id
    int ch0, ch1, fa0, fa1;         // children and original entries (CSR ranges)
};
struct pg_child { const double* U; const double* g; long long relptr; int cld, cb; };

#define PG_ASM_POSES 32                              // poses per workgroup of 192 threads
#define PG_NB4 24                                   // 4-column blocks per panel
#define PG_RSU_MAX_TILES 320
#define PG_RSU32_MAX_TILES 80
#define PG_BWD2_LD 97
#define PG_BWD2_SX 8192                         // rows of x2 the LDS stages; taller fronts read x2 through the row map
#define PG_BWD_SPLIT 2048
#define PG_PARTS_CUT_MAX 12                          // ... and a boundary between two parts may be spanned by at most this many loop closures (every one puts a separator into the interface)
#define PG_PARTS_IFACE_MAX 256                       // one rank analysed by parts (pg_symbolic_parts): the interface between the parts, one dense front, at most this many separators
#define PG_LOCAL_IFACE_MAX 160                       // rank-local analysis (dsss_pg.hip): the interface is ONE dense front of at most this many separators (960 scalar columns, 7.4 MB summed per trial); beyond, the replicated analysis with its interface tree
#define PG_BWD_RC 512
#define RNG_PER_THREAD 16

// ---- dsss_pg_chain.hip
template <bool WJ> __global__ void pg_linearize_kernel(int n, int ne, const pose_t* __restrict__ X, const pose_t* __restrict__ meas, pg_weights W, const int* __restrict__ ea, const int* __restrict__ eb, const int* __restrict__ eo, const pose_t* __restrict__ emeas, const double* __restrict__ ew, double* __restrict__ r, double* __restrict__ Ji, double* __restrict__ partial, int mp0, int mp1);
__global__ void pg_final_sum_kernel(const double* __restrict__ partial, int n, double scale, double* __restrict__ out);
__global__ void pg_assemble_kernel(int n, pg_weights W, const double* __restrict__ r, const double* __restrict__ Ji, const int* __restrict__ adj_ptr, const int* __restrict__ adj_edge, const double* __restrict__ ew, const double* __restrict__ lambda_ptr, double* __restrict__ D, double* __restrict__ C, double* __restrict__ g, const int* __restrict__ eo, int mp0, int mp1, const int* __restrict__ plist, int np);
template <bool FROMJ> __global__ void pg_segment_kernel(int nseg, const int* __restrict__ seg_order, const int* __restrict__ sep_pose, const double* __restrict__ D, const double* __restrict__ C, const double* __restrict__ g, double* __restrict__ E, double* __restrict__ Dl, double* __restrict__ gi, double* __restrict__ segDL, double* __restrict__ segDR, double* __restrict__ segGL, double* __restrict__ segGR, double* __restrict__ segS, int* __restrict__ fail, int mp0, int mp1, const double* __restrict__ rf, const double* __restrict__ Jf, pg_weights W, const double* __restrict__ lambda_ptr);
__global__ void pg_chain1_kernel(int ns1, const int* __restrict__ sep1, const double* __restrict__ D, const double* __restrict__ g, const double* __restrict__ segDL, const double* __restrict__ segDR, const double* __restrict__ segGL, const double* __restrict__ segGR, const double* __restrict__ segS, double* __restrict__ D1, double* __restrict__ C1, double* __restrict__ g1, int mp0, int mp1);
__global__ void pg_scatter_base_kernel(int ns, const int* __restrict__ sep_pose, const int* __restrict__ perm, const double* __restrict__ D, const double* __restrict__ g, const double* __restrict__ segDL, const double* __restrict__ segDR, const double* __restrict__ segGL, const double* __restrict__ segGR, const double* __restrict__ segS, const int* __restrict__ dest, double* __restrict__ Lvals, double* __restrict__ aval, double* __restrict__ rhs, const int* __restrict__ if_slot, double* __restrict__ aval_if, double* __restrict__ x_if, int mp0, int mp1);
__global__ void pg_scatter_lc_kernel(int n, int ne, int ns, const double* __restrict__ Ji, const double* __restrict__ ew, const int* __restrict__ dest, double* __restrict__ Lvals, double* __restrict__ aval, double* __restrict__ aval_if, const int* __restrict__ eo, int mp0, int mp1, const int* __restrict__ lc_link);
__global__ void pg_comm_pack_kernel(const int* __restrict__ it_child, const int* __restrict__ it_row, const pg_pack* __restrict__ PK);
__global__ void pg_comm_xif_kernel(int nif, const int* __restrict__ if_sep, const int* __restrict__ perm, const double* __restrict__ x_if, double* __restrict__ x);
__global__ void pg_comm_scal_kernel(const double* __restrict__ scal, const int* __restrict__ fail, double* __restrict__ red);
__global__ void pg_sep_delta_kernel(int ns, const int* __restrict__ sep_pose, const int* __restrict__ perm, const double* __restrict__ x, double* __restrict__ delta);
template <bool FROMJ> __global__ void pg_backsub_kernel(int nseg, const int* __restrict__ seg_order, const int* __restrict__ sep_pose, const double* __restrict__ C, const double* __restrict__ E, const double* __restrict__ Dl, const double* __restrict__ gi, double* __restrict__ delta, int mp0, int mp1, const double* __restrict__ Jf, pg_weights W);
__global__ void pg_linerr_kernel(int n, int ne, pg_weights W, const int* __restrict__ ea, const int* __restrict__ eb, const int* __restrict__ eo, const double* __restrict__ ew, const double* __restrict__ r, const double* __restrict__ Ji, const double* __restrict__ delta, double* __restrict__ partial, int mp0, int mp1);
__global__ void pg_retract_kernel(int n, const pose_t* __restrict__ X, const double* __restrict__ delta, pose_t* __restrict__ Xn);
__global__ void pg_rng_attempts_kernel(long long nattempts, double* __restrict__ pairs, int* __restrict__ flags);
__global__ void pg_flag_blocksum_kernel(const int* __restrict__ flags, long long n, int* __restrict__ bsum);
__global__ void pg_flag_scan_kernel(int* __restrict__ bsum, int nb, int* __restrict__ total);
__global__ void pg_flag_compact_kernel(const int* __restrict__ flags, const double* __restrict__ pairs, long long n, const int* __restrict__ bsum, long long need_pairs, double* __restrict__ normals);
__global__ void pg_gather_dr_kernel(const unsigned long long* __restrict__ fptr, const int* __restrict__ foff, double* __restrict__ out);
__global__ void pg_gather_pose_kernel(int n, const int* __restrict__ idx, const pose_t* __restrict__ X, pose_t* __restrict__ out);
__global__ void pg_sep_xy_kernel(int ns, const int* __restrict__ sep_pose, const double* __restrict__ dr6, double* __restrict__ xy);
__global__ void pg_init_kernel(int n, const double* __restrict__ dr6, const double* __restrict__ normals, int add_noise, pose_t* __restrict__ X, pose_t* __restrict__ meas);
__global__ void pg_rpy_kernel(int n, const pose_t* __restrict__ X, double* __restrict__ rpy6);
// ---- dsss_pg_bins.hip
__global__ void pg_build_map_kernel(int nupd, const int* __restrict__ rlrow, const int* __restrict__ rlptr, const int* __restrict__ rlcol, const int* __restrict__ rlpos, const int* __restrict__ colptr, const int* __restrict__ rowidx, const long long* __restrict__ mapptr, int* __restrict__ upd_map, const int* __restrict__ nupd_dev);
__global__ void pg_rl_count_kernel(int ns, const int* __restrict__ colptr, const int* __restrict__ rowidx, const char* __restrict__ binned, int* __restrict__ cnt);
template <typename T, int VAL>
__global__ void pg_scan_block_kernel(int n, const int* __restrict__ cnt, const int* __restrict__ colptr, T* __restrict__ out, T* __restrict__ block_sum);
template <typename T>
__global__ void pg_scan_tops_kernel(int nblocks, T* __restrict__ block_sum, T* __restrict__ total);
template <typename T>
__global__ void pg_scan_add_kernel(int n, T* __restrict__ out, const T* __restrict__ block_sum, const T* __restrict__ total);
__global__ void pg_rl_fill_kernel(int ns, const int* __restrict__ colptr, const int* __restrict__ rowidx, const char* __restrict__ binned, const int* __restrict__ rlptr, int* __restrict__ cur, int* __restrict__ rlcol, int* __restrict__ rlpos);
__global__ void pg_rl_sort_kernel(int ns, const int* __restrict__ rlptr, int* __restrict__ rlcol, int* __restrict__ rlpos, int* __restrict__ rlrow, int* __restrict__ fail);
__global__ void pg_fill_map_kernel(int* __restrict__ upd_map, const long long* __restrict__ total);
__global__ void pg_anc_rel_kernel(int ns, const int* __restrict__ colptr, const int* __restrict__ rowidx, const char* __restrict__ binned, const int* __restrict__ root_of, int* __restrict__ anc_first, int* __restrict__ anc_rel);
__global__ void pg_factor_subtree_kernel(const int* __restrict__ bin_perm, const int* __restrict__ binptr, const int* __restrict__ bincols, const int* __restrict__ colptr, const int* __restrict__ rlptr, const int* __restrict__ rlcol, const int* __restrict__ rlpos, const long long* __restrict__ mapptr, const int* __restrict__ upd_map, double* __restrict__ Lvals, double* __restrict__ x, int* __restrict__ fail, const int* __restrict__ binroot_ptr, const int* __restrict__ binroot_idx, const int* __restrict__ broot_b, const long long* __restrict__ broot_uoff, const int* __restrict__ broot_of_col, const int* __restrict__ anc_first, const int* __restrict__ anc_rel, double* __restrict__ ubin, double* __restrict__ rdiag);
__global__ void pg_bwd_subtree_kernel(const int* __restrict__ bin_perm, const int* __restrict__ binptr, const int* __restrict__ bincols, const int* __restrict__ colptr, const int* __restrict__ rowidx, const double* __restrict__ Lvals, double* __restrict__ x, const double* __restrict__ rdiag);
// ---- dsss_pg_fronts.hip
__global__ void pg_front_asm_kernel(const int* __restrict__ it_front, const int* __restrict__ it_row, const pg_front* __restrict__ FD, const pg_child* __restrict__ CH, const int* __restrict__ rel, const int* __restrict__ xr_ptr, const int* __restrict__ xr_child, const int* __restrict__ xr_row, const int* __restrict__ fa_rowptr, const int* __restrict__ fa_src, const int* __restrict__ fa_col, const int* __restrict__ fa_tr, const double* __restrict__ aval, const double* __restrict__ x, double* __restrict__ F, double* __restrict__ R);
__global__ void pg_front_syrk_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD, const int* __restrict__ tile_item, const int* __restrict__ tile_ij, double* __restrict__ F);
__global__ void pg_front_diag4_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD, double* __restrict__ F, double* __restrict__ R, int* __restrict__ fail, double* __restrict__ Tinv);
__global__ void pg_front_trsm2_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD, double* __restrict__ F, double* __restrict__ R, const double* __restrict__ Tinv);
template <int TS>
__global__ void pg_front_rsu_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD, const int* __restrict__ tile_item, const int* __restrict__ tile_ij, double* __restrict__ F, double* __restrict__ FL, double* __restrict__ R, const double* __restrict__ Tinv);
__global__ void pg_front_bwd_part_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD, const int* __restrict__ f_rows, const double* __restrict__ FL, const double* __restrict__ x, double* __restrict__ part, int maxchunks);
__global__ void pg_front_bwd2_kernel(const int* __restrict__ it_front, const int* __restrict__ it_step, const pg_front* __restrict__ FD, const int* __restrict__ f_rows, const double* __restrict__ F, const double* __restrict__ FL, const double* __restrict__ R, double* __restrict__ x, const double* __restrict__ Tinv, const double* __restrict__ part, int maxchunks);
