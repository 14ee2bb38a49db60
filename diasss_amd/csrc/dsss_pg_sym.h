// diasss_amd/csrc/dsss_pg_sym.h -- host-side analysis of the reduced pose-graph system (no HIP in here).
//
// The reduced system of the batch LM (dsss_pg.hip) is a sparse symmetric matrix of 6x6 blocks over the "separators"
// (LC-touched poses + a few chain poses).  This module decides how it is factorised:
//   1. ordering     nested dissection; the first levels split by RANK PARTITION (contiguous frame blocks, one per GPU),
//                   below that by geometry (recursive coordinate bisection of the DR positions);
//   2. structure    column structures of L and the elimination tree;
//   3. bottom       whole subtrees with little work are binned: one workgroup factorises a bin column by column
//                   (left-looking, block-sparse) and leaves the UPDATE MATRIX of every subtree root for the level above;
//   4. top          multifrontal: every fundamental supernode of the remaining columns is a dense FRONT
//                   [ F11 (its s columns)  .  ]   assembled from the original entries and the update matrices of its children
//                   [ F21 (boundary rows) F22 ]   (extend-add), then partially factorised in 96-column panels on the f64
//                                                 matrix cores; F22 is its own update matrix for the parent front;
//   5. schedule     panel steps of all fronts grouped into levels (a front starts when its children are done);
//   6. ranks        every column belongs to one rank's interior or to the replicated INTERFACE (the partition-level
//                   separators); the update matrices that cross from an interior into the interface are packed into one
//                   buffer and summed over the ranks (the reduced-Hessian all-reduce).
// Replaces GTSAM's ordering + symbolic factorisation inside ISAM2 / LevenbergMarquardtOptimizer
// (/root/reference/src/core/optimizer.cpp:134-139, 265-279).
#pragma once
#include <functional>
#include <vector>

// fn(t) for t in [0, T) on the process-wide worker pool of the analysis (threads that outlive the call: no thread is created per
// use); returns when every t is done.  The caller runs t = 0 itself.
void dsss_pool_run(int T, const std::function<void(int)>& fn);

#define PG_PW 16                        // panel width in block columns (96 scalar columns)

struct pg_sym {
    int ns = 0, nparts = 1;
    std::vector<int> perm;              // chain-order separator -> elimination index
    std::vector<int> order;             // elimination index -> chain-order separator
    std::vector<int> colptr, rowidx;    // block structure of L by column (ascending rows, first the diagonal)
    std::vector<int> parent;            // elimination tree
    std::vector<int> col_part;          // rank that eliminates the column, or -1 for interface (replicated) columns
    // ---- bottom: bins of whole subtrees, left-looking block columns
    std::vector<char> binned;
    std::vector<int> binptr, bincols;   // columns of every bin, ascending
    std::vector<int> bin_part;          // rank of every bin
    std::vector<double> bin_work;       // work of every bin (the packing's estimate; pg_symbolic_parts orders the bins of all parts by it)
    std::vector<int> bin_perm;          // launch order of the bins: within every rank's range by descending work (the kernel's time is its longest
                                        // bin plus whatever starts late: long bins first, the short ones fill in behind them)
    std::vector<int> rlptr, rlcol, rlpos, rlrow;   // per target column: source columns k < j with L(j,k) != 0 and the position of that block (binned targets only)
    std::vector<long long> mapptr;      // update map offsets (binned targets only; see pg_build_map_kernel)
    std::vector<int> broot;             // subtree roots inside bins that have ancestors outside (they hand an update matrix up)
    std::vector<int> broot_of_col;      // per column: index into broot of its subtree root, or -1
    std::vector<int> root_of;           // per binned column: its subtree root (column), -1 for the others
    std::vector<int> broot_b;           // boundary size in blocks = |struct(root)| - 1
    std::vector<long long> broot_uoff;  // offset (doubles) of U_root in the bin-update arena: (6b)^2 row-major, then 6b of right-hand side
    std::vector<int> binroot_ptr, binroot_idx;     // roots per bin (CSR into broot)
    std::vector<int> anc_first;         // per binned column: first position (within the column) whose row lies beyond its subtree root
    std::vector<int> anc_rel;           // per L position of binned columns: index of that row in the root's boundary list, or -1
    long long ubin_doubles = 0;
    // ---- top: fronts
    std::vector<int> f_c0, f_s, f_n;    // first column, number of own block columns, number of block rows (= |struct(c0)|)
    std::vector<int> f_rowptr, f_rows;  // block rows of every front (elimination indices): its own columns first, then the boundary, ascending
    std::vector<int> f_ld;              // leading dimension in scalars (6 n rounded up to 16)
    std::vector<long long> f_off;       // offset (doubles) of the ld x ld front in the front arena; right-hand side at f_roff
    std::vector<long long> f_roff;
    std::vector<int> f_parent, f_part;  // parent front (-1 root), rank (-1 interface)
    std::vector<int> f_level0, f_npan, f_pan0;     // first panel level, number of panel steps, first global panel id
    std::vector<int> front_of_col;      // per column: front index or -1 (binned)
    long long front_doubles = 0, frhs_doubles = 0;
    // children of every front (CSR), in the fixed order the extend-add sums them: kind 0 = front, 1 = bin root
    std::vector<int> ch_ptr, ch_kind, ch_id;
    std::vector<long long> ch_relptr;   // offset into rel of the child's boundary -> parent row map
    std::vector<int> rel;
    // original entries of every front (CSR, sorted by destination row): A-value index, destination block row / column, transpose
    std::vector<int> fa_ptr, fa_src, fa_row, fa_col, fa_tr;
    // where the assembled blocks go: value index k in [0, ns) diagonal of separator k, [ns, 2ns-1) chain coupling k|k+1,
    // then LC edges.  dest_bin[v] = (Lvals position << 1 | transpose) for a binned destination column, -1 for a front
    std::vector<int> dest_bin;
    // per front block row (index f_rowptr[f] + R): the original entries that land in it (fa_rowptr: CSR into fa_*), and the
    // child rows that extend-add into it, in child order (xr_ptr: CSR into xr_child = index into ch_*, xr_row = child row)
    std::vector<int> fa_rowptr, xr_ptr, xr_child, xr_row;
    // ---- schedule: panel steps per level, fronts to assemble per level
    int nlev = 0, npanels = 0;
    std::vector<int> lv_ptr, lv_front, lv_step;
    std::vector<int> asm_ptr, asm_front;
    // ---- ranks: children that cross from an interior front / bin root into an interface front ("comm children")
    std::vector<int> comm_kind, comm_id, comm_part;  // the crossing children in a fixed global order
    std::vector<long long> comm_off;    // packed offset (doubles) in the update-matrix part of the comm buffer: (6b)^2 + 6b each
    long long comm_doubles = 0;
    // original values whose destination is an interface front (summed over the ranks too): value index list; their
    // dest_bin code is -2 - slot and the fronts read them at value index nval + slot.  Interface separators (chain order).
    std::vector<int> comm_vals, iface_seps;
    int nval = 0;
    // statistics
    double flops_factor = 0, flops_fronts = 0; long long nnzL = 0; int max_front_n = 0;
    // ownership invariant of the partitioned numeric phase: a separator with a neighbour of a higher rank must be interface (its
    // diagonal block is summed over the ranks).  Counted by pg_symbolic, must be 0.
    int ownership_violations = 0;
};

// launch lists of a subset of the fronts (all of them, one rank's interior, or the interface), level by level
struct pg_sched {
    int nlev = 0;
    std::vector<int> lv_ptr, lv_front, lv_step;                // panel steps per level
    std::vector<int> asmrow_ptr, asmrow_front, asmrow_row;     // (front, block row) work items of the assembly per level
    std::vector<int> tile_ptr, tile_item, tile_ij;             // (panel item within its level, ti << 16 | tj) 64 x 64 trailing-update tiles per level
    std::vector<int> trsm_chunks;                              // 64-row chunks below the panel, maximum over the level's items
    std::vector<double> fl_diag, fl_trsm, fl_syrk, fl_bwd;     // algorithmic flops per level
    std::vector<int> max_w6, max_rows;                         // per level: scalar columns of its widest panel step, scalar rows below its tallest one (the dependent chains of the level)
    int max_n6 = 6;
};
// part_lo <= front part < part_hi selects fronts; (-1, 0) selects the interface fronts
void pg_build_schedule(const pg_sym& S, int part_lo, int part_hi, pg_sched& out);

struct pg_sym_opts {
    int leaf = 24;                      // nested-dissection leaf size
    int nd_both_axes = 64;              // node sets of at least this size try the median cut along both axes and keep the smaller separator
    int nd_index_cuts = 1;              // a third cut candidate of every node set: the cheapest cut of the CHAIN ORDER with both sides between a third and two thirds of the set (dsss_pg_sym.cpp, nd_order).  0 = coordinate medians only: the ordering of rounds 2 - 4
    bool nd_geo_first = true;           // a node set that spans several ranks may take a geometric cut when its separator is smaller than the rank cut's (its separator is interface then)
    double bin_cost = 1000;             // work bound of a binned subtree
    double pack_cost = 0;               // work bound of a BIN (several subtrees packed together); 0 = bin_cost
    int threads = 4;
    // relaxed amalgamation of a front into its parent (columns adjacent): accepted when it adds at most relax_zero_blocks
    // zero blocks, or when the merged front costs at most relax_flops x the two separate ones (relax_flops_small while the
    // merged front still fits one 96-column panel: such a merge removes a whole level for very little arithmetic)
    double relax_zero_blocks = 8, relax_flops = 1.05, relax_flops_small = 1.6;
    double relax_abs_flops = 0;         // extra flops a merge may cost when it removes a panel step
    // called (on the analysing thread) as soon as everything the BOTTOM of the tree needs is final -- ordering, column structures, bins,
    // update lists, dest_bin -- while the fronts and the schedule are still to come: the caller can upload those tables and run the
    // scatter and the bins of the first trial under the rest of the analysis.  Only with one partition (interface values get their
    // dest_bin codes at the very end).  None of the vectors it may read is touched afterwards.
    std::function<void()> on_bottom_ready;
    // called earlier still (same conditions), as soon as the column structures, the binned flags and the subtree roots are final: all the
    // device needs to build the bins' update lists, update map and root-boundary indices itself (lists_on_device) while the host packs the bins
    std::function<void()> on_lists_ready;
    bool lists_on_device = false;                // the caller builds the update lists, the update-map offsets and the root-boundary indices of the bins itself (dsss_pg.hip: on the device); rlptr .. anc_rel stay empty
    std::function<void()> before_order;          // called once the adjacency is built, before the first use of the coordinates (which may still be on their way)
    // RANK-LOCAL analysis (dsss_pg.hip, several ranks): the graph handed in is one rank's own separators plus the INTERFACE nodes listed
    // here (ascending node ids).  They are eliminated last, in this order, as ONE dense front whatever their edges (every rank lays the
    // interface out the same way: the front is summed over the ranks as it is); the nested dissection orders the other nodes only.
    // Their columns get col_part -1, their values the interface codes of dest_bin; no update matrix is packed for a collective.
    std::vector<int> iface_last;
    bool to_be_joined = false;          // pg_symbolic_parts: the panel levels and the per-row views of the fronts are built once, on the joined tables
    bool iface_plain = false;           // with iface_last: the interface values stay ordinary values of the value array (no summed slots): one rank analysing by parts (pg_symbolic_parts)
};

void pg_sym_opts_env(pg_sym_opts& opt);    // DSSS_PG_ND_BOTH / DSSS_PG_LEAF overrides (analysis knobs kept for tools/pg_sweep.sh)

// edges: pairs of chain-order separator indices, the ns-1 chain couplings (k, k+1) first, then the LC edges.
// part[k] (may be null): rank that owns separator k, non-decreasing in k.  cx, cy: DR positions of the separators.
void pg_symbolic(int ns, const std::vector<std::pair<int, int>>& edges, int nchain, const double* cx, const double* cy,
                 const int* part, int nparts, const pg_sym_opts& opt, pg_sym& S);

// ONE rank, analysed BY PARTS (round 6).  part[k]: part of separator k, non-decreasing, K parts.  The interface -- the separators with a
// neighbour in a higher part -- is prescribed as the last, dense front; every part's own separators + the interface are ordered and analysed
// independently of the other parts, all parts at the same time on the worker pool, and the results are joined into ONE ordinary
// single-partition pg_sym (columns [part 0][part 1]...[interface], arenas and index spaces behind each other, the interface front taking
// the children of all parts).  The phases of pg_symbolic do not speed up with threads at C3's size (a dozen fork / joins of 0.1 ms of work
// each: one thread 4.0 ms, eight 3.4); whole parts do.  Returns false (S untouched) when the interface is wider than max_iface separators.
bool pg_symbolic_parts(int ns, const std::vector<std::pair<int, int>>& edges, int nchain, const double* cx, const double* cy,
                       const int* part, int K, int max_iface, const pg_sym_opts& opt, pg_sym& S);

// Host twin of the numeric phase, used by the CPU test-suite only (the product path is dsss_pg.hip): factorises the matrix
// given by `aval` (36 doubles per value index, see dest_bin) and solves for `rhs` (6 per separator, chain order).  Returns 0
// or -1 when a pivot is not positive.
int pg_host_solve(const pg_sym& S, int ne, const std::vector<std::pair<int, int>>& edges, const double* aval, const double* rhs, double* x);
