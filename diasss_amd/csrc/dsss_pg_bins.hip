// diasss_amd/csrc/dsss_pg_bins.hip -- the BOTTOM of the elimination tree of the reduced pose-graph system (dsss_pg_sym.h): whole subtrees
// binned into one workgroup each.  First the kernels that build the bins' index tables on the device (update lists, update map, root
// boundary indices), then the numeric kernels (left-looking block columns, backward substitution).
// These kernels are compared with the oracle at 1e-6 on the poses, not bit for bit, so they may fuse multiply-adds; everything else in
// the library stays at -ffp-contract=off.
#include "dsss_pg_kernels.h"
#include "dsss_pg_dev.h"
#pragma clang fp contract(fast)
// ---- sparse block Cholesky of the binned columns, left-looking.
// Column j holds blocks L(i, j), i in rowidx[colptr[j] .. colptr[j+1]) ascending, first the diagonal.
// rowlist(j) = binned columns k < j with L(j, k) != 0 and the position of that block.
// upd_map (built once per solve by pg_build_map_kernel): for update t of column j and target block q the position of
// L(i_q, k_t) or -1; layout [mapptr[j] + t * m_j + q], so the factor kernel has no dependent index search.
__global__ __launch_bounds__(256) void pg_build_map_kernel(int nupd, const int* __restrict__ rlrow, const int* __restrict__ rlptr,
                                                           const int* __restrict__ rlcol, const int* __restrict__ rlpos,
                                                           const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                           const long long* __restrict__ mapptr, int* __restrict__ upd_map, const int* __restrict__ nupd_dev)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= (nupd_dev ? *nupd_dev : nupd)) return;             // (lists built on the device: their total stays there, the grid covers the bound)
    const int j = rlrow[e], k = rlcol[e];
    const int c0 = colptr[j], m = colptr[j + 1] - c0;
    int* mp = upd_map + mapptr[j] + (long long)(e - rlptr[j]) * m;
    int q = 0;
    for (int p = rlpos[e]; p < colptr[k + 1]; ++p) {           // rows of column k from row j downwards: a subset of column j's rows
        const int i = rowidx[p];
        while (q < m && rowidx[c0 + q] < i) ++q;
        if (q < m && rowidx[c0 + q] == i) mp[q] = p;
    }
}

// ---- The bins' index tables built ON THE DEVICE (round 3): the update lists (per target column j the binned source columns k < j
// with L(j, k) != 0, ascending, and the position of that block), the offsets of the update map, and for the rows of a binned column
// beyond its subtree root their index in the root's boundary list.  On the host they were the last 0.9 ms of the analysis before the
// first trial could touch the bins (35 ms at the 4 M-pose graph of config 5); they are independent of everything the analysis does
// afterwards.  Same content, same order (a list is sorted by source column), so the factor is bit-identical to the host-built one
// (DSSS_PG_LISTS=host keeps the host path; the host twin of the CPU tests always uses it).
__global__ __launch_bounds__(256) void pg_rl_count_kernel(int ns, const int* __restrict__ colptr, const int* __restrict__ rowidx, const char* __restrict__ binned,
                                                          int* __restrict__ cnt)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns || !binned[k]) return;
    for (int q = colptr[k] + 1; q < colptr[k + 1]; ++q) { const int j = rowidx[q]; if (binned[j]) atomicAdd(&cnt[j], 1); }
}
// exclusive scan in three launches (1024 values per workgroup, up to 1024 x 1024 values): VAL 0 = the counts themselves (int),
// VAL 1 = count x blocks of the column (long long: the update map holds one entry per (update, target block))
template <typename T, int VAL>
__global__ __launch_bounds__(1024) void pg_scan_block_kernel(int n, const int* __restrict__ cnt, const int* __restrict__ colptr, T* __restrict__ out, T* __restrict__ block_sum)
{
    __shared__ T s_w[16];
    const int i = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T v = 0;
    if (i < n) v = VAL == 0 ? (T)cnt[i] : (T)cnt[i] * (T)(colptr[i + 1] - colptr[i]);
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const T t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const T t = s_w[k]; if (k < w) base += t; tot += t; }
    if (i < n) out[i] = base + inc - v;
    if (threadIdx.x == 0) block_sum[blockIdx.x] = tot;
}
template <typename T>
__global__ __launch_bounds__(1024) void pg_scan_tops_kernel(int nblocks, T* __restrict__ block_sum, T* __restrict__ total)
{
    __shared__ T s_w[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const T v = i < nblocks ? block_sum[i] : (T)0;
        T inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const T t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        __syncthreads();
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        T base = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const T t = s_w[k]; if (k < w) base += t; tot += t; }
        if (i < nblocks) block_sum[i] = carry + base + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}
template <typename T>
__global__ __launch_bounds__(1024) void pg_scan_add_kernel(int n, T* __restrict__ out, const T* __restrict__ block_sum, const T* __restrict__ total)
{
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += block_sum[blockIdx.x];
    if (i == 0) out[n] = *total;
}
__global__ __launch_bounds__(256) void pg_rl_fill_kernel(int ns, const int* __restrict__ colptr, const int* __restrict__ rowidx, const char* __restrict__ binned,
                                                         const int* __restrict__ rlptr, int* __restrict__ cur, int* __restrict__ rlcol, int* __restrict__ rlpos)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns || !binned[k]) return;
    for (int q = colptr[k] + 1; q < colptr[k + 1]; ++q) {
        const int j = rowidx[q];
        if (!binned[j]) continue;
        const int at = rlptr[j] + atomicAdd(&cur[j], 1);          // any order: pg_rl_sort_kernel sorts the list by source column
        rlcol[at] = k; rlpos[at] = q;
    }
}
// one wavefront per target column: its list sorted by source column (the keys are distinct: rank = number of smaller keys)
#define PG_RL_SORT_CAP 1024
__global__ __launch_bounds__(256) void pg_rl_sort_kernel(int ns, const int* __restrict__ rlptr, int* __restrict__ rlcol, int* __restrict__ rlpos, int* __restrict__ rlrow,
                                                         int* __restrict__ fail)
{
    __shared__ int s_k[4][PG_RL_SORT_CAP], s_q[4][PG_RL_SORT_CAP];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wv;
    if (j >= ns) return;
    const int b = rlptr[j], len = rlptr[j + 1] - b;
    if (len <= 0) return;
    if (len > PG_RL_SORT_CAP) { if (lane == 0) *fail = 2; return; }      // (a bin's lists sum to a few hundred entries: cannot happen; never silent)
    for (int e = lane; e < len; e += 64) { s_k[wv][e] = rlcol[b + e]; s_q[wv][e] = rlpos[b + e]; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < len; e += 64) {
        const int key = s_k[wv][e];
        int rank = 0;
        for (int f = 0; f < len; ++f) rank += s_k[wv][f] < key;
        rlcol[b + rank] = key; rlpos[b + rank] = s_q[wv][e]; rlrow[b + rank] = j;
    }
}
__global__ __launch_bounds__(256) void pg_fill_map_kernel(int* __restrict__ upd_map, const long long* __restrict__ total)
{
    const long long n = *total;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) upd_map[i] = -1;
}
__global__ __launch_bounds__(256) void pg_anc_rel_kernel(int ns, const int* __restrict__ colptr, const int* __restrict__ rowidx, const char* __restrict__ binned,
                                                         const int* __restrict__ root_of, int* __restrict__ anc_first, int* __restrict__ anc_rel)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ns || !binned[k]) return;
    const int r = root_of[k], c0 = colptr[k], m = colptr[k + 1] - c0;
    int q = 0;
    while (q < m && rowidx[c0 + q] <= r) ++q;
    anc_first[k] = q;
    const int* __restrict__ rb = rowidx + colptr[r] + 1; const int nb = colptr[r + 1] - colptr[r] - 1;
    int w = 0;
    for (; q < m; ++q) { const int row = rowidx[c0 + q]; while (w < nb && rb[w] < row) ++w; anc_rel[c0 + q] = (w < nb && rb[w] == row) ? w : -1; }
}

#define PG_TCH 128
// ---- bottom of the elimination tree: whole subtrees per workgroup (no grid-wide level barriers).
// A bin is a list of columns in ascending order whose sources all lie in the same bin, so the workgroup can run them back to
// back with workgroup barriers only (left-looking, block-sparse, update map).  Only columns with at most 42 blocks
// (6m <= 256 rows: one pass) are binned.  Every finished column also adds its outer product over the rows BEYOND its subtree
// root to the root's update matrix U_root (and L y to its right-hand side part): what the first front above the bin
// extend-adds, exactly like the F22 of a child front.  One workgroup owns a bin, columns in fixed order: deterministic.
// acc[s] += sum_t L(i,k_t)[r][:] . L(j,k_t)[s][:] for the tn updates staged in LDS.  Updates that do not touch this
// row (map entry -1) are masked instead of skipped, and four updates are in flight at once, so the index load and the
// six operand loads of different updates overlap instead of forming one dependent chain per update.
__device__ inline void pg_acc_rows(const int* __restrict__ mp, int m, int tn, const double* __restrict__ Lvals, int r,
                                   const double* __restrict__ s_Ljk, double* acc)
{
    int t = 0;
    for (; t + 4 <= tn; t += 4) {
        int pos[4]; double a[4][6];
#pragma unroll
        for (int u = 0; u < 4; ++u) pos[u] = mp[(size_t)(t + u) * m];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double* Lik = Lvals + (size_t)(pos[u] < 0 ? 0 : pos[u]) * 36 + r * 6;
#pragma unroll
            for (int c = 0; c < 6; ++c) a[u][c] = Lik[c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (pos[u] < 0) continue;
            const double* B = s_Ljk + (t + u) * 36;
#pragma unroll
            for (int s = 0; s < 6; ++s)
                acc[s] += a[u][0] * B[s * 6] + a[u][1] * B[s * 6 + 1] + a[u][2] * B[s * 6 + 2] + a[u][3] * B[s * 6 + 3] + a[u][4] * B[s * 6 + 4] + a[u][5] * B[s * 6 + 5];
        }
    }
    for (; t < tn; ++t) {
        const int pos = mp[(size_t)t * m];
        if (pos < 0) continue;
        const double* Lik = Lvals + (size_t)pos * 36 + r * 6;
        const double* B = s_Ljk + t * 36;
        const double a0 = Lik[0], a1 = Lik[1], a2 = Lik[2], a3 = Lik[3], a4 = Lik[4], a5 = Lik[5];
#pragma unroll
        for (int s = 0; s < 6; ++s)
            acc[s] += a0 * B[s * 6] + a1 * B[s * 6 + 1] + a2 * B[s * 6 + 2] + a3 * B[s * 6 + 3] + a4 * B[s * 6 + 4] + a5 * B[s * 6 + 5];
    }
}
__global__ __launch_bounds__(256) void pg_factor_subtree_kernel(const int* __restrict__ bin_perm, const int* __restrict__ binptr, const int* __restrict__ bincols,
                                                                const int* __restrict__ colptr, const int* __restrict__ rlptr,
                                                                const int* __restrict__ rlcol, const int* __restrict__ rlpos,
                                                                const long long* __restrict__ mapptr, const int* __restrict__ upd_map,
                                                                double* __restrict__ Lvals, double* __restrict__ x, int* __restrict__ fail,
                                                                const int* __restrict__ binroot_ptr, const int* __restrict__ binroot_idx,
                                                                const int* __restrict__ broot_b, const long long* __restrict__ broot_uoff,
                                                                const int* __restrict__ broot_of_col, const int* __restrict__ anc_first,
                                                                const int* __restrict__ anc_rel, double* __restrict__ ubin, double* __restrict__ rdiag)
{
    __shared__ double s_Ljk[PG_TCH * 36];          // update staging; reused for the column's ancestor blocks (42 x 36)
    __shared__ double s_yk[PG_TCH * 6];
    __shared__ double s_diag[36];
    __shared__ int s_arel[48];                     // boundary indices of the column's ancestor rows (at most 42)
    __shared__ double s_ri[6];                     // 1 / L(j, j)[a][a]: the solves below multiply (a dependent f64 division is ~15 instructions)
    __shared__ double s_y[6];
    __shared__ double s_xj[6];
    __shared__ int s_ok;
    const int bin = bin_perm[blockIdx.x];          // bins in descending order of work
    for (int q = binroot_ptr[bin]; q < binroot_ptr[bin + 1]; ++q) {        // zero the update matrices of this bin's roots
        const int ri = binroot_idx[q]; const long long b6 = 6LL * broot_b[ri]; double* U = ubin + broot_uoff[ri];
        for (long long e = threadIdx.x; e < b6 * b6 + b6; e += 256) U[e] = 0.0;
    }
    // (round 5) what a column needs to know about itself -- column -> pointers -> root descriptor, three dependent look-ups of a microsecond
    // together -- is fetched one column AHEAD; and a thread keeps its row of the column in registers from the update to the triangular solve
    // (it was stored after the update and read back after the pivot: a round trip through the caches on every column's chain)
    struct colinfo { int j, c0, m, t0, T, af, ri, b6; long long mapoff, uoff; };
    auto col_of = [&](int ci2) {
        colinfo k;
        k.j = bincols[ci2];
        k.c0 = colptr[k.j]; k.m = colptr[k.j + 1] - k.c0;
        k.t0 = rlptr[k.j]; k.T = rlptr[k.j + 1] - k.t0;
        k.mapoff = mapptr[k.j]; k.af = anc_first[k.j]; k.ri = broot_of_col[k.j];
        k.b6 = k.ri >= 0 ? 6 * broot_b[k.ri] : 0; k.uoff = k.ri >= 0 ? broot_uoff[k.ri] : 0;
        return k;
    };
    const int ci_end = binptr[bin + 1];
    colinfo cur = {}, nxt = {};
    if (binptr[bin] < ci_end) cur = col_of(binptr[bin]);
    for (int ci = binptr[bin]; ci < ci_end; ++ci, cur = nxt) {
        if (ci + 1 < ci_end) nxt = col_of(ci + 1);
        const int j = cur.j;
        const int c0 = cur.c0, m = cur.m;
        const int t0 = cur.t0, T = cur.T;
        const int* mp = upd_map + cur.mapoff;
        const int idx = threadIdx.x;
        const bool act = idx < 6 * m;
        const int q = act ? idx / 6 : 0, r = idx - q * 6;
        const bool rhs = threadIdx.x >= 250;
        const int rs_ = threadIdx.x - 250;
        double acc[6] = { 0, 0, 0, 0, 0, 0 }, accy = 0;
        double rowv[6] = { 0, 0, 0, 0, 0, 0 };                             // row r of block q of the column, as assembled: requested before the updates
        if (act) {
#pragma unroll
            for (int s2 = 0; s2 < 6; ++s2) rowv[s2] = Lvals[(size_t)(c0 + q) * 36 + r * 6 + s2];
        }
        for (int tc = 0; tc < T; tc += PG_TCH) {
            const int tn = min(PG_TCH, T - tc);
            __syncthreads();
            for (int x0 = threadIdx.x; x0 < tn * 36; x0 += 4 * 256) {      // four dependent (position -> block) loads in flight per thread
                int pos[4]; double val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int xx = x0 + u * 256; pos[u] = xx < tn * 36 ? rlpos[t0 + tc + xx / 36] : 0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int xx = x0 + u * 256; val[u] = xx < tn * 36 ? Lvals[(size_t)pos[u] * 36 + (xx % 36)] : 0.0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int xx = x0 + u * 256; if (xx < tn * 36) s_Ljk[xx] = val[u]; }
            }
            for (int xx = threadIdx.x; xx < tn * 6; xx += 256) s_yk[xx] = x[(size_t)rlcol[t0 + tc + xx / 6] * 6 + (xx % 6)];
            __syncthreads();
            if (rhs)
                for (int t = 0; t < tn; ++t) {
                    const double* yk = s_yk + t * 6; const double* B = s_Ljk + t * 36 + rs_ * 6;
                    accy += B[0] * yk[0] + B[1] * yk[1] + B[2] * yk[2] + B[3] * yk[3] + B[4] * yk[4] + B[5] * yk[5];
                }
            if (act)
                pg_acc_rows(mp + (size_t)tc * m + q, m, tn, Lvals, r, s_Ljk, acc);
        }
        // (round 5) the updated pivot block and right-hand side reach the one-thread section below through LDS: stored to global memory
        // and read back by another thread they were a round trip through the cache hierarchy on every column's chain.  The pivot thread
        // writes both back (the factor and the solved right-hand side), as before.
        if (rhs) s_xj[rs_] = x[(size_t)j * 6 + rs_] - accy;               // (accy = 0 without updates)
        if (act) {
#pragma unroll
            for (int s = 0; s < 6; ++s) rowv[s] -= acc[s];                  // (no updates: acc = 0, x - 0 = x)
            if (idx < 6) { for (int s = 0; s < 6; ++s) s_diag[r * 6 + s] = rowv[s]; }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // (round 4: in-kernel stamps put this one-thread section at 4 200 cycles per column, 28 % of the kernel -- a square root and
            // 21 dependent divisions; one reciprocal square root per pivot and multiplications by it from here on)
            double A[36], xj[6], ri[6];
#pragma unroll
            for (int a = 0; a < 36; ++a) A[a] = s_diag[a];
#pragma unroll
            for (int a = 0; a < 6; ++a) xj[a] = s_xj[a];
            const int bad = chol6_recip(A, ri);
            if (bad) *fail = 1;
            s_ok = !bad;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) { const double v = b <= a ? A[a * 6 + b] : 0.0; s_diag[a * 6 + b] = v; Lvals[(size_t)c0 * 36 + a * 6 + b] = v; }
#pragma unroll
            for (int a = 0; a < 6; ++a) { s_ri[a] = ri[a]; rdiag[(size_t)j * 6 + a] = ri[a]; }
            if (!bad) {
                double v[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    double t = xj[a];
#pragma unroll
                    for (int b = 0; b < 6; ++b) if (b < a) t -= A[a * 6 + b] * v[b];
                    v[a] = t * ri[a];
                }
#pragma unroll
                for (int a = 0; a < 6; ++a) { x[(size_t)j * 6 + a] = v[a]; s_y[a] = v[a]; }
            }
        }
        __syncthreads();
        if (!s_ok) return;
        const int af = cur.af, ta = m - af;                             // block rows beyond the subtree root (a suffix of the column)
        if (act && idx >= 6) {
            double* row = Lvals + (size_t)(c0 + q) * 36 + r * 6;
            double xr[6];
#pragma unroll
            for (int s = 0; s < 6; ++s) { double v = rowv[s];
#pragma unroll
                                          for (int c = 0; c < 6; ++c) if (c < s) v -= xr[c] * s_diag[s * 6 + c];
                                          xr[s] = v * s_ri[s]; }
            for (int s = 0; s < 6; ++s) row[s] = xr[s];
            if (q >= af) for (int s = 0; s < 6; ++s) s_Ljk[(q - af) * 36 + r * 6 + s] = xr[s];       // keep the ancestor rows for the update matrix
        }
        __syncthreads();
        const int ri = cur.ri;
        if (ri >= 0 && ta > 0) {
            const int b6 = cur.b6;
            double* __restrict__ U = ubin + cur.uoff;
            double* __restrict__ g = U + (size_t)b6 * b6;
            // (round 4) the boundary indices of the column's ancestor rows go to LDS once, and the block pairs ib <= ia are ONE flat loop:
            // per ancestor row the pass was a dependent index load, a load and a store of U in global memory, and the rows followed each
            // other (in-kernel stamps: 3 200 cycles per column, a quarter of the kernel).  Every entry of U still receives one term per column.
            if ((int)threadIdx.x < ta) s_arel[threadIdx.x] = anc_rel[c0 + af + threadIdx.x];
            __syncthreads();
            if ((int)threadIdx.x < 6 * ta) {                            // right-hand side: g[ia] -= L_a y_j
                const int pa = threadIdx.x / 6, a = threadIdx.x - pa * 6;
                const double* La = s_Ljk + pa * 36 + a * 6;
                g[s_arel[pa] * 6 + a] -= La[0] * s_y[0] + La[1] * s_y[1] + La[2] * s_y[2] + La[3] * s_y[3] + La[4] * s_y[4] + La[5] * s_y[5];
            }
            const int npair = ta * (ta + 1) / 2;                        // U[ia][ib] -= L_a L_b^T for the block pairs ib <= ia
            for (int e = threadIdx.x; e < 36 * npair; e += 256) {
                const int pr = e / 36, ab = e - 36 * pr;
                int pa = (int)((sqrtf(8.0f * (float)pr + 1.0f) - 1.0f) * 0.5f);      // pr = pa (pa + 1) / 2 + pb, pb <= pa
                while (pa * (pa + 1) / 2 > pr) --pa;
                while ((pa + 1) * (pa + 2) / 2 <= pr) ++pa;
                const int pb = pr - pa * (pa + 1) / 2, a = ab / 6, b = ab - 6 * a;
                const double* La = s_Ljk + pa * 36 + a * 6; const double* Lb = s_Ljk + pb * 36 + b * 6;
                U[(size_t)(s_arel[pa] * 6 + a) * b6 + s_arel[pb] * 6 + b] -= La[0] * Lb[0] + La[1] * Lb[1] + La[2] * Lb[2] + La[3] * Lb[3] + La[4] * Lb[4] + La[5] * Lb[5];
            }
        }
        __syncthreads();
        __threadfence_block();
    }
}
// backward substitution through a bin, columns in descending order, one wave per bin
__global__ __launch_bounds__(64) void pg_bwd_subtree_kernel(const int* __restrict__ bin_perm, const int* __restrict__ binptr, const int* __restrict__ bincols,
                                                            const int* __restrict__ colptr, const int* __restrict__ rowidx,
                                                            const double* __restrict__ Lvals, double* __restrict__ x, const double* __restrict__ rdiag)
{
    const int lane = threadIdx.x;
    const int bin = bin_perm[blockIdx.x];
    // (round 5) column -> its pointers, and the pivot block with its reciprocals -- nothing of which depends on the solution so far -- are
    // fetched one column ahead of the chain (x of the ancestors -> sums -> the 6 x 6 back-substitution -> x of the column)
    const int ci_lo = binptr[bin];
    int jn = 0, cn0 = 0, cn1 = 0;
    double ldn[36], rin[6];
    auto ahead = [&](int ci2) {
        jn = bincols[ci2]; cn0 = colptr[jn]; cn1 = colptr[jn + 1];
        if (lane == 0) {
            const double* Ld = Lvals + (size_t)cn0 * 36;
#pragma unroll
            for (int a = 0; a < 36; ++a) ldn[a] = Ld[a];
#pragma unroll
            for (int a = 0; a < 6; ++a) rin[a] = rdiag[(size_t)jn * 6 + a];
        }
    };
    if (binptr[bin + 1] - 1 >= ci_lo) ahead(binptr[bin + 1] - 1);
    for (int ci = binptr[bin + 1] - 1; ci >= ci_lo; --ci) {
        const int j = jn, c0 = cn0, c1 = cn1;
        double ld[36], ri[6];
#pragma unroll
        for (int a = 0; a < 36; ++a) ld[a] = ldn[a];
#pragma unroll
        for (int a = 0; a < 6; ++a) ri[a] = rin[a];
        if (ci - 1 >= ci_lo) ahead(ci - 1);
        double acc[6] = { 0, 0, 0, 0, 0, 0 };
        for (int p = c0 + 1 + lane; p < c1; p += 64) {
            const double* B = Lvals + (size_t)p * 36; const double* xi = x + (size_t)rowidx[p] * 6;
            for (int a = 0; a < 6; ++a) { double s = 0; for (int b = 0; b < 6; ++b) s += B[b * 6 + a] * xi[b]; acc[a] += s; }
        }
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) acc[a] += __shfl_xor(acc[a], o, 64);
        if (lane == 0) {
            double v[6], xj[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) xj[a] = x[(size_t)j * 6 + a];
#pragma unroll
            for (int a = 5; a >= 0; --a) { double s = xj[a] - acc[a];
#pragma unroll
                                           for (int b = 0; b < 6; ++b) if (b > a) s -= ld[b * 6 + a] * v[b];
                                           v[a] = s * ri[a]; }
#pragma unroll
            for (int a = 0; a < 6; ++a) x[(size_t)j * 6 + a] = v[a];
        }
        __threadfence_block();
        __builtin_amdgcn_s_barrier();
    }
}

// the instances the driver launches (the templates are declared in dsss_pg_kernels.h)
template __global__ void pg_scan_block_kernel<int, 0>(int, const int* __restrict__, const int* __restrict__, int* __restrict__, int* __restrict__);
template __global__ void pg_scan_block_kernel<long long, 1>(int, const int* __restrict__, const int* __restrict__, long long* __restrict__, long long* __restrict__);
template __global__ void pg_scan_tops_kernel<int>(int, int* __restrict__, int* __restrict__);
template __global__ void pg_scan_tops_kernel<long long>(int, long long* __restrict__, long long* __restrict__);
template __global__ void pg_scan_add_kernel<int>(int, int* __restrict__, const int* __restrict__, const int* __restrict__);
template __global__ void pg_scan_add_kernel<long long>(int, long long* __restrict__, const long long* __restrict__, const long long* __restrict__);
