// diasss_amd/csrc/dsss_pg_dev.h -- small dense device helpers (6 x 6 row-major) of the pose-graph kernels
#pragma once
#include <hip/hip_runtime.h>

// ------------------------------------------------------------------ small dense helpers (6x6 row-major)
// 6 x 6 Cholesky with the reciprocal of a correctly rounded square root (one sqrt and one division per pivot), ri[j] = 1 / L[j][j]: for the bins, whose 17 k columns
// carry the whole dynamic range of the chain condensation -- with rsqrt here two elimination orders of the C3 graph end 1.6e-6 apart, with
// this 3e-7 (test_config_C4_full_size_8_partitions_and_2_ranks)
__device__ inline int chol6_recip(double* A, double* ri)
{
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0) || !isfinite(d)) { bad = 1; d = 1.0; }
        const double sq = sqrt(d), r = 1.0 / sq;
        A[j * 6 + j] = sq; ri[j] = r;
#pragma unroll
        for (int i = 0; i < 6; ++i) if (i > j) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k < j) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s * r;
        }
    }
    return bad;
}
// the same with 1 / L[j][j] left ON the diagonal (what the solves multiply by): no separate reciprocal array, twelve registers less
__device__ inline int chol6_rdiag(double* A)
{
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < j) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0) || !isfinite(d)) { bad = 1; d = 1.0; }
        const double r = rsqrt(d);
        A[j * 6 + j] = r;
#pragma unroll
        for (int i = 0; i < 6; ++i) if (i > j) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k < j) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s * r;
        }
    }
    return bad;
}
