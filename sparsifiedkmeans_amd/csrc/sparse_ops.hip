// Drop-in device versions of the reference's stand-alone sparse mex kernels (gfx950).
// Each thread owns one (column[, centroid]) accumulator and walks the column's stored
// entries in storage order, so every output is bit-identical to the reference loops:
//   k_dist_full    private/SparseMatrixMinusCluster.c:133-182   (K x n distances)
//   k_dist_beta    private/SparseMatrixMinusCluster.c:118-129   (beta form, K = 1)
//   k_innerprod    private/SparseMatrixInnerProduct.c:87-100
//   k_colnormsq    private/SparseMatrixColumnNormSq.c:71-77
// These are HBM-bound helpers (<= 1.25 flop/B); the Lloyd hot path uses assign.hip instead.
#include "common.h"

// Ct[r*K + k] = C[k*p + r]  (row-major copy so that the K accumulators of one point read
// contiguous centroid values)
__global__ void k_transpose_centers(const double* __restrict__ C, int p, int K, double* __restrict__ Ct)
{
    const size_t total = (size_t)p * K;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(t % K);
        const size_t r = t / K;
        Ct[t] = C[(size_t)k * p + r];
    }
}

// grid.y strides over k in chunks of blockDim.x lanes; one wave-row of threads shares a point.
template <typename IR>
__global__ __launch_bounds__(256) void k_dist_full(const long long* __restrict__ jc, const IR* __restrict__ ir,
                                                   const double* __restrict__ x, const double* __restrict__ Ct,
                                                   int K, long long n, double* __restrict__ dist)
{
    // thread -> (point i, centroid k): k fastest so that Ct reads and dist writes coalesce
    const long long total = n * (long long)K;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long i = t / K;
        const int k = (int)(t - i * K);
        double acc = 0.0;
        const long long j1 = jc[i + 1];
        for (long long j = jc[i]; j < j1; j++) {
            const double d = x[j] - Ct[(size_t)ir[j] * K + k];
            acc = acc + d * d;
        }
        dist[t] = sqrt(acc);
    }
}

template <typename IR>
__global__ __launch_bounds__(256) void k_dist_beta(const long long* __restrict__ jc, const IR* __restrict__ ir,
                                                   const double* __restrict__ x, const double* __restrict__ c,
                                                   double beta, long long n, double* __restrict__ dist)
{
    const double b = beta * -2.0; // SparseMatrixMinusCluster.c:121
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        double acc = 0.0;
        const long long j1 = jc[i + 1];
        for (long long j = jc[i]; j < j1; j++) {
            const double xv = x[j], cv = c[ir[j]];
            const double t1 = xv * xv;
            const double t2 = (b * xv) * cv;
            const double t3 = cv * cv;
            acc = acc + ((t1 + t2) + t3); // :125, C evaluation order
        }
        dist[i] = sqrt(acc);
    }
}

template <typename IR>
__global__ __launch_bounds__(256) void k_innerprod(const long long* __restrict__ jc, const IR* __restrict__ ir,
                                                   const double* __restrict__ x, const double* __restrict__ c,
                                                   long long n, double* __restrict__ ip, double* __restrict__ nx2)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        double a = 0.0, nrm = 0.0;
        const long long j1 = jc[i + 1];
        for (long long j = jc[i]; j < j1; j++) {
            const double xv = x[j];
            a = a + xv * c[ir[j]];
            nrm = nrm + xv * xv;
        }
        ip[i] = a;
        nx2[i] = nrm;
    }
}

__global__ __launch_bounds__(256) void k_colnormsq(const long long* __restrict__ jc, const double* __restrict__ x,
                                                   long long n, double* __restrict__ nx2)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        double nrm = 0.0;
        const long long j1 = jc[i + 1];
        for (long long j = jc[i]; j < j1; j++) nrm = nrm + x[j] * x[j];
        nx2[i] = nrm;
    }
}

template __global__ void k_dist_full<unsigned short>(const long long*, const unsigned short*, const double*,
    const double*, int, long long, double*);
template __global__ void k_dist_full<unsigned int>(const long long*, const unsigned int*, const double*,
    const double*, int, long long, double*);
template __global__ void k_dist_beta<unsigned short>(const long long*, const unsigned short*, const double*,
    const double*, double, long long, double*);
template __global__ void k_dist_beta<unsigned int>(const long long*, const unsigned int*, const double*,
    const double*, double, long long, double*);
template __global__ void k_innerprod<unsigned short>(const long long*, const unsigned short*, const double*,
    const double*, long long, double*, double*);
template __global__ void k_innerprod<unsigned int>(const long long*, const unsigned int*, const double*,
    const double*, long long, double*, double*);
