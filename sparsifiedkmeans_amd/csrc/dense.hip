// Dense (unsampled) data: the two kernels behind the reference's two-pass outputs
// (kmeans_sparsified.m:540-562, private/recalculateAssignmentLargeFile.m:85-113).
//
//   k_dense_assign      [assignments, distances] = findClusterAssignments(full(X), centers), dense branch,
//                       expanded quadratic (private/findClusterAssignments.m:157-165,168-171):
//                           distances(k,:) = nrm2 - 2*(X'*c_k)' + norm(c_k)^2 ; sqrt ; min over k (first index).
//                       The K x n Gram block X'*C is GEMM-shaped f64 work and runs on the matrix cores
//                       (v_mfma_f64_16x16x4_f64); the K x n matrix is never written: the epilogue forms the
//                       distances in registers and keeps the running minimum.
//   k_dense_accumulate  per-cluster sums of the dense columns (mean(full(X(:,ind)),2) numerators) over the
//                       counting-sort segments that the sparse path already builds.
//
// The reference's X'*c is a BLAS call and its pdist2 alternative is closed source, so the summation order is
// not defined by the reference: parity here is to a tolerance (tests), not bitwise.
// sqrt of a slightly negative rounded value (a point equal to a centre) is clamped to 0; MATLAB would return
// a complex number there.
#include "common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#define DA_PTS 64    // points per workgroup (16 per wave)
#define DA_ROWS 64   // rows staged per step
#define DA_KP 128    // centroids per pass (8 MFMA tiles of 16)
#define DA_LD 66     // LDS leading dimension in doubles: (4*pt + 2*kq) mod 64 is conflict-free for b64 reads

typedef double d4v __attribute__((ext_vector_type(4)));

// out[i] = sum_r A[i*ld + r]^2   (one wave per row of A)
__global__ __launch_bounds__(256) void k_rows_normsq(const double* __restrict__ A, long long nrows, int p,
                                                     double* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long i = w; i < nrows; i += nw) {
        const double* a = A + (size_t)i * p;
        double s = 0.0;
        for (int r = lane; r < p; r += 64) s += a[r] * a[r];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) out[i] = s;
    }
}

// X: n x p (point i at X + i*p), C: K x p (centre k at C + k*p)
__global__ __launch_bounds__(256) void k_dense_assign(const double* __restrict__ X, long long n, int p,
                                                      const double* __restrict__ C, int K,
                                                      const double* __restrict__ xn2, const double* __restrict__ cn2,
                                                      int* __restrict__ assign, double* __restrict__ dist)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* Xs = reinterpret_cast<double*>(smem);          // [DA_PTS][DA_LD]
    double* Cs = Xs + DA_PTS * DA_LD;                      // [DA_KP][DA_LD]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const long long i0 = (long long)blockIdx.x * DA_PTS;
    const long long my_pt = i0 + wv * 16 + col;
    const double my_n2 = my_pt < n ? xn2[my_pt] : 0.0;
    double best = __builtin_inf();
    int bk = 0x7fffffff;
    for (int kb = 0; kb < K; kb += DA_KP) {
        d4v acc[DA_KP / 16];
#pragma unroll
        for (int t = 0; t < DA_KP / 16; t++) acc[t] = d4v{0.0, 0.0, 0.0, 0.0};
        const int ntile = (min(K - kb, DA_KP) + 15) >> 4;
        for (int r0 = 0; r0 < p; r0 += DA_ROWS) {
            __syncthreads();
            for (int idx = tid; idx < DA_PTS * DA_ROWS; idx += 256) {
                const int pt = idx >> 6, r = idx & 63;
                const long long i = i0 + pt;
                Xs[pt * DA_LD + r] = (i < n && r0 + r < p) ? X[(size_t)i * p + r0 + r] : 0.0;
            }
            for (int idx = tid; idx < ntile * 16 * DA_ROWS; idx += 256) {
                const int kk = idx >> 6, r = idx & 63;
                Cs[kk * DA_LD + r] = (kb + kk < K && r0 + r < p) ? C[(size_t)(kb + kk) * p + r0 + r] : 0.0;
            }
            __syncthreads();
            const double* xrow = Xs + (wv * 16 + col) * DA_LD + kq;
            const double* crow = Cs + col * DA_LD + kq;
#pragma unroll 4
            for (int rr = 0; rr < DA_ROWS; rr += 4) {
                const double b = xrow[rr];
#pragma unroll
                for (int t = 0; t < DA_KP / 16; t++)
                    if (t < ntile) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(crow[t * 16 * DA_LD + rr], b, acc[t], 0, 0, 0);
            }
        }
        // f64 16x16x4 result layout: register j of lane (kq, col) is D[4*j + kq][col]: centroid kb + 16 t + 4 j + kq
#pragma unroll
        for (int t = 0; t < DA_KP / 16; t++) {
            if (t < ntile) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = kb + 16 * t + 4 * j + kq;
                    if (k < K) {
                        const double d2 = (my_n2 - 2.0 * acc[t][j]) + cn2[k];
                        const double d = sqrt(d2 > 0.0 ? d2 : 0.0);
                        if (d < best || (d == best && k < bk)) { best = d; bk = k; }
                    }
                }
            }
        }
    }
    // the four lanes (kq = 0..3) that share a point hold disjoint centroid subsets
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const double ob = __shfl_xor(best, off);
        const int ok = __shfl_xor(bk, off);
        if (ob < best || (ob == best && ok < bk)) { best = ob; bk = ok; }
    }
    if (kq == 0 && my_pt < n) {
        if ((unsigned)bk >= (unsigned)K) bk = 0; // non-finite distances: index 1 as MATLAB's min(), never out of range
        assign[my_pt] = bk;
        dist[my_pt] = best;
    }
}

// One workgroup per counting-sort item (cluster k, segment of its points): sums[k*p + r] += sum over the
// segment's points of X[i*p + r] (point order), one hardware f64 atomic per row and segment.
__global__ __launch_bounds__(256) void k_dense_accumulate(const double* __restrict__ X, int p,
                                                          const int* __restrict__ perm,
                                                          const long long* __restrict__ offs,
                                                          const int4* __restrict__ items,
                                                          const int* __restrict__ nitems, double* __restrict__ sums)
{
    const int ni = *nitems;
    for (int it = blockIdx.x; it < ni; it += gridDim.x) {
        const int4 item = items[it];
        const int* pp = perm + offs[item.x] + item.y;
        for (int r = threadIdx.x; r < p; r += blockDim.x) {
            double s = 0.0;
            for (int j = 0; j < item.z; j++) s += X[(size_t)pp[j] * p + r];
            unsafeAtomicAdd(&sums[(size_t)item.x * p + r], s);
        }
    }
}

__global__ void k_nk_add_f64(const unsigned long long* __restrict__ nk, int K, double* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) out[k] += (double)nk[k];
}
