// Device sparsifier: the step that produces the hot path's input (gfx950).
//
// Replaces  X = randsample_fixedNumberEntries(mix(X), small_p)  (kmeans_sparsified.m:316-334,
// private/randsample_fixedNumberEntries.m:30-64, private/randsample_block.m:37-92): per column exactly
// s distinct rows, uniformly at random and independent between columns, stored ascending, values
// mixed(row)/(s/p2).  The reference draws with MATLAB's generator (randperm prefix when 4s > p2, rejection
// of duplicates otherwise: two implementations of the same distribution); any exact sampler without
// replacement is equivalent, bit parity of the random draws is not defined.  Here: Knuth's selection
// sampling (Algorithm S: row r is taken with probability (s - taken)/(p2 - r)) driven by Philox4x32-10
// keyed with (seed, GLOBAL column index), so the sample of a column does not depend on chunking, on the
// number of GPUs or on launch geometry.
//
//   k_sample_rows   one lane per column -> ascending row ids (HBM-write bound; ~15 VALU per row visited)
//   k_fwht_lds      (fwht.hip) with the gather epilogue: the mixed column never leaves LDS; only the s
//                   sampled values x = (y/sqrt(p2))/(s/p2) are written (two true divisions, as the reference)
#include "common.h"

struct philox4 { unsigned x, y, z, w; };

__host__ __device__ inline philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                                 unsigned k1)
{
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return philox4{c0, c1, c2, c3};
}

// ir_out[(c)*s + t] = t-th smallest sampled row of global column col0 + c.
template <typename IR>
__global__ __launch_bounds__(256) void k_sample_rows(unsigned long long seed, long long col0, long long n, int p2,
                                                     int s, IR* __restrict__ ir_out, long long stride_bytes = 0)
{
    // stride_bytes > 0: column c's ids start that many BYTES after column c - 1's (the record layout)
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < n;
         c += (long long)gridDim.x * blockDim.x) {
        const unsigned long long gc = (unsigned long long)(col0 + c);
        IR* out = stride_bytes > 0 ? reinterpret_cast<IR*>(reinterpret_cast<char*>(ir_out) + (size_t)c * (size_t)stride_bytes)
                                   : ir_out + (size_t)c * s;
        int taken = 0;
        for (int r0 = 0; r0 < p2 && taken < s; r0 += 4) {
            const philox4 rnd = philox4x32_10((unsigned)gc, (unsigned)(gc >> 32), (unsigned)(r0 >> 2), 0u,
                                              (unsigned)seed, (unsigned)(seed >> 32));
            const unsigned u[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = r0 + q;
                if (r < p2 && taken < s) {
                    // take row r with probability (s - taken) / (p2 - r):  floor(u * (p2 - r) / 2^32) < s - taken
                    const unsigned t = (unsigned)(((unsigned long long)u[q] * (unsigned)(p2 - r)) >> 32);
                    if (t < (unsigned)(s - taken)) out[taken++] = (IR)r;
                }
            }
        }
    }
}

// Widening copy in front of the sparsifier: a streamed chunk arrives in the source's own element type (8-bit pixels,
// float32 features -- what a 1e9-point dataset is stored as) and is widened to the double
// the reference's pipeline works in (sampleAndMixFromLargeFile.m:104-113 reads doubles).  Every uint8 / int16 /
// float32 value is exactly representable: the copy is exact.  16 B stored per lane and iteration.
//   kind: 1 float32, 2 uint8, 3 int16, 4 int32
__global__ __launch_bounds__(256) void k_widen_f64(const void* __restrict__ src, int kind, long long count,
                                                   double* __restrict__ dst)
{
    const long long stride = (long long)gridDim.x * blockDim.x * 2;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < count; i += stride) {
        double a, b = 0.0;
        const bool two = i + 1 < count;
        switch (kind) {
        case 1: a = (double)static_cast<const float*>(src)[i]; if (two) b = (double)static_cast<const float*>(src)[i + 1]; break;
        case 2: a = (double)static_cast<const unsigned char*>(src)[i]; if (two) b = (double)static_cast<const unsigned char*>(src)[i + 1]; break;
        case 3: a = (double)static_cast<const short*>(src)[i]; if (two) b = (double)static_cast<const short*>(src)[i + 1]; break;
        default: a = (double)static_cast<const int*>(src)[i]; if (two) b = (double)static_cast<const int*>(src)[i + 1]; break;
        }
        dst[i] = a;
        if (two) dst[i + 1] = b;
    }
}
