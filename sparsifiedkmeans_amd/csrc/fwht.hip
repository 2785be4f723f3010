// Fast Walsh-Hadamard transform of the columns of a dense m x n f64 matrix (gfx950).
//
// Replaces hadamard_apply_vector / hadamard_apply_matrix(_threads)
// (private/hadamard.c:57-92, private/hadamard_pthreads.c:69-107,121-204): unnormalised,
// Sylvester ("hadamard") ordering, stages applied in the reference's order bit = 1, 2, 4, ..., m/2.
// Every stage is the same set of disjoint (y[j], y[j|bit]) <- (y[j]+y[j|bit], y[j]-y[j|bit])
// updates as the reference, so the result is bit-identical (add/sub only, no reassociation).
//
// Optional fusions for the preconditioner  mix(X) = hadamard(D*[X*(1+2eps); 0]) / sqrt(p2)
// (kmeans_sparsified.m:241-248,286-295): rows >= p_in read as zero, sign vector d, premultiplier,
// post-divide.  HBM-bound: 16 B moved per element, m*log2(m) add/sub per column.
//
// Kernel shape: T = m/16 threads own one column; stages are taken 4 at a time in registers
// (16 elements per thread), with an LDS exchange between rounds.  LDS index padding
// P(e) = e + 2*(e>>4) keeps the 16-B-aligned per-thread runs conflict-free.
#include "common.h"

__device__ __forceinline__ int padidx(int e) { return e + ((e >> 4) << 1); }

// One round of NB (<= 4) butterfly stages, bits b .. b+NB-1, on the 16 elements this thread owns:
//   NB == 4: e(q) = (H << (b+4)) | (q << b) | L      with tau = (H << b) | L
//   NB <  4 (last round, b + NB == logm): the 4-NB spare bits of q come from the top of L
// Stages are applied in ascending bit order (the reference's order, hadamard.c:66-77).
template <int NB>
__device__ __forceinline__ void fwht_round(double* __restrict__ col, int tau, int b, double postdiv)
{
    auto idx = [&](int q) {
        if constexpr (NB == 4) {
            const int L = tau & ((1 << b) - 1), H = tau >> b;
            return (H << (b + 4)) | (q << b) | L;
        } else {
            return ((q & ((1 << NB) - 1)) << b) | ((q >> NB) << (b - (4 - NB))) | tau;
        }
    };
    double a[16];
#pragma unroll
    for (int q = 0; q < 16; q++) a[q] = col[padidx(idx(q))];
#pragma unroll
    for (int s = 0; s < NB; s++) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if ((q & (1 << s)) == 0) {
                const double u = a[q], w = a[q | (1 << s)];
                a[q] = u + w;
                a[q | (1 << s)] = u - w;
            }
        }
    }
    if (postdiv > 0.0) {
#pragma unroll
        for (int q = 0; q < 16; q++) a[q] = a[q] / postdiv;
    }
#pragma unroll
    for (int q = 0; q < 16; q++) col[padidx(idx(q))] = a[q];
}

__global__ __launch_bounds__(1024) void k_fwht_lds(const double* __restrict__ x, double* __restrict__ y, int m,
                                                   int logm, long long n, int p_in,
                                                   const double* __restrict__ dsign, double premul,
                                                   double postdiv, int cols_per_block,
                                                   const void* __restrict__ gather_ir, int gather_bits, int gather_s,
                                                   double gather_level, long long gather_stride = 0)
{
    // gather epilogue (sample.hip): when gather_ir != null the transformed column stays in LDS and only
    // its gather_s sampled rows are written, y[c*gather_s + t] = (Y[row_t] / postdiv) / gather_level.
    // gather_stride > 0: column c's row ids start gather_stride BYTES after column c - 1's, and so do its values -- the
    // record layout (k_build_records: a point's values and row ids side by side), written directly by the sparsifier.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int T = m >> 4;                    // threads per column
    const int csub = threadIdx.x / T;        // which column of this block
    const int tau0_ = threadIdx.x % T;
    const int colstride = m + (m >> 3);      // padded doubles per column
    double* col = lds + (size_t)csub * colstride;

    const int tau_c = tau0_;
    // T <= 64: a column lives inside one wave, whose LDS operations complete in program order -- a compiler
    // fence is enough and the four waves of a workgroup never wait for each other
    const bool wave_local = T <= 64;
    auto sync = [&]() {
        if (wave_local) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    for (long long cbase = (long long)blockIdx.x * cols_per_block; cbase < n;
         cbase += (long long)gridDim.x * cols_per_block) {
        // hipcc hoists the ~100 per-thread index expressions of an iteration out of this loop and then spills
        // them to scratch (428 B per lane); making the thread index opaque per iteration keeps them in flight
        int tau = tau_c;
        asm volatile("" : "+v"(tau));
        // ---- coalesced load of cols_per_block columns into LDS (with the fused input ops) ----
        // thread (csub, tau) fetches elements tau + T*i, i = 0..15, of its own column: 16 independent loads
        // in flight per lane (a rolled loop would expose 16 HBM latencies one after the other)
        const long long ncols = (n - cbase < cols_per_block) ? (n - cbase) : cols_per_block;
        {
            double v[16];
            double sg[16];
            const bool have = csub < ncols;
            // unconditional loads on clamped indices (a predicated load would be waited for before the next
            // one is issued); out-of-range rows / columns are zeroed afterwards
            const double* xc = x + (size_t)(cbase + (have ? csub : 0)) * p_in;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int r = tau + T * i;
                v[i] = xc[r < p_in ? r : p_in - 1];
            }
            if (dsign) {
#pragma unroll
                for (int i = 0; i < 16; i++) sg[i] = dsign[tau + T * i];
            }
            if (have) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int r = tau + T * i;
                    double w = (r < p_in) ? v[i] : 0.0;
                    if (premul != 1.0) w = w * premul;
                    if (dsign) w = sg[i] * w;
                    col[padidx(r)] = w;
                }
            }
        }
        sync();

        // barriers must be reached by every lane of every wave the same number of times: the
        // rounds are predicated per column instead of branching around them
        const bool active = cbase + csub < n;
        for (int b = 0; b < logm; b += 4) {
            if (active) {
                const int nb = (logm - b < 4) ? (logm - b) : 4;
                const bool last = b + nb >= logm;
                const double pd = (last && !gather_ir) ? postdiv : 0.0;
                switch (nb) {
                case 4: fwht_round<4>(col, tau, b, pd); break;
                case 3: fwht_round<3>(col, tau, b, pd); break;
                case 2: fwht_round<2>(col, tau, b, pd); break;
                default: fwht_round<1>(col, tau, b, pd); break;
                }
            }
            // a round touches disjoint element sets per thread, but the next round reads other threads'
            // elements: block-wide barrier
            sync();
        }

        // ---- coalesced store (dense), or gather of the sampled rows ----
        if (!gather_ir) {
            if (csub < ncols) {
                double* yc = y + (size_t)(cbase + csub) * m;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int r = tau + T * i;
                    yc[r] = col[padidx(r)];
                }
            }
        } else {
            // each column is gathered by the threads that own it (stays wave-local for T <= 64)
            if (csub < ncols) {
                const size_t cc = (size_t)(cbase + csub);
                const char* irc = reinterpret_cast<const char*>(gather_ir) + (gather_stride > 0 ? cc * (size_t)gather_stride : cc * (size_t)gather_s * (gather_bits / 8));
                double* yc = gather_stride > 0 ? reinterpret_cast<double*>(reinterpret_cast<char*>(y) + cc * (size_t)gather_stride) : y + cc * gather_s;
                for (int t = tau; t < gather_s; t += T) {
                    const int r = (gather_bits == 16) ? (int)reinterpret_cast<const unsigned short*>(irc)[t]
                                                      : (int)reinterpret_cast<const unsigned int*>(irc)[t];
                    double v = col[padidx(r)];
                    if (postdiv > 0.0) v = v / postdiv;
                    yc[t] = v / gather_level;
                }
            }
        }
        sync();
    }
}

// m in {2,4,8}: one thread per column, registers only.
__global__ void k_fwht_small(const double* __restrict__ x, double* __restrict__ y, int m, long long n, int p_in,
                             const double* __restrict__ dsign, double premul, double postdiv)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    double a[8];
    for (int r = 0; r < m; r++) {
        double v = 0.0;
        if (r < p_in) { v = x[(size_t)c * p_in + r]; if (premul != 1.0) v = v * premul; }
        if (dsign) v = dsign[r] * v;
        a[r] = v;
    }
    for (int bit = 1; bit < m; bit <<= 1)
        for (int j = 0; j < m; j++)
            if ((j & bit) == 0) { const double u = a[j], w = a[j | bit]; a[j] = u + w; a[j | bit] = u - w; }
    for (int r = 0; r < m; r++) y[(size_t)c * m + r] = (postdiv > 0.0) ? a[r] / postdiv : a[r];
}

// m > 16384: stage-by-stage in global memory (one launch per stage; rare).
__global__ void k_fwht_load(const double* __restrict__ x, double* __restrict__ y, long long m, long long n,
                            long long p_in, const double* __restrict__ dsign, double premul)
{
    const long long total = m * n;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long c = t / m, r = t - c * m;
        double v = 0.0;
        if (r < p_in) { v = x[c * p_in + r]; if (premul != 1.0) v = v * premul; }
        if (dsign) v = dsign[r] * v;
        y[t] = v;
    }
}
__global__ void k_fwht_stage(double* __restrict__ y, long long m, long long n, long long bit, double postdiv)
{
    const long long half = (m >> 1) * n;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < half;
         t += (long long)gridDim.x * blockDim.x) {
        const long long c = t / (m >> 1), h = t - c * (m >> 1);
        const long long j = ((h & ~(bit - 1)) << 1) | (h & (bit - 1));
        double* col = y + c * m;
        const double u = col[j], w = col[j | bit];
        double s = u + w, d = u - w;
        if (postdiv > 0.0) { s = s / postdiv; d = d / postdiv; }
        col[j] = s;
        col[j | bit] = d;
    }
}
