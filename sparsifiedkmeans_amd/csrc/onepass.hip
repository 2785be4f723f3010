// One pass over the points for few centroids (K x p small enough for the per-cluster sums to live in LDS beside the
// centroid tile): screen, certificate and accumulation fused, reading each point's RECORD (f64 values | row ids, the
// layout of the exact pass) once -- 510 B per point at s = 51 instead of the screen's 306-B copy plus the exact pass's
// 510 B.  What the reference does in these places: findClusterAssignments.m:76-82,168-171 (distances to every centroid,
// min), kmeans_sparsified.m:430-431,447-448 (per-cluster sums and counts).
//
// Taken by run_screen (api.hip) for a LAZY call (spkm_shard_set_lazy_stats: no distances, no objective asked for) that
// would otherwise run the full accumulation pass -- a run's first call, or one in which more than a third of the points
// moved -- when K <= 16 and (p + 1) x KP x 4 + K x p x 10 bytes fit the CU's LDS (KP = K rounded up to 4; p = 1024: K <= 10,
// which is config 5 and the MNIST config) AND SPKM_ONEPASS=1: the form is OFF by default, because it measured slower than the two
// kernels it replaces (config 5: 28.4 ms against 9.2 + 14.8; DESIGN.md 4.2g says where the time goes).
//
// Arithmetic: the screen's (screen.hip header) -- x~ = fl32(x), c~ = fl32(c / gamma), a~_k = sum fl32((x~ - c~)^2) in f32,
// certified iff (r1 + eps1)(1 + 2^-45) < (r2 - eps2)(1 - 2^-45) with the same eps -- so a certified point's cluster is
// the reference's; the others go to the exact list (k_assign_list) and are added by k_onepass_listed afterwards.  Sums are
// the members' f64 values added by LDS atomics per workgroup and global atomics per chunk: "the members' sum to rounding",
// as in every other accumulation path; counts are integers.
//
// Layout: 4 lanes per point, 16 points per wave and step; lane l4 of a quad holds entries l4, l4 + 4, ... of its point
// (values f64 for the sums, converted to f32 for the screen) and evaluates them against ALL KP centroids (a tile row is
// KP floats: NQ 16-byte reads); the quad adds its four partial sums at the end (2 DPP steps per centroid).
// LDS: tile (p + 1) x KP f32 | sums K x p f64 | counts K x p u16 (pairs in u32) | cluster sizes K u32.  The 16-bit counts
// are what makes K = 10 at p = 1024 fit (151 KB): a workgroup works through CHUNKS of at most 65 520 points and empties
// its slab into the global sums behind each (no count can reach 65 536).
#include "common.h"

typedef float op_f2 __attribute__((ext_vector_type(2)));
typedef float op_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float op_quad_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)); // [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)); // [2,3,0,1]
    return v;
}


// One or two rounds of the screen's inner block: a lane's entry (value xx.x, tile row at LDS byte address a) against the
// 2 NP centroids of its row -- NP 8-byte reads, then per pair t = c~ + x~ (v_pk_add_f32, x broadcast by op_sel_hi) and
// acc += t * t (v_pk_fma_f32).  Hand-placed: left to the scheduler, the reads of all 13 rounds go to the front (150
// registers in flight, spilled).  The two-round form issues the second round's reads before the first round's arithmetic.
#define OP_RD(t, a, o) "ds_read_b64 %[" #t "], %[" #a "] offset:" #o "\n\t"
#define OP_MATH(t, x, c) "v_pk_add_f32 %[" #t "], %[" #t "], %[" #x "] op_sel_hi:[1,0]\n\tv_pk_fma_f32 %[" #c "], %[" #t "], %[" #t "], %[" #c "]\n\t"
template <int NP> struct op_rounds;
template <> struct op_rounds<2> {
    static __device__ __forceinline__ void one(unsigned a, op_f2 x, op_f2* acc)
    {
        op_f2 t0, t1;
        asm volatile(OP_RD(t0, a, 0) OP_RD(t1, a, 8) "s_waitcnt lgkmcnt(0)\n\t" OP_MATH(t0, x, c0) OP_MATH(t1, x, c1)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [c0] "+v"(acc[0]), [c1] "+v"(acc[1]) : [a] "v"(a), [x] "v"(x));
    }
    static __device__ __forceinline__ void two(unsigned a, op_f2 x, unsigned b, op_f2 y, op_f2* acc)
    {
        op_f2 t0, t1, u0, u1;
        asm volatile(OP_RD(t0, a, 0) OP_RD(t1, a, 8) OP_RD(u0, b, 0) OP_RD(u1, b, 8) "s_waitcnt lgkmcnt(2)\n\t"
                     OP_MATH(t0, x, c0) OP_MATH(t1, x, c1) "s_waitcnt lgkmcnt(0)\n\t" OP_MATH(u0, y, c0) OP_MATH(u1, y, c1)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [u0] "=&v"(u0), [u1] "=&v"(u1), [c0] "+v"(acc[0]), [c1] "+v"(acc[1])
                     : [a] "v"(a), [x] "v"(x), [b] "v"(b), [y] "v"(y));
    }
};
template <> struct op_rounds<4> {
    static __device__ __forceinline__ void one(unsigned a, op_f2 x, op_f2* acc)
    {
        op_f2 t0, t1, t2, t3;
        asm volatile(OP_RD(t0, a, 0) OP_RD(t1, a, 8) OP_RD(t2, a, 16) OP_RD(t3, a, 24) "s_waitcnt lgkmcnt(0)\n\t"
                     OP_MATH(t0, x, c0) OP_MATH(t1, x, c1) OP_MATH(t2, x, c2) OP_MATH(t3, x, c3)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [c0] "+v"(acc[0]), [c1] "+v"(acc[1]),
                       [c2] "+v"(acc[2]), [c3] "+v"(acc[3])
                     : [a] "v"(a), [x] "v"(x));
    }
    static __device__ __forceinline__ void two(unsigned a, op_f2 x, unsigned b, op_f2 y, op_f2* acc)
    {
        op_f2 t0, t1, t2, t3, u0, u1, u2, u3;
        asm volatile(OP_RD(t0, a, 0) OP_RD(t1, a, 8) OP_RD(t2, a, 16) OP_RD(t3, a, 24) OP_RD(u0, b, 0) OP_RD(u1, b, 8)
                     OP_RD(u2, b, 16) OP_RD(u3, b, 24) "s_waitcnt lgkmcnt(4)\n\t"
                     OP_MATH(t0, x, c0) OP_MATH(t1, x, c1) OP_MATH(t2, x, c2) OP_MATH(t3, x, c3) "s_waitcnt lgkmcnt(0)\n\t"
                     OP_MATH(u0, y, c0) OP_MATH(u1, y, c1) OP_MATH(u2, y, c2) OP_MATH(u3, y, c3)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u0] "=&v"(u0), [u1] "=&v"(u1),
                       [u2] "=&v"(u2), [u3] "=&v"(u3), [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3])
                     : [a] "v"(a), [x] "v"(x), [b] "v"(b), [y] "v"(y));
    }
};
template <> struct op_rounds<6> {
    static __device__ __forceinline__ void one(unsigned a, op_f2 x, op_f2* acc)
    {
        op_f2 t0, t1, t2, t3, t4, t5;
        asm volatile(OP_RD(t0, a, 0) OP_RD(t1, a, 8) OP_RD(t2, a, 16) OP_RD(t3, a, 24) OP_RD(t4, a, 32) OP_RD(t5, a, 40)
                     "s_waitcnt lgkmcnt(0)\n\t" OP_MATH(t0, x, c0) OP_MATH(t1, x, c1) OP_MATH(t2, x, c2) OP_MATH(t3, x, c3)
                     OP_MATH(t4, x, c4) OP_MATH(t5, x, c5)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
                       [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]), [c5] "+v"(acc[5])
                     : [a] "v"(a), [x] "v"(x));
    }
    static __device__ __forceinline__ void two(unsigned a, op_f2 x, unsigned b, op_f2 y, op_f2* acc)
    {
        op_f2 t0, t1, t2, t3, t4, t5, u0, u1, u2, u3, u4, u5;
        asm volatile(OP_RD(t0, a, 0) OP_RD(t1, a, 8) OP_RD(t2, a, 16) OP_RD(t3, a, 24) OP_RD(t4, a, 32) OP_RD(t5, a, 40)
                     OP_RD(u0, b, 0) OP_RD(u1, b, 8) OP_RD(u2, b, 16) OP_RD(u3, b, 24) OP_RD(u4, b, 32) OP_RD(u5, b, 40)
                     "s_waitcnt lgkmcnt(6)\n\t" OP_MATH(t0, x, c0) OP_MATH(t1, x, c1) OP_MATH(t2, x, c2) OP_MATH(t3, x, c3)
                     OP_MATH(t4, x, c4) OP_MATH(t5, x, c5) "s_waitcnt lgkmcnt(0)\n\t" OP_MATH(u0, y, c0) OP_MATH(u1, y, c1)
                     OP_MATH(u2, y, c2) OP_MATH(u3, y, c3) OP_MATH(u4, y, c4) OP_MATH(u5, y, c5)
                     : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
                       [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3), [u4] "=&v"(u4), [u5] "=&v"(u5),
                       [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]), [c3] "+v"(acc[3]), [c4] "+v"(acc[4]), [c5] "+v"(acc[5])
                     : [a] "v"(a), [x] "v"(x), [b] "v"(b), [y] "v"(y));
    }
};
template <> struct op_rounds<8> {
    static __device__ __forceinline__ void one(unsigned a, op_f2 x, op_f2* acc)
    {
        op_rounds<4>::one(a, x, acc);
        op_rounds<4>::one(a + 32u, x, acc + 4);
    }
    static __device__ __forceinline__ void two(unsigned a, op_f2 x, unsigned b, op_f2 y, op_f2* acc)
    {
        op_rounds<4>::two(a, x, b, y, acc);
        op_rounds<4>::two(a + 32u, x, b + 32u, y, acc + 4);
    }
};
#undef OP_RD
#undef OP_MATH

// NRT: rounds of 4 entries per column as a compile-time constant (13 for s = 51), or 0: read from fixed_s (up to 16)
template <typename IR, int NQ, int NRT>
__global__ __launch_bounds__(1024) void k_onepass(const char* __restrict__ rec, int R, int p, long long n, int fixed_s, int K,
                                                  const double* __restrict__ Cs, const double* __restrict__ xn1,
                                                  const double* __restrict__ xn2,
                                                  const unsigned long long* __restrict__ cmax_bits, int* __restrict__ assign,
                                                  float* __restrict__ bnd, long long npad, const double* __restrict__ cum,
                                                  int lib_valid, int* __restrict__ list, unsigned* __restrict__ nlist,
                                                  double* __restrict__ gsum, double* __restrict__ gcnt,
                                                  unsigned long long* __restrict__ nk, int chunk)
{
    constexpr int KP = 4 * NQ;
    constexpr int MAXR = NRT ? NRT : 16; // columns of up to 64 entries
    extern __shared__ __attribute__((aligned(16))) char smem_op[];
    float* tile = reinterpret_cast<float*>(smem_op);
    double* ssum = reinterpret_cast<double*>(smem_op + (size_t)(p + 1) * KP * 4);
    const int pk = K * p;
    unsigned* scnt = reinterpret_cast<unsigned*>(ssum + pk);
    unsigned* s_nk = scnt + (pk + 1) / 2;
    unsigned& s_amb = s_nk[K];     // (three more words of the dynamic segment: no static LDS in this kernel)
    unsigned& s_mov = s_nk[K + 1];
    unsigned& s_chg = s_nk[K + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int ps = lane >> 2, l4 = lane & 3;
    const int NR = NRT ? NRT : (fixed_s + 3) >> 2;
    const int nvl = fixed_s - 4 * (NR - 1); // entries of the last round (1..4); every other round is full
    // tile[r * KP + k] = -fl32(c_k[r] / gamma) (k_prep_tiles_f32's value); row p and the columns k >= K are zero
    for (int t = tid; t < (p + 1) * KP; t += blockDim.x) {
        const int r = t / KP, k = t - r * KP;
        tile[t] = (r < p && k < K) ? -(float)Cs[(size_t)r * K + k] : 0.f;
    }
    for (int t = tid; t < pk; t += blockDim.x) ssum[t] = 0.0;
    for (int t = tid; t < (pk + 1) / 2; t += blockDim.x) scnt[t] = 0u;
    for (int t = tid; t < K; t += blockDim.x) s_nk[t] = 0u;
    if (tid == 0) { s_amb = 0u; s_mov = 0u; s_chg = 0u; }
    __syncthreads();
    float* ubv = bnd;
    float* lbv = bnd + npad;
    int* alib = reinterpret_cast<int*>(bnd + 2 * npad);
    const double cum_now = cum ? *cum : 0.0;
    const double cmax = __builtin_bit_cast(double, *cmax_bits);
    const double u = 0x1p-24;
    const double eu = (2.0 * u + u * u) * (1.0 + 1e-9);
    const double gacc = (double)(fixed_s + 1) * u * (1.0 + 1e-4);
    const double nu = 0x1p-45;
    unsigned nambig = 0, nmov = 0;
    bool changed = false;
    for (long long c = blockIdx.x; c * chunk < n; c += gridDim.x) {
        const long long c0 = c * (long long)chunk;
        const long long c1 = (c0 + chunk < n) ? c0 + chunk : n;
        const int nsteps = (int)((c1 - c0 + 15) >> 4);
        for (int t = wave; t < nsteps; t += nwaves) {
            const long long i = c0 + 16LL * t + ps;
            const bool valid = i < c1;
            const long long ic = valid ? i : c1 - 1;
            const char* rb = rec + (size_t)ic * (size_t)R;
            const double* xd = reinterpret_cast<const double*>(rb);
            const IR* rd = reinterpret_cast<const IR*>(rb + (size_t)fixed_s * 8);
            // every load of the step up front, unconditionally from a clamped entry (a load inside a divergent branch is
            // waited for at the end of the branch)
            double x[MAXR];
            int row[MAXR];
#pragma unroll
            for (int r = 0; r < MAXR; r++) {
                if (r < NR) {
                    const bool okr = (r < NR - 1) || l4 < nvl;
                    const int ec = okr ? 4 * r + l4 : fixed_s - 1;
                    const double xv = __builtin_nontemporal_load(xd + ec);
                    const int rv = (int)__builtin_nontemporal_load(rd + ec);
                    x[r] = okr ? xv : 0.0;
                    row[r] = okr ? rv : p;
                } else { x[r] = 0.0; row[r] = p; }
            }
            const double n1 = xn1[ic], n2 = xn2[ic];
            const int old = lib_valid ? alib[ic] : -1;
            op_f2 acc[2 * NQ];
#pragma unroll
            for (int q = 0; q < 2 * NQ; q++) acc[q] = (op_f2){0.f, 0.f};
            // (the tile's LDS byte address, for the hand-placed reads)
            const unsigned tile_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_op;
#pragma unroll
            for (int r = 0; r < MAXR; r += 2) {
                if (r < NR) {
                    const unsigned a0 = tile_base + (unsigned)row[r] * (unsigned)(KP * 4);
                    const op_f2 x0 = (op_f2){(float)x[r], 0.f};
                    if (r + 1 < MAXR && r + 1 < NR) {
                        const unsigned a1 = tile_base + (unsigned)row[r + 1 < MAXR ? r + 1 : r] * (unsigned)(KP * 4);
                        const op_f2 x1 = (op_f2){(float)x[r + 1 < MAXR ? r + 1 : r], 0.f};
                        op_rounds<2 * NQ>::two(a0, x0, a1, x1, acc);
                    } else
                        op_rounds<2 * NQ>::one(a0, x0, acc);
                }
            }
            // the quad's four partial sums per centroid; smallest / second smallest / first argmin over k < K
            float lo = __builtin_inff(), hi = __builtin_inff();
            int klo = -1;
            bool bad = false;
#pragma unroll
            for (int q = 0; q < 2 * NQ; q++) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int k = 2 * q + h;
                    const float v = op_quad_sum(h ? acc[q].y : acc[q].x);
                    if (k < K) { // (uniform; selects, no branches: ascending k, the first smallest wins)
                        bad = bad || !(v == v);
                        const bool less = v < lo;
                        const float mid = less ? lo : v; // the larger of (lo, v), NaN-safe enough: a NaN sets `bad`
                        hi = mid < hi ? mid : hi;
                        klo = less ? k : klo;
                        lo = less ? v : lo;
                    }
                }
            }
            const double W = (n2 + 2.0 * cmax * n1 + (double)fixed_s * cmax * cmax) * (1.0 + 1e-9);
            const double E = eu * sqrt(W) * (1.0 + 1e-9);
            const double r1 = sqrt((double)lo), r2 = sqrt((double)hi);
            const double e1 = E + gacc * r1 + 1e-20, e2 = E + gacc * r2 + 1e-20;
            const bool certified = !bad && klo >= 0 && ((r1 + e1) * (1.0 + nu) < (r2 - e2) * (1.0 - nu));
            const int newk = klo >= 0 ? klo : 0;
            const bool head = l4 == 0 && valid;
            if (head) {
                assign[i] = newk; // (the caller's buffer; tentative for an uncertified point)
                if (certified) {
                    if (old != newk) {
                        alib[i] = newk;
                        if (lib_valid) { changed = true; nmov++; }
                    }
                    ubv[i] = __double2float_ru((r1 + e1) * (1.0 + nu) * (1.0 + 1e-12));
                    atomicAdd(&s_nk[newk], 1u);
                }
                lbv[i] = __double2float_rd((certified ? fmax(0.0, (r2 - e2) * (1.0 - nu)) : 0.0) + cum_now);
                if (!(r2 >= 2.25 * r1)) nambig++;
            }
            {   // uncertified points -> the exact list (one atomic per wave)
                const unsigned long long um = __ballot(head && !certified);
                if (um) {
                    unsigned at = 0;
                    if (lane == __builtin_ctzll(um)) at = atomicAdd(nlist, (unsigned)__popcll(um));
                    at = (unsigned)__builtin_amdgcn_readlane((int)at, __builtin_ctzll(um));
                    if (head && !certified) list[at + __popcll(um & ((1ull << lane) - 1ull))] = (int)i;
                }
            }
            if (certified && valid) {
                const int base = newk * p;
#pragma unroll
                for (int r = 0; r < MAXR; r++) {
                    if (r < NR && ((r < NR - 1) || l4 < nvl)) {
                        const int idx = base + row[r];
                        unsafeAtomicAdd(&ssum[idx], x[r]);
                        atomicAdd(&scnt[idx >> 1], 1u << ((idx & 1) << 4));
                    }
                }
            }
        }
        // the chunk's sums and counts into the global ones; the slab starts the next chunk empty
        __syncthreads();
        for (int t = tid; t < (pk + 1) / 2; t += blockDim.x) {
            const unsigned cc = scnt[t];
            if (cc) {
                scnt[t] = 0u;
                const unsigned c_lo = cc & 0xffffu, c_hi = cc >> 16;
                if (c_lo) { unsafeAtomicAdd(&gsum[2 * t], ssum[2 * t]); unsafeAtomicAdd(&gcnt[2 * t], (double)c_lo); ssum[2 * t] = 0.0; }
                if (c_hi) { unsafeAtomicAdd(&gsum[2 * t + 1], ssum[2 * t + 1]); unsafeAtomicAdd(&gcnt[2 * t + 1], (double)c_hi); ssum[2 * t + 1] = 0.0; }
            }
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) { nmov += __shfl_down(nmov, off); nambig += __shfl_down(nambig, off); }
    if (lane == 0) { if (nmov) atomicAdd(&s_mov, nmov); if (nambig) atomicAdd(&s_amb, nambig); }
    if (__any(changed) && lane == 0) s_chg = 1u;
    __syncthreads();
    if (tid == 0) {
        if (s_amb) atomicAdd(nlist + 1, s_amb);
        if (s_chg) atomicAdd(nlist + 5, 1u);
        if (s_mov) atomicAdd(nlist + 14, s_mov);
    }
    for (int k = tid; k < K; k += blockDim.x)
        if (s_nk[k]) atomicAdd(&nk[k], (unsigned long long)s_nk[k]);
}

// The listed points of a one-pass call, after k_assign_list has their exact clusters: their entries into the global sums
// and counts, their clusters' sizes; one wave per point, lanes over its entries.  Thread 0 also books the call's
// "points streamed by the accumulation" (counters[13], running total at counters[32..33]: every point, in this form).
template <typename IR>
__global__ __launch_bounds__(256) void k_onepass_listed(const char* __restrict__ rec, int R, int p, long long n, int fixed_s,
                                                        const int* __restrict__ list, unsigned* __restrict__ counters,
                                                        const int* __restrict__ assign, double* __restrict__ gsum,
                                                        double* __restrict__ gcnt, unsigned long long* __restrict__ nk)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(counters + 32), (unsigned long long)n);
        counters[13] = (unsigned)(n > 0xffffffffLL ? 0xffffffffLL : n);
    }
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long cnt = counters[0];
    for (long long q = wave; q < cnt; q += nwaves) {
        const long long i = list[q];
        const int a = assign[i];
        const char* rb = rec + (size_t)i * (size_t)R;
        const double* xd = reinterpret_cast<const double*>(rb);
        const IR* rd = reinterpret_cast<const IR*>(rb + (size_t)fixed_s * 8);
        for (int e = lane; e < fixed_s; e += 64) {
            const size_t idx = (size_t)a * p + (size_t)rd[e];
            unsafeAtomicAdd(&gsum[idx], xd[e]);
            unsafeAtomicAdd(&gcnt[idx], 1.0);
        }
        if (lane == 0) atomicAdd(&nk[a], 1ull);
    }
}
