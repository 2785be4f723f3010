// Certified f32 screen + exact f64 confirmation: the fast path of the fused Lloyd iteration
// for K >= 2 (gfx950).  RESULTS ARE IDENTICAL to the exact kernel of assign.hip -- assignments
// bit-for-bit, min-distances bit-for-bit -- because nothing computed in f32 is ever output:
//
//   1. k_screen_quad     (columns of up to 64 entries; k_screen_tile, the 16-lanes-per-point kernel, beyond)
//                        for every point and centroid, an f32 estimate of the squared distance, 32 centroids
//                        per LDS tile; per (point, tile): the leader's estimate m1, its centroid, and m2, a
//                        LOWER BOUND of the estimate of every other centroid of the tile (the second smallest
//                        estimate -- or, in the two-phase form, the second smallest partial sum: partial sums
//                        of the non-negative terms bound the full sums from below because f32 addition is
//                        monotone, and only the leader by partial sum is summed to the end).
//   2. k_combine_screen  best / second-best over the tiles and a RIGOROUS bound on
//                        |sqrt(estimate) - true distance| (below).  If best + bound < second - bound
//                        the reference's argmin is certified (uniquely: no tie can occur inside the
//                        gap); otherwise the point is appended to a list.
//   3. k_assign_list     listed points: exact reference arithmetic over all K centroids
//                        (SparseMatrixMinusCluster.c:173-180 + first-index min).
//   4. k_exact_accumulate  (after the counting sort by cluster) per point, the distance to ITS centroid
//                        in exact reference arithmetic -- the value the reference's min() returns --
//                        fused with the per-cluster sum / count accumulation that reads the same entries.
//
// Error bound (u = 2^-24).  Reference value for centroid k: dist_k = fl64(sqrt(sum_j fl64((x_j-c_jk)^2))), c = C/gamma
// in f64; true distance D_k = ||x - c_k|| over the point's support.  The screen uses x~ = fl32(x),
// c~ = fl32(c), t~_j = fl32(x~_j - c~_jk):  |t~_j - (x_j - c_jk)| <= (2u+u^2)(|x_j|+|c_jk|) =: e_j, so by the
// triangle inequality | ||t~|| - D_k | <= E := sqrt(sum e_j^2) <= (2u+u^2) sqrt(W),
// W = sum_j (|x_j| + Cmax)^2 = xn2 + 2 Cmax xn1 + s Cmax^2 <= (sqrt(xn2) + sqrt(s) Cmax)^2  (xn1 = sum |x_j| <= sqrt(s xn2),
// Cauchy-Schwarz over the column's <= s entries): ONE per-point value, xnr >= sqrt(sum x_j^2) rounded up to f32, is kept
// (4 B per point in the certification pass instead of the 16 B of xn1 and xn2 in f64; E grows by < 25 %, of 1e-7 r).
// The f32 FMA accumulation of s terms gives a~ in ||t~||^2 (1 +- g), g = (s+1)u(1+1e-4), hence
// |sqrt(a~) - ||t~||| <= g sqrt(a~).  Together |sqrt(a~_k) - D_k| <= eps_k := E + g sqrt(a~_k) + 1e-20
// (the last term covers f32 subnormal products).  dist_k itself is within D_k (1 +- 2^-45).
// Certified iff (r1 + eps_1)(1+2^-45) < (r2 - eps_2)(1-2^-45) with r1 = sqrt(a~) of the best leader and r2 = sqrt of
// the smallest of: the other tiles' leaders, every tile's m2.  Every other centroid has sqrt(a~_k) >= r2 (its
// full sum is at least any lower bound of it) and r - eps(r) is increasing in r, so dist_1 < dist_k for all k.
// Overflow makes a~ = inf and fails the test (-> list).  Empty / duplicate columns fail it (-> list).
#include "common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

// tail of a shard's bounds buffer (floats) behind ub[npad] | lb[npad] | assignment[npad]:
//   delta[K] | dmax              drift of every centroid on a point's support (k_center_drift) and its maximum
//   hterm[K] at HB_HTERM         hint_w x (full 2-norm drift)^2 per centroid: the hints' estimate, not a bound
#define HB_KMAX 65536
#define HB_HTERM (HB_KMAX + 16)
#define HB_TAIL (2 * HB_KMAX + 32)

// T32[g][r][kk] = -fl32(C[(g*32+kk)*p + r] / gamma), row p zero; cmax_bits = max |C/gamma| (f64 bits, atomicMax).
// 4-lanes-per-point kernel only:
//  * pl_last = 1 / 2: the last tile holds <= 16 centroids in floats 0..15 of each row and a second copy of them
//    in floats 16..31 (the two point pairs of an LDS phase read different copies);
//  * pl_last = 5: the last <= 4 centroids are carried by the workgroups of tile G-2; the buffer of "tile" G-1
//    then holds their (p+1) x 4 table E[r][j] = -fl32(C[(32 (G-1) + j)*p + r] / gamma) (16-B rows).
//  * swz != 0: within each 64-B half-row the four 16-B pieces are permuted by the row: logical piece q sits at
//    q ^ ((r >> 1) & 3) (the kernel's stored row ids carry the same two bits, k_screen_reorder).
__global__ void k_prep_tiles_f32(const double* __restrict__ C, int p, int K, int G, double gamma,
                                 float* __restrict__ T32, unsigned long long* __restrict__ cmax_bits, int pl_last,
                                 int swz, double* __restrict__ Cs, double* __restrict__ keep)
{
    // Two more views of the same centres ride along (they were launches of their own: a small shard's settled iteration is
    // a chain of ~5-us kernels): Cs[r*K + k] = C[k*p + r] / gamma, row-major for the exact list kernel (k_assign_list),
    // and keep = C itself, the library's copy that the NEXT call's drift is measured from (k_center_drift has already
    // read the previous copy: it is launched before this kernel).
    if (Cs != nullptr || keep != nullptr) {
        const size_t pkk = (size_t)p * K;
        for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < pkk; t += (size_t)gridDim.x * blockDim.x) {
            if (keep != nullptr) keep[t] = C[t];
            if (Cs != nullptr) {
                const int k = (int)(t % K);
                const size_t r = t / K;
                double v = C[(size_t)k * p + r];
                if (gamma > 0.0) v = v / gamma;
                Cs[t] = v;
            }
        }
    }
    const size_t total = (size_t)G * (p + 1) * SCREEN_KT;
    double mx = 0.0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        int kk = (int)(t % SCREEN_KT);
        const size_t rest = t / SCREEN_KT;
        int r = (int)(rest % (p + 1));
        const int g = (int)(rest / (p + 1));
        if (swz) kk ^= ((r >> 1) & 3) << 2; // physical -> logical position
        int k = g * SCREEN_KT + kk;
        if (g == G - 1 && pl_last == 5) {
            const size_t local = t - (size_t)g * (p + 1) * SCREEN_KT; // compact (p+1) x 4 table first, rest unused
            r = (int)(local >> 2);
            k = local < (size_t)(p + 1) * 4 ? g * SCREEN_KT + (int)(local & 3) : K;
        } else if (g == G - 1 && pl_last < 4) {
            kk &= 15;
            k = g * SCREEN_KT + kk;
        }
        float v = 0.f;
        if (r < p && k < K) {
            double c = C[(size_t)k * p + r];
            if (gamma > 0.0) c = c / gamma;
            mx = fmax(mx, fabs(c));
            v = -(float)c;
        }
        T32[t] = v;
    }
    // one atomic per workgroup (2000 waves on one address were most of this kernel's 20 us)
    __shared__ double s_mx[16];
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off));
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) mx = fmax(mx, s_mx[w]);
        if (mx > 0.0) atomicMax(cmax_bits, __builtin_bit_cast(unsigned long long, mx));
    }
}

// (the f64 sum of s squares is within s 2^-53 of the true one: 1e-12 covers it and the sqrt; NaN / inf stay NaN / inf and
//  fail every certificate)
__device__ __forceinline__ float point_norm_up(double sumsq) { return __double2float_ru(sqrt(sumsq) * (1.0 + 1e-12)); }

// xnr[i] >= sqrt(sum_j x_j^2) over column i, rounded up to f32 (any summation order: used only inside an upper bound),
// and the f32 copy of the values the screen streams (4 B instead of 8 B per entry).
// 16 lanes per point so that the loads of a wave cover four contiguous columns.
__global__ __launch_bounds__(256) void k_point_norms(const long long* __restrict__ jc, const double* __restrict__ x,
                                                     long long n, int fixed_s, float* __restrict__ xnr,
                                                     float* __restrict__ xf)
{
    const int sub = threadIdx.x & 15;
    const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
    const long long rounds = (n + ng - 1) / ng;
    for (long long t = 0; t < rounds; t++) {
        const long long i = g0 + t * ng;
        double b = 0.0;
        if (i < n) {
            const long long j0 = fixed_s > 0 ? i * fixed_s : jc[i];
            const long long j1 = fixed_s > 0 ? j0 + fixed_s : jc[i + 1];
            for (long long j = j0 + sub; j < j1; j += 16) {
                const double v = x[j];
                b += v * v;
                if (xf) xf[j] = (float)v; // the screen's x~ = fl32(x)
            }
        }
        for (int off = 8; off > 0; off >>= 1) b += __shfl_xor(b, off);
        if (i < n && sub == 0) xnr[i] = point_norm_up(b);
    }
}

// maximum over the 16 lanes of a DPP row, in every lane (v_max_i32_dpp: the compiler's update_dpp + max takes three
// instructions per stage)
__device__ __forceinline__ int row_max16_i32(int v)
{
    asm("s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}

// The screen's own copy of a fixed-stride shard, laid out the way the 4-lanes-per-point kernel consumes it.
//  * values as f32;
//  * per point, the entries ordered BY |x| DESCENDING in three segments -- the first 4 A1, the next 4 (A2 - A1), the rest;
//    A1 = quad_split(NR), A2 = quad_split_late(NR): the rounds the two-phase forms evaluate for all centroids (policy.h).
//    The sum of squares the screen estimates does not depend on the order, and a partial sum of its non-negative terms is
//    a lower bound of the full sum whatever the order; the order decides how LARGE the partial sums are when the
//    two-phase forms look at them: a far centroid's term (x_j - c_j)^2 is x_j^2 + c_j^2 on average, so the largest |x_j|
//    first makes the competition clear a point's hint after a quarter of the rounds that storage order needs;
//  * inside a segment, PARTITIONED BY ROW PARITY -- points with an even index list their even rows first, odd points
//    their odd rows first: the order inside a segment is free, and it decides which LDS banks the kernel hits: the two
//    points that read the same half-row in one 16-lane phase then touch rows of opposite parity (= different 128-B
//    halves of the 64 banks) for all but the few rounds in which one of them has already switched class;
//  * STEP-MAJOR: a step is 16 consecutive points, a round 4 entries of each.  Element (step t, round r, lane L)
//    sits at (t * NR + r) * 64 + L, where lane L = 4 * (point - 16 t) + l4 holds entry 4 r + l4 of its point:
//    every load of a wave is one contiguous 256-B (values) / 128-B (16-bit row ids) piece, and a step that the hinted
//    form settles on its first rounds fetches only those pieces.  Slots past the
//    column (4 NR > fixed_s) and past the last point hold x = 0 on the all-zero row p;
//  * row ids are stored as row * 8 ^ ((row >> 1) & 3): shifted left by 4 that is the row's LDS offset (128-B
//    rows) with the tile's piece swizzle in bits 4..5 (k_prep_tiles_f32, swz), so the kernel's address
//    arithmetic stays one XOR per broadcast entry.  Needs 8 (p + 1) <= 65536 for 16-bit ids (LDS: p <= 1279).
//  map != nullptr (a regrouped shard, api_lloyd.hip): step-major slot i holds the point map[i] of the records.
template <typename IR>
__global__ __launch_bounds__(256) void k_screen_reorder(const IR* __restrict__ ir, const double* __restrict__ x,
                                                        long long n, int fixed_s, int p, float* __restrict__ xfs,
                                                        IR* __restrict__ irs, float* __restrict__ xnr,
                                                        const char* __restrict__ rec = nullptr,
                                                        int rec_R = 0, const int* __restrict__ map = nullptr)
{
    // rec != nullptr: the entries are read from the record layout (a shard that never had CSC arrays, or has let them go)
    // one step per workgroup pass: 16 lanes per point, up to 4 passes of 16 entries (fixed_s <= 64) held in
    // registers; the ordered columns are staged in LDS, then written out in lane order.
    // Round 6: straight-line code per step (the round-5 version ran ~1300 instructions per wave and step through 179 basic
    // blocks -- 12 group maxima of 23 instructions, 24 ballot-ranked placements behind a branch each -- and was VALU bound
    // at 2.7 TB/s).  Now: (1) each lane sorts its 4 keys, the e2 largest of the point come off the sorted heads (a 16-lane
    // DPP maximum + 4 conditional moves per extraction) and leave two THRESHOLDS -- keys are unique, so "key >= T" is the
    // segment test; (2) the six (segment, row-parity class) buckets are counted in one packed word per lane, ranked by ONE
    // 16-lane DPP prefix scan, the bucket bases come from the group total (ds_swizzle broadcast of lane 15).  Order inside
    // a bucket: by lane, then by pass (the round-5 order was by entry index; the order inside a class is free).
    __shared__ float s_x[16][64];
    __shared__ IR s_r[16][64];
    const int sub = threadIdx.x & 15;
    const int grp = threadIdx.x >> 4;
    const int NR = (fixed_s + 3) >> 2;
    const int e1 = 4 * quad_split(NR), e2 = 4 * (quad_split_late(NR) > quad_split(NR) ? quad_split_late(NR) : quad_split(NR));
    const long long nsteps = (n + 15) >> 4;
    for (long long st = blockIdx.x; st < nsteps; st += gridDim.x) {
        const long long i = st * 16 + grp;
        const bool live = i < n;
        const long long src = live ? (map != nullptr ? (long long)map[i] : i) : 0;
        const long long j0 = src * fixed_s;
        const unsigned want = (unsigned)(i & 1);
        float xf[4];
        unsigned rv[4];
        int key[4];
        double nb = 0.0;
        // (unconditional loads: a slot past the column, or a point past the end, re-reads entry 0 of a valid point)
        const char* rb = rec != nullptr ? rec + (size_t)src * (size_t)rec_R : nullptr;
        const double* xp = rec != nullptr ? reinterpret_cast<const double*>(rb) : x + j0;
        const IR* rp = rec != nullptr ? reinterpret_cast<const IR*>(rb + (size_t)fixed_s * 8) : ir + j0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = u * 16 + sub;
            const bool ok = live && e < fixed_s;
            const int ec = e < fixed_s ? e : 0;
            const double xl = xp[ec];
            const unsigned rl = (unsigned)rp[ec];
            const double xv = ok ? xl : 0.0;
            rv[u] = ok ? rl : (unsigned)p;          // no entry: x = 0 on the all-zero row p
            nb += xv * xv;
            xf[u] = (float)xv;
            // |x| as f32 bits (monotone for non-negative floats, NaN above everything) with the low 6 bits replaced by the
            // entry's index: unique per entry, a total order whatever the data (the 2^-17 it moves a value by decides nothing
            // but the order of near-equal entries); -1: no entry
            key[u] = ok ? (((__builtin_bit_cast(int, xf[u]) & 0x7fffffff) & ~63) | (63 - e)) : -1;
        }
        // (1) thresholds: T1 = the e1-th largest key of the point, T2 = the e2-th largest (-1 when the point has fewer entries)
        int a0 = max(key[0], key[1]), a1 = min(key[0], key[1]), a2 = max(key[2], key[3]), a3 = min(key[2], key[3]);
        { const int h = max(a0, a2), l = min(a0, a2); a0 = h; a2 = l; }
        { const int h = max(a1, a3), l = min(a1, a3); a1 = h; a3 = l; }
        { const int h = max(a1, a2), l = min(a1, a2); a1 = h; a2 = l; }
        int T1 = -1, T2 = -1;
        auto extract = [&]() {               // the largest key still at a head of the point's 16 lanes; its lane moves up
            const int m = row_max16_i32(a0);
            const bool own = a0 == m;        // (m = -1: the point has run out of entries -- every head is -1 and stays so)
            a0 = own ? a1 : a0;
            a1 = own ? a2 : a1;
            a2 = own ? a3 : a2;
            a3 = own ? -1 : a3;
            return m;
        };
        for (int t = 0; t < e1; t++) T1 = extract();
        T2 = T1;
        for (int t = e1; t < e2; t++) T2 = extract();
        if (e1 == 0) T1 = 0x7fffffff;        // (no early split compiled for this round count: nothing is in segment 0)
        if (e2 == 0) T2 = 0x7fffffff;
        // (2) buckets b = 2 * segment + class (class 0 = the point's own row parity) for the entries, b = 6 for the empty
        // slots (they follow the entries: x = 0 on row p), counted per lane in packed 8-bit fields: buckets 0..3 in word A,
        // 4..6 in word B.  Every one of the point's 64 slots is written exactly once.
        int shv[4];
        unsigned cA = 0u, cB = 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int sg = (key[u] < T1 ? 1 : 0) + (key[u] < T2 ? 1 : 0);
            const int b = key[u] >= 0 ? 2 * sg + (int)((rv[u] ^ want) & 1u) : 6;
            shv[u] = 8 * b;
            cA += b < 4 ? 1u << (shv[u] & 31) : 0u;
            cB += b < 4 ? 0u : 1u << (shv[u] & 31);
        }
        unsigned iA = cA, iB = cB; // inclusive prefix over the point's 16 lanes (fields cannot overflow: 64 slots in all)
        iA += (unsigned)__builtin_amdgcn_update_dpp(0, (int)iA, 0x111, 0xf, 0xf, true); // row_shr:1
        iB += (unsigned)__builtin_amdgcn_update_dpp(0, (int)iB, 0x111, 0xf, 0xf, true);
        iA += (unsigned)__builtin_amdgcn_update_dpp(0, (int)iA, 0x112, 0xf, 0xf, true); // row_shr:2
        iB += (unsigned)__builtin_amdgcn_update_dpp(0, (int)iB, 0x112, 0xf, 0xf, true);
        iA += (unsigned)__builtin_amdgcn_update_dpp(0, (int)iA, 0x114, 0xf, 0xf, true); // row_shr:4
        iB += (unsigned)__builtin_amdgcn_update_dpp(0, (int)iB, 0x114, 0xf, 0xf, true);
        iA += (unsigned)__builtin_amdgcn_update_dpp(0, (int)iA, 0x118, 0xf, 0xf, true); // row_shr:8
        iB += (unsigned)__builtin_amdgcn_update_dpp(0, (int)iB, 0x118, 0xf, 0xf, true);
        const unsigned tA = (unsigned)__builtin_amdgcn_ds_swizzle((int)iA, 0x1F0); // lane 15 of the 16-lane row: the point's totals
        const unsigned tB = (unsigned)__builtin_amdgcn_ds_swizzle((int)iB, 0x1F0);
        // bucket bases (exclusive prefix over the buckets) packed the same way, + the lanes below = where this lane's first
        // slot of each bucket goes
        const unsigned n0 = tA & 0xffu, n1 = (tA >> 8) & 0xffu, n2 = (tA >> 16) & 0xffu, n3 = tA >> 24, n4 = tB & 0xffu, n5 = (tB >> 8) & 0xffu;
        const unsigned s1 = n0, s2 = s1 + n1, s3 = s2 + n2, s4 = s3 + n3, s5 = s4 + n4, s6 = s5 + n5;
        unsigned pA = (iA - cA) + ((s1 << 8) | (s2 << 16) | (s3 << 24));
        unsigned pB = (iB - cB) + (s4 | (s5 << 8) | (s6 << 16));
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool inA = shv[u] < 32;
            const int sh = shv[u] & 31;
            const int pos = (int)(((inA ? pA : pB) >> sh) & 63u);
            pA += inA ? 1u << sh : 0u;
            pB += inA ? 0u : 1u << sh;
            s_x[grp][pos] = xf[u];
            s_r[grp][pos] = (IR)((rv[u] << 3) ^ ((rv[u] >> 1) & 3u)); // LDS row offset / 16 with the tile swizzle folded in
        }
        __syncthreads();
        float* xo = xfs + (size_t)st * NR * 64;
        IR* ro = irs + (size_t)st * NR * 64;
#pragma unroll
        for (int k = 0; k < 4; k++) { // NR <= 16: at most 1024 elements per step
            const int idx = (int)threadIdx.x + 256 * k;
            if (idx < NR * 64) {
                const int r = idx >> 6, L = idx & 63;
                const int e = 4 * r + (L & 3);
                xo[idx] = s_x[L >> 2][e];
                ro[idx] = s_r[L >> 2][e];
            }
        }
        __syncthreads();
        // the certificate's per-point norm (root of sum x^2, rounded up; any order) rides on the same pass over x
        for (int off = 8; off > 0; off >>= 1) nb += __shfl_xor(nb, off);
        if (live && sub == 0 && xnr) xnr[i] = point_norm_up(nb);
    }
}

template <int N, int IRB>
struct RunScreenP {
    static __device__ __forceinline__ void run(int nb, int koff, int roffA, double xA, int roffB, double xB,
                                               double& accA, double& accB, const void* xbase, const void* rbase,
                                               unsigned voxA, unsigned voxB, unsigned vorA, unsigned vorB,
                                               float& xAn, float& xBn, int& rAn, int& rBn)
    {
        if (nb == N)
            screen2p<N, IRB>(koff, roffA, xA, roffB, xB, accA, accB, xbase, rbase, voxA, voxB, vorA, vorB, xAn, xBn,
                             rAn, rBn);
        else
            RunScreenP<N - 1, IRB>::run(nb, koff, roffA, xA, roffB, xB, accA, accB, xbase, rbase, voxA, voxB, vorA,
                                        vorB, xAn, xBn, rAn, rBn);
    }
};
template <int IRB>
struct RunScreenP<0, IRB> {
    static __device__ __forceinline__ void run(int, int, int, double, int, double, double&, double&, const void*,
                                               const void*, unsigned, unsigned, unsigned, unsigned, float&, float&,
                                               int&, int&) {}
};

template <int N, int IRB>
struct RunScreen32P {
    static __device__ __forceinline__ void run(int nb, int koff, double rpA, double xpA, double rpB, double xpB,
                                               double& accA, double& accB, const void* xbase, const void* rbase,
                                               unsigned voxA, unsigned voxB, unsigned vorA, unsigned vorB,
                                               double& xAn, double& xBn, int& rA0n, int& rA1n, int& rB0n, int& rB1n)
    {
        if (nb == N)
            screen32p<N, IRB>(koff, rpA, xpA, rpB, xpB, accA, accB, xbase, rbase, voxA, voxB, vorA, vorB, xAn, xBn,
                              rA0n, rA1n, rB0n, rB1n);
        else
            RunScreen32P<N - 1, IRB>::run(nb, koff, rpA, xpA, rpB, xpB, accA, accB, xbase, rbase, voxA, voxB, vorA,
                                          vorB, xAn, xBn, rA0n, rA1n, rB0n, rB1n);
    }
};
template <int IRB>
struct RunScreen32P<0, IRB> {
    static __device__ __forceinline__ void run(int, int, double, double, double, double, double&, double&,
                                               const void*, const void*, unsigned, unsigned, unsigned, unsigned,
                                               double&, double&, int&, int&, int&, int&) {}
};

// two 32-bit values in one 64-bit register (low, high)
__device__ __forceinline__ double pack2(unsigned lo, unsigned hi)
{
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// min over the 16-lane DPP row with the DPP operand folded into v_min_f32 (one instruction per
// stage; IEEE minnum, so NaNs lose against numbers)
__device__ __forceinline__ float row_min16_f32(float v)
{
    asm("v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}

// (x, x) as one 64-bit register for v_pk_add_f32
__device__ __forceinline__ double pack_xx(float x)
{
    const unsigned b = __builtin_bit_cast(unsigned, x);
    return __builtin_bit_cast(double, ((unsigned long long)b << 32) | b);
}

// smallest / second smallest of the 32 estimates of one point (16 lanes x 2) and the centroid of the smallest
__device__ __forceinline__ void store_screen_winner(double acc2, int slot, int kk, int kbase, int K, int i, bool valid,
                                                    float* __restrict__ m1o, float* __restrict__ m2o,
                                                    int* __restrict__ ko)
{
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, acc2);
    float a0 = __builtin_bit_cast(float, (unsigned)bits), a1 = __builtin_bit_cast(float, (unsigned)(bits >> 32));
    const int k0 = kbase + 2 * kk;
    if (kbase + SCREEN_KT > K) { // wave-uniform: only the last tile has padding slots
        if (k0 >= K) a0 = __builtin_inff();
        if (k0 + 1 >= K) a1 = __builtin_inff();
    }
    const float lo = fminf(a0, a1), hi = fmaxf(a0, a1);
    float m1 = lo;
    asm volatile("s_nop 1" : "+v"(m1)); // VALU write -> DPP read of the same VGPR: 2 wait states
    m1 = row_min16_f32(m1);
    const bool mine = (lo == m1);
    const unsigned long long seg = (__ballot(mine) >> (slot * 16)) & 0xffffull;
    const int first = __builtin_ctzll(seg | (1ull << 63));
    const bool none = seg == 0ull; // NaN estimates are never equal to anything: report "no candidate"
    // second smallest: the winning lane contributes its OTHER value, every other lane its smaller one
    float m2 = (kk == first) ? hi : lo;
    asm volatile("s_nop 1" : "+v"(m2));
    m2 = row_min16_f32(m2);
    if ((none ? kk == 0 : kk == first) && valid) {
        m1o[i] = none ? __builtin_inff() : m1;
        m2o[i] = none ? __builtin_inff() : m2;
        ko[i] = none ? -1 : ((a0 == m1) ? k0 : k0 + 1);
    }
}

// The f32 screen over one 32-centroid tile.  Same geometry as k_assign_tile<16,.,FIXED=true>
// (workgroup = one tile in LDS, waves draw pairs of 4-point groups from an LDS ticket), fixed-stride
// shards only.  T32 tile: (p+1) rows of 32 floats = the same 128 B per row as the f64 tile.
template <typename IR>
__global__ __launch_bounds__(1024) void k_screen_tile(
    const IR* __restrict__ ir, const float* __restrict__ xval, const float* __restrict__ T32, int p, int n,
    int fixed_s, int K, const spkm_blockmap* __restrict__ bmap, int chunk_points, float* __restrict__ scr_m1,
    float* __restrict__ scr_m2, int* __restrict__ scr_k)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spkm_blockmap bm = bmap[blockIdx.x];
    if (bm.tile < 0) return;
    const int g = bm.tile;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const size_t tile_bytes = (size_t)(p + 1) * SCREEN_KT * 4;
    {
        const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T32) + (size_t)g * tile_bytes);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (size_t t = tid; t < tile_bytes / 16; t += nthreads) dst[t] = src[t];
        if (tid == 0) *reinterpret_cast<unsigned*>(smem + tile_bytes) = 0u;
    }
    __syncthreads();

    constexpr int PPW = 4, PPS = 8, IRB = (int)sizeof(IR);
    const int lane = tid & 63;
    const int nwaves = nthreads >> 6;
    (void)nwaves;
    const int slot = lane >> 4;
    const int kk = lane & 15;
    const int koff = kk * 8;
    const int jl = kk;
    const int nchunks = (n + chunk_points - 1) / chunk_points;
    float* m1o = scr_m1 + (size_t)g * n;
    float* m2o = scr_m2 + (size_t)g * n;
    int* ko = scr_k + (size_t)g * n;

    const int R = chunk_points / PPS;
    const int my_chunks = (nchunks > bm.stream) ? (nchunks - bm.stream + bm.nstreams - 1) / bm.nstreams : 0;
    const int T = my_chunks * R;
    // batches of 32 entries: lane l of a row owns entries 2l, 2l+1 of the batch (screen32p)
    const int nbatches = (fixed_s + 31) >> 5;
    const int tail = fixed_s - 32 * (nbatches - 1);
    unsigned* ticket = reinterpret_cast<unsigned*>(smem + tile_bytes);
    auto draw = [&]() {
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(ticket, 1u);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };
    auto base_of = [&](int u) {
        const int ci = u / R, r = u - ci * R;
        return (bm.stream + ci * bm.nstreams) * chunk_points + r * PPS;
    };
    auto lane_offs = [&](int base, unsigned& voxA, unsigned& voxB, unsigned& vorA, unsigned& vorB) {
        const int iA = base + slot, iB = base + PPW + slot;
        const int cA = iA < n ? iA : n - 1, cB = iB < n ? iB : n - 1;
        const unsigned eA = (unsigned)(cA - base) * (unsigned)fixed_s + 2u * (unsigned)jl;
        const unsigned eB = (unsigned)(cB - base) * (unsigned)fixed_s + 2u * (unsigned)jl;
        voxA = eA * 4u; voxB = eB * 4u;
        vorA = eA * (unsigned)IRB; vorB = eB * (unsigned)IRB;
    };
    int t = draw();
    int tn = draw();
    int base = (t < T) ? base_of(t) : n;
    if (base < n && base >= 0) {
        unsigned voxA, voxB, vorA, vorB;
        lane_offs(base, voxA, voxB, vorA, vorB);
        const char* xb = reinterpret_cast<const char*>(xval + (size_t)base * (size_t)fixed_s);
        const char* rb = reinterpret_cast<const char*>(ir + (size_t)base * (size_t)fixed_s);
        auto ldx = [&](const char* b_, unsigned off) {
            const float* q = reinterpret_cast<const float*>(b_ + off);
            return pack2(__builtin_bit_cast(unsigned, q[0]), __builtin_bit_cast(unsigned, q[1]));
        };
        double xpA = ldx(xb, voxA), xpB = ldx(xb, voxB);
        int rA0 = (int)reinterpret_cast<const IR*>(rb + vorA)[0], rA1 = (int)reinterpret_cast<const IR*>(rb + vorA)[1];
        int rB0 = (int)reinterpret_cast<const IR*>(rb + vorB)[0], rB1 = (int)reinterpret_cast<const IR*>(rb + vorB)[1];
        while (true) {
            int nbase = (tn < T) ? base_of(tn) : n;
            const bool more = (nbase < n) && (nbase >= 0);
            if (!more) nbase = base;
            unsigned nvoxA, nvoxB, nvorA, nvorB;
            lane_offs(nbase, nvoxA, nvoxB, nvorA, nvorB);
            const char* nxb = reinterpret_cast<const char*>(xval + (size_t)nbase * (size_t)fixed_s);
            const char* nrb = reinterpret_cast<const char*>(ir + (size_t)nbase * (size_t)fixed_s);
            double accA = 0.0, accB = 0.0; // two +0.0f each
            double xAn, xBn;
            int rA0n, rA1n, rB0n, rB1n;
            for (int b = 0; b + 1 < nbatches; b++) {
                xb += 32 * 4;
                rb += 32 * IRB;
                screen32p<32, IRB>(koff, pack2((unsigned)rA0 * 128u, (unsigned)rA1 * 128u), xpA,
                                   pack2((unsigned)rB0 * 128u, (unsigned)rB1 * 128u), xpB, accA, accB, xb, rb, voxA,
                                   voxB, vorA, vorB, xAn, xBn, rA0n, rA1n, rB0n, rB1n);
                xpA = xAn; xpB = xBn; rA0 = rA0n; rA1 = rA1n; rB0 = rB0n; rB1 = rB1n;
            }
            if (tail == 32)
                screen32p<32, IRB>(koff, pack2((unsigned)rA0 * 128u, (unsigned)rA1 * 128u), xpA,
                                   pack2((unsigned)rB0 * 128u, (unsigned)rB1 * 128u), xpB, accA, accB, nxb, nrb, nvoxA,
                                   nvoxB, nvorA, nvorB, xAn, xBn, rA0n, rA1n, rB0n, rB1n);
            else
                RunScreen32P<31, IRB>::run(tail, koff, pack2((unsigned)rA0 * 128u, (unsigned)rA1 * 128u), xpA,
                                           pack2((unsigned)rB0 * 128u, (unsigned)rB1 * 128u), xpB, accA, accB, nxb, nrb,
                                           nvoxA, nvoxB, nvorA, nvorB, xAn, xBn, rA0n, rA1n, rB0n, rB1n);
            xpA = xAn; xpB = xBn; rA0 = rA0n; rA1 = rA1n; rB0 = rB0n; rB1 = rB1n;

            const int iA = base + slot, iB = base + PPW + slot;
            store_screen_winner(accA, slot, kk, g * SCREEN_KT, K, iA, iA < n, m1o, m2o, ko);
            store_screen_winner(accB, slot, kk, g * SCREEN_KT, K, iB, iB < n, m1o, m2o, ko);
            if (!more) break;
            t = tn;
            tn = draw();
            base = nbase;
            voxA = nvoxA; voxB = nvoxB; vorA = nvorA; vorB = nvorB;
            xb = nxb; rb = nrb;
        }
    }
}

// Bounds carried from one call to the next (exact acceleration in the manner of Hamerly's k-means, adapted to the
// masked distances): after a screen call every point has
//     ub[i] >= D_a(i)        its TRUE distance to its assigned centroid (from the exact phase-2 value), and
//     lb[i] <= D_k(i)        for every other centroid k (from the certificate: r2 - eps2; 0 if uncertified).
// The masked distance D_k(i) = || x_i - c_k/gamma || over the support S of x_i obeys the triangle inequality in
// R^S, and || (c'_k - c_k)/gamma ||_S <= || (c'_k - c_k)/gamma ||_2 =: delta_k.  So for the next centroids c'
//     D'_a <= ub + delta_a,      D'_k >= lb - max_k delta_k,
// and (ub + delta_a)(1+nu) < (lb - dmax)(1-nu) proves -- without touching the point -- that the reference's argmin
// is unchanged (strictly: no tie).  A 16-point step whose points all pass is skipped by the screen; phase 2 still
// evaluates every point's exact distance to its centroid and the sums, so every output stays exact.
// Buffer layout (floats): ub[npad] | lb[npad] | a[npad] (int32) | delta[K] | dmax | ... | hterm[K] at HB_HTERM
//
// delta_k (rounded up, f32) for all k; delta[K] = max (bit pattern atomicMax: the values are >= 0).
// SUPPORT-AWARE DRIFT (top_s > 0: every point of the shard stores exactly top_s entries).  What the triangle inequality
// needs is || (c'_k - c_k)/gamma ||_S for the point's support S, |S| = s, and for ANY set of s rows that is at most the
// root of the sum of the s LARGEST squared entries of the difference -- a quantity of the centroid alone, about half of
// the full 2-norm for a difference spread over p = 1024 rows at s = 51, and far below it once a centroid only moves in
// a few coordinates.  Exact selection: the s-th largest square is found by a radix search over the high 32 bits of the
// (non-negative) doubles, entries above it are added up, the rest of the s slots are charged the upper end of its
// bucket -- an upper bound that is tight to 2^-20.  top_s <= 0 or >= p (ragged shards, SPKM_NO_SUPPORT_DRIFT): the
// full 2-norm, as before.
// The reference divides each entry by gamma in f64 first: 2^-50 (|c| + |c'|)/gamma covers those roundings.
// hterm[k] = hint_w x (full 2-norm drift)^2: the screen's hints estimate sqrt(ub^2 + hterm) (k_bounds_steps) -- an
// estimate of the typical support, not a bound, hence the full norm scaled by s / p.
// same != nullptr: same[k] = 1 iff centroid k is BITWISE the one of the previous call (the unchanged-cluster shortcut of
// the exact pass, k_cluster_need)
__global__ __launch_bounds__(256) void k_center_drift(const double* __restrict__ prev, const double* __restrict__ cur,
                                                      int K, int p, double gamma, float* __restrict__ delta,
                                                      int* __restrict__ same, int top_s, float hint_w,
                                                      float* __restrict__ hterm)
{
    __shared__ double sh[4][4];
    __shared__ int s_diff;
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_prefix, s_need;
    const int k = blockIdx.x;
    const double g = gamma > 0.0 ? gamma : 1.0;
    double d2 = 0.0, a2 = 0.0, b2 = 0.0;
    bool diff = false;
    if (threadIdx.x == 0) s_diff = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < p; r += blockDim.x) {
        const double a = prev[(size_t)k * p + r], b = cur[(size_t)k * p + r];
        d2 += (b - a) * (b - a);
        a2 += a * a;
        b2 += b * b;
        diff |= __double_as_longlong(a) != __double_as_longlong(b);
    }
    if (diff) s_diff = 1; // (benign race: every writer stores 1)
    for (int off = 32; off > 0; off >>= 1) { d2 += __shfl_down(d2, off); a2 += __shfl_down(a2, off); b2 += __shfl_down(b2, off); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = d2; sh[1][threadIdx.x >> 6] = a2; sh[2][threadIdx.x >> 6] = b2; }
    __syncthreads();
    d2 = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]; // (every thread: the full sums)
    a2 = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    b2 = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
    double sel2 = d2; // sum of the top_s largest squared differences (upper bound); the full sum when not applicable
    if (top_s > 0 && top_s < p && d2 > 0.0 && d2 < __builtin_inf()) { // (NaN / inf: the full norm, which fails every test)
        // radix search, 8 bits at a time from the top, for the high word T of the top_s-th largest square:
        // prefix = bits fixed so far, need = how many of the top_s are still to be found among the entries matching it
        if (threadIdx.x == 0) { s_prefix = 0u; s_need = (unsigned)top_s; }
        for (int shift = 24; shift >= 0; shift -= 8) {
            s_hist[threadIdx.x] = 0u;
            __syncthreads();
            const unsigned prefix = s_prefix;
            const unsigned mask_hi = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
            for (int r = threadIdx.x; r < p; r += blockDim.x) {
                const double e = cur[(size_t)k * p + r] - prev[(size_t)k * p + r];
                const unsigned hi = (unsigned)((unsigned long long)__double_as_longlong(e * e) >> 32);
                if ((hi & mask_hi) == (prefix & mask_hi)) atomicAdd(&s_hist[(hi >> shift) & 255u], 1u);
            }
            __syncthreads();
            {
                // the digit in which the need-th largest entry falls: the largest b whose suffix count (entries with digit
                // >= b) reaches `need`.  Thread b owns bin b; suffix sums by a shuffle scan per wave + the totals of the
                // waves above (one thread walking 256 bins paid an LDS latency per bin: 46 us per call for four passes)
                const unsigned need = s_need;
                const unsigned h = s_hist[threadIdx.x];
                const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
                unsigned suf = h;
                for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_down(suf, off); if (ln + off < 64) suf += t; }
                __shared__ unsigned s_wtot[4];
                if (ln == 0) s_wtot[wv] = suf;
                __syncthreads();
                for (int w = wv + 1; w < 4; w++) suf += s_wtot[w];
                const unsigned suf_next = suf - h; // entries with a larger digit
                if (suf >= need && suf_next < need) { s_prefix = prefix | ((unsigned)threadIdx.x << shift); s_need = need - suf_next; }
            }
            __syncthreads();
        }
        const unsigned T = s_prefix;  // high word of the top_s-th largest square
        const unsigned ties = s_need; // how many entries with exactly this high word belong to the top_s
        double above = 0.0;
        for (int r = threadIdx.x; r < p; r += blockDim.x) {
            const double e = cur[(size_t)k * p + r] - prev[(size_t)k * p + r];
            const double e2 = e * e;
            if ((unsigned)((unsigned long long)__double_as_longlong(e2) >> 32) > T) above += e2;
        }
        for (int off = 32; off > 0; off >>= 1) above += __shfl_down(above, off);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[3][threadIdx.x >> 6] = above;
        __syncthreads();
        above = sh[3][0] + sh[3][1] + sh[3][2] + sh[3][3];
        const double bucket_top = __longlong_as_double((long long)(((unsigned long long)T << 32) | 0xffffffffull));
        sel2 = fmin(d2, above * (1.0 + 1e-12) + (double)ties * bucket_top);
    }
    if (threadIdx.x == 0) {
        const double d = (sqrt(sel2) * (1.0 + 1e-9) + 0x1p-50 * (sqrt(a2) + sqrt(b2))) / g * (1.0 + 1e-9);
        float f = __double2float_ru(d);
        if (!(f >= 0.f)) f = __builtin_inff(); // NaN centres: nothing is skipped
        if (!s_diff) f = 0.f; // bitwise the same centroid: every distance to it is what it was (the rounding allowance above
                              // is for a centroid that CHANGED; a settled cluster's members must not get their bounds bumped
                              // -- and stored -- call after call)
        delta[k] = f;
        atomicMax(reinterpret_cast<unsigned*>(delta + K), __builtin_bit_cast(unsigned, f));
        if (hterm) {
            const float full = (float)(sqrt(d2) / g);
            hterm[k] = hint_w * full * full;
        }
        if (same) same[k] = s_diff ? 0 : 1;
    }
}

// One thread per point, 16-lane groups = the screen's steps.  A step whose points all pass the carried-bounds
// test (k_center_drift's comment) is settled here: assignment = the previous one, lower bound moved by the largest
// drift.  Every other step is appended to todo[] (its index; order within the list does not matter) -- the list
// the screen kernel and k_combine_screen iterate; counters[4] = length, counters[3] = steps skipped.
#define BOUNDS_SPAN 16384   // points per workgroup of k_bounds_steps (1024 steps)
#define BOUNDS_SPAN_PT 4096 // ... when it lists points
// pt_mode: the bounds are applied POINT BY POINT -- a point that passes keeps its assignment (and gets its lower
// bound moved) whatever its 15 neighbours do, the others are listed one by one, and k_screen_quad / k_combine_screen
// run over the listed points only.  With the points of a cluster scattered over the shard (data in arbitrary order)
// a 16-point step is settled only if all 16 pass: 98 % certifiable points leave 28 % of the steps on the screen;
// point by point it is 2 %.  The host selects it from the previous call's counters (both modes count the points that
// passed, counters[12], and the steps whose 16 points all passed, counters[3]): >= 90 % of the points passed and the
// steps left over hold several times as many points as failed (entered at 4x, left below 2.5x).  The screen then fetches 16 B per quad instead of
// 256 B per wave, which only pays while few points are listed.
template <bool MAPPED> // MAPPED: a regrouped shard (map != nullptr) -- a compile-time flag: with the choice made at run time the
// compiler puts the caller-buffer load behind a branch, apart from the other loads of its group (44 -> 113 us per settled call)
__global__ __launch_bounds__(256) void k_bounds_steps(float* __restrict__ bnd, long long npad, long long n, int K,
                                                      int* __restrict__ assign,
                                                      int* __restrict__ todo, unsigned* __restrict__ counters,
                                                      float* __restrict__ hintu, int skip_enabled,
                                                      int pt_mode, const double* __restrict__ cum_in,
                                                      double* __restrict__ cum_out, int span,
                                                      unsigned* __restrict__ blkstat, int erode,
                                                      float* __restrict__ sp_slack = nullptr, unsigned* __restrict__ sp_mask = nullptr,
                                                      int* __restrict__ sp_valid = nullptr, int sp_reset = 0,
                                                      const int* __restrict__ same = nullptr,
                                                      const int* __restrict__ map = nullptr)
{
    // assign == nullptr: the caller's buffer is known to hold the library's copy already (lazy statistics, the same buffer as
    // in the previous call: spkm.h) -- it is neither read nor restored.  map != nullptr (a regrouped shard, api_lloyd.hip): point i of
    // the library's order is the caller's point map[i].
    // BLOCK SUMMARIES (sp_slack != nullptr; lazy calls with K <= 128 only, api_lloyd.hip): per 1024 consecutive points
    //   sp_slack = min_i [ lb_i (1 - 1e-6) - ub_i (1 + 1e-6) ]   (lb as stored: relative to the accumulated drift `cum`)
    //   sp_mask  = the clusters its points belong to (K bits)
    //   sp_valid = every point passed the test of the call that wrote the summary
    // A valid block none of whose clusters moved in this call (same[k], k_center_drift: centroid bitwise unchanged, so
    // delta_a = 0 for every point) passes the per-point test for ALL its points iff cum_now (1 - 1e-6) < sp_slack -- one
    // comparison instead of 16 KB of loads: nothing of the block is read, nothing written, its part of the caller's
    // assignment buffer included (the lazy contract, spkm.h).  A settled run spends most of a call on this test (1.6 GB
    // at N = 1e8); with the points of a cluster stored together nearly every block qualifies once most clusters have stopped.
    // Any other block takes the per-point path below, which rewrites its summary.  sp_reset: every summary is rewritten
    // (the bounds were last written by a call that did not maintain them, or the caller passed another buffer).
    // (Measured and dropped: letting a block whose clusters moved a little pass too, against a per-block lag that its
    //  points' upper bounds take later -- no block more was skipped in iterations 10-40 of the headline run, where nearly
    //  every cluster still exchanges a few points per call and the slack of a block's worst point is small.)
    // erode != 0 (a call whose exact pass will not run: spkm_shard_set_lazy_stats, api_lloyd.hip): a point that passes keeps
    // its centroid but gets no fresh upper bound from anybody, so the bound is moved by its centroid's drift here,
    // ub <- ub + delta_a rounded up (Hamerly's update).  A store only where the centroid moved at all: the members of
    // settled clusters (delta_a = 0) are not written.
    // span: points per list flush (<= BOUNDS_SPAN_PT when points are listed, <= 16 BOUNDS_SPAN_PT for steps; a multiple
    // of 1024) -- the host shortens it on small shards so that every CU still gets several workgroups
    // Lower bounds are stored RELATIVE to the drift accumulated so far: bnd[npad + i] = lb_i + cum at the time lb_i was
    // certified (rounded down), cum = sum over the calls since of the largest centroid drift (each rounded up).  The
    // bound that holds now is the stored value minus today's cum -- exactly the "lower bound moved by the largest drift"
    // of every call in between -- so a point that passes needs NO store: the test reads 12 B per point and writes only
    // the list (a gigabyte of stores costs as much as nine of loads on this part; the eroded bound used to be written
    // back for every settled point in every call).  cum_out <- cum_in + this call's largest drift.
    // hintu != nullptr: every point's ESTIMATE of its distance to its previous centroid under the NEW centroids --
    // the hint of the two-phase screen (k_screen_quad), what the competition's partial sums are compared with.  Not
    // the rigorous ub + delta_a (far too pessimistic: a centroid's move is almost orthogonal to x - c, and only its
    // part on the point's support counts) but sqrt(ub^2 + hterm_a), hterm = (2 s / p) x (full-norm drift)^2
    // (k_center_drift): hints steer work, they prove nothing.  skip_enabled == 0: that is all this kernel does (SPKM_NO_BOUNDS).
    // the steps that stay are collected per workgroup in LDS and appended with ONE global atomic, the skipped ones
    // counted per workgroup (wave-level atomics on one address cost ~8 ms at N = 1e8 when nothing can be skipped)
    __shared__ int s_todo[BOUNDS_SPAN_PT]; // >= BOUNDS_SPAN / 16
    __shared__ unsigned s_cnt, s_pos, s_skip, s_kept;
    const float dmx = bnd[3 * npad + K];
    // rounded UP explicitly: once cum is much larger than dmx the 1e-12 guard on dmx is below the rounding of the addition
    // itself, and cum must stay an upper bound of the total drift.  NaN / inf drift: nothing passes
    // (x + |x| 2^-52 >= the next double above x: one ulp more than the rounded sum can have lost)
    const double cum_rn = *cum_in + (double)dmx * (1.0 + 1e-12);
    const double cum_now = cum_rn + fabs(cum_rn) * 0x1p-52;
    if (skip_enabled && blockIdx.x == 0 && threadIdx.x == 0) *cum_out = cum_now;
    const int lane = threadIdx.x & 63;
    unsigned nskip = 0, nkept = 0;
    if (threadIdx.x == 0) { s_cnt = 0; s_skip = 0; s_kept = 0; }
    __syncthreads();
    constexpr int UN = 4; // rounds whose (dependent) loads are in flight together
    // a workgroup takes several spans: its list is flushed per span (one global atomic, none for a span that lists
    // nothing), its statistics once at the end (at one span per workgroup the 3-4 same-address atomics of 24000
    // workgroups took longer than the test itself)
    __shared__ float s_rmin[4];
    __shared__ unsigned s_moved[4], s_bmask[4];
    const bool sp_on = sp_slack != nullptr && skip_enabled;
    if (sp_on) { // the clusters whose centroid moved in this call, as a K-bit mask
        if (threadIdx.x < 4) { s_moved[threadIdx.x] = 0u; s_bmask[threadIdx.x] = 0u; }
        __syncthreads();
        if ((int)threadIdx.x < K && !same[threadIdx.x]) atomicOr(&s_moved[threadIdx.x >> 5], 1u << (threadIdx.x & 31));
        __syncthreads();
    }
    // per block this workgroup will visit (up to 256 of them: its first spans): passes as a whole.  The summaries are fetched
    // together up front -- a workgroup walks ~100 blocks at N = 1e8, and one dependent fetch per block was 130 us of latency
    // for 2 MB of summaries
    __shared__ int s_pass[256];
    __shared__ unsigned long long s_bits[4];
    auto block_passes = [&](long long b) -> int {
        if ((b << 10) >= npad || !sp_valid[b]) return 0;
        const uint4 mk = *reinterpret_cast<const uint4*>(sp_mask + 4 * b);
        const bool still = ((mk.x & s_moved[0]) | (mk.y & s_moved[1]) | (mk.z & s_moved[2]) | (mk.w & s_moved[3])) == 0u;
        return (still && cum_now * 0.999999 < (double)sp_slack[b]) ? 1 : 0;
    };
    const int nbs = span >> 10; // blocks per span
    if (sp_on && !sp_reset) {
        const long long sp_i = threadIdx.x / nbs;                  // which of the workgroup's spans, which block of it
        const long long sp0 = ((long long)blockIdx.x + sp_i * gridDim.x) * span;
        const int okb = sp0 < npad ? block_passes((sp0 >> 10) + threadIdx.x % nbs) : 0;
        s_pass[threadIdx.x] = okb;
        const unsigned long long bal = __ballot(okb != 0); // ... and as bits, so that a span whose blocks all pass costs one test
        if (lane == 0) s_bits[threadIdx.x >> 6] = bal;
        __syncthreads();
    }
    int span_i = 0;
    for (long long span0 = (long long)blockIdx.x * span; span0 < npad; span0 += (long long)gridDim.x * span, span_i++) {
    if (sp_on && !sp_reset && (span_i + 1) * nbs <= 256 && span0 + span <= n) {
        // every block of this span passes (nbs is a power of two <= 16: a span's bits lie in one word): nothing to read
        const int slot0 = span_i * nbs;
        const unsigned m = (unsigned)(s_bits[slot0 >> 6] >> (slot0 & 63)) & ((1u << nbs) - 1u);
        if (m == (1u << nbs) - 1u) {
            nkept += 256u * (unsigned)nbs;
            nskip += 16u * (unsigned)nbs;
            continue;
        }
    }
    bool span_read = false; // some block of this span took the per-point path (uniform): only then is there a list to flush
    for (int it0 = 0; it0 < span / 256; it0 += UN) {
        if (span0 + it0 * 256 >= npad) break; // npad: whole waves
        const long long blk0 = span0 + (long long)it0 * 256;
        const long long bsp = blk0 >> 10;
        if (sp_on && !sp_reset) { // (the same word for every thread of the workgroup: a uniform branch)
            const int slot = span_i * nbs + (it0 >> 2);
            if (slot < 256 ? s_pass[slot] : block_passes(bsp)) { // every point of the block passes
                if (blk0 + 1024 <= n) { // (a whole block: every wave has 4 x 64 points, 4 x 4 steps of it -- a uniform, scalar test:
                    nkept += 256u;      //  per-lane 64-bit arithmetic here was 70 of the kernel's 100 us at N = 1e8)
                    nskip += 16u;
                } else {
#pragma unroll
                    for (int u = 0; u < UN; u++) {
                        const long long live = n - (blk0 + u * 256 + (long long)(threadIdx.x & ~63)); // points of this wave's 64 that exist
                        nkept += (unsigned)(live <= 0 ? 0 : (live >= 64 ? 64 : live));
                        nskip += (unsigned)(live <= 0 ? 0 : (live >= 64 ? 4 : (live + 15) / 16));
                    }
                }
                continue;
            }
        }
        span_read = true;
        float ubv[UN], lbv[UN], dav[UN];
        int apv[UN], curv[UN];
        bool blk_kept = true;          // sp_on: every point of the block passed
        float tmin = __builtin_inff(); // ... and the smallest slack among them
        unsigned bm0 = 0u, bm1 = 0u, bm2 = 0u, bm3 = 0u;
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long long i = span0 + (it0 + u) * 256 + threadIdx.x;
            const bool in = i < n;
            ubv[u] = in ? bnd[i] : 0.f;
            lbv[u] = in ? bnd[npad + i] : 0.f;
            apv[u] = in ? reinterpret_cast<const int*>(bnd)[2 * npad + i] : 0;
            // what the caller's buffer holds now, fetched with the rest (behind the test it would be a second memory
            // round trip per group): a point that keeps its assignment is stored only if the buffer differs
            if (MAPPED) curv[u] = (in && skip_enabled && assign != nullptr) ? assign[map[i]] : apv[u];
            else curv[u] = (in && skip_enabled) ? assign[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < UN; u++) dav[u] = bnd[3 * npad + apv[u]];
        if (hintu != nullptr && !skip_enabled) { // (with the test on, only the points that stay on the screen get a hint: below)
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const long long i = span0 + (it0 + u) * 256 + threadIdx.x;
                if (i < n) hintu[i] = sqrtf(ubv[u] * ubv[u] + bnd[3 * npad + HB_HTERM + apv[u]]);
            }
        }
        if (!skip_enabled) continue;
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long long i = span0 + (it0 + u) * 256 + threadIdx.x;
            if (span0 + (it0 + u) * 256 >= npad) break;
            const bool keep = !(i < n) || (double)(ubv[u] + dav[u]) * 1.000001 < ((double)lbv[u] - cum_now) * 0.999999; // false for NaN
            const unsigned long long b = __ballot(keep);
            nkept += (unsigned)__popcll(__ballot(keep && i < n));
            const float ubn = (erode && dav[u] > 0.f) ? __double2float_ru((double)ubv[u] + (double)dav[u]) : ubv[u]; // Hamerly's update
            if (erode && keep && i < n && dav[u] > 0.f) bnd[i] = ubn;
            if (sp_on && i < n) {
                blk_kept = blk_kept && keep;
                tmin = fminf(tmin, __double2float_rd((double)lbv[u] * 0.999999 - (double)ubn * 1.000001));
                const unsigned bit = 1u << (apv[u] & 31);
                const int w = apv[u] >> 5;
                bm0 |= w == 0 ? bit : 0u; bm1 |= w == 1 ? bit : 0u; bm2 |= w == 2 ? bit : 0u; bm3 |= w == 3 ? bit : 0u;
            }
            if (pt_mode) {
                if (keep && i < n && curv[u] != apv[u]) assign[MAPPED ? (long long)map[i] : i] = apv[u]; // (see the step mode below)
                if (!keep && hintu != nullptr) hintu[i] = sqrtf(ubv[u] * ubv[u] + bnd[3 * npad + HB_HTERM + apv[u]]);
                const unsigned long long lm = ~b; // (lanes past n count as kept)
                if (lm) {
                    unsigned basepos = 0;
                    if (lane == 0) basepos = atomicAdd(&s_cnt, (unsigned)__popcll(lm));
                    basepos = __builtin_amdgcn_readfirstlane(basepos);
                    if (!keep) s_todo[basepos + __popcll(lm & ((1ull << lane) - 1ull))] = (int)i;
                }
                // (statistics in the same unit as the other mode: steps whose 16 points all passed)
                const bool whole = ((unsigned)(b >> (lane & 48)) & 0xffffu) == 0xffffu;
                nskip += (unsigned)__popcll(__ballot((lane & 15) == 0 && (i - (lane & 15)) < n && whole));
                continue;
            }
            const unsigned grp = (unsigned)(b >> (lane & 48)) & 0xffffu;
            const bool skip = grp == 0xffffu;
            const bool live_step = (i - (lane & 15)) < n; // the step has at least one point
            // a caller that passes the same buffer call after call already holds this value: a 4-B read instead of a
            // 4-B store (a gigabyte of stores costs as much as several of loads here)
            if (skip && i < n && curv[u] != apv[u]) assign[MAPPED ? (long long)map[i] : i] = apv[u];
            if (!skip && i < n && hintu != nullptr) hintu[i] = sqrtf(ubv[u] * ubv[u] + bnd[3 * npad + HB_HTERM + apv[u]]);
            const bool lead = (lane & 15) == 0 && live_step && !skip;
            const unsigned long long lm = __ballot(lead);
            if (lm) {
                unsigned basepos = 0;
                if (lane == 0) basepos = atomicAdd(&s_cnt, (unsigned)__popcll(lm));
                basepos = __builtin_amdgcn_readfirstlane(basepos);
                if (lead) s_todo[basepos + __popcll(lm & ((1ull << lane) - 1ull))] = (int)(i >> 4);
            }
            nskip += (unsigned)__popcll(__ballot((lane & 15) == 0 && live_step && skip));
        }
        if (sp_on) { // the block's new summary (every thread of the workgroup is here: the trip counts are uniform)
            for (int off = 32; off > 0; off >>= 1) {
                tmin = fminf(tmin, __shfl_xor(tmin, off));
                bm0 |= __shfl_xor(bm0, off); bm1 |= __shfl_xor(bm1, off); bm2 |= __shfl_xor(bm2, off); bm3 |= __shfl_xor(bm3, off);
            }
            const int allk = __syncthreads_and(blk_kept ? 1 : 0);
            if (lane == 0) {
                s_rmin[threadIdx.x >> 6] = tmin;
                if (bm0) atomicOr(&s_bmask[0], bm0);
                if (bm1) atomicOr(&s_bmask[1], bm1);
                if (bm2) atomicOr(&s_bmask[2], bm2);
                if (bm3) atomicOr(&s_bmask[3], bm3);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                sp_slack[bsp] = fminf(fminf(s_rmin[0], s_rmin[1]), fminf(s_rmin[2], s_rmin[3]));
                *reinterpret_cast<uint4*>(sp_mask + 4 * bsp) = make_uint4(s_bmask[0], s_bmask[1], s_bmask[2], s_bmask[3]);
                sp_valid[bsp] = allk;
                s_bmask[0] = s_bmask[1] = s_bmask[2] = s_bmask[3] = 0u;
            }
            __syncthreads();
        }
    }
    if (span_read) { // (a span whose blocks all passed as blocks listed nothing: four barriers saved, ~100 spans per workgroup)
    __syncthreads();
    if (threadIdx.x == 0) s_pos = s_cnt ? atomicAdd(counters + 4, s_cnt) : 0u;
    __syncthreads();
    for (unsigned j = threadIdx.x; j < s_cnt; j += 256) todo[s_pos + j] = s_todo[j];
    __syncthreads();
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    }
    }
    if (lane == 0 && nskip) atomicAdd(&s_skip, nskip);
    if (lane == 0 && nkept) atomicAdd(&s_kept, nkept);
    __syncthreads();
    if (threadIdx.x == 0) { // points that passed the test (either mode), steps whose 16 points all passed: k_call_tail adds them up
        blkstat[2 * blockIdx.x] = s_kept;
        blkstat[2 * blockIdx.x + 1] = s_skip;
    }
}

// Unchanged-cluster shortcut of the exact pass.  Cluster k needs no work in this call when (i) its centroid is bitwise
// the one the previous call was given (same[k], k_center_drift) and (ii) no point left or entered it (touched[k] == 0,
// k_combine_screen / k_assign_list): every member's distance to it, the library's upper bounds, the per-cluster sums and counts, obj2
// and the largest distance are then exactly what the previous call produced, and they are reused (k_cluster_restore,
// k_cluster_stats) instead of streamed again.  force != 0: everything is processed (first call of a shard, changed K or
// gamma, distances requested).  need[k] = 1: process.  counters[32..33]: running total of the points processed.
// The sums and counts of a cluster depend on its MEMBERS only: they are taken from the cache whenever no point left or
// entered (touched[k] == 0), also when the cluster is streamed again because its centroid moved -- what the pass adds up
// for it then is discarded (k_cluster_restore).  That is what lets a settled cluster's centroid become bitwise stable in
// the first place: sums recomputed by atomics in no fixed order would differ in the last bit from call to call, and so
// would the centroid.  On return touched[k] = 1 marks the clusters whose freshly accumulated sums are the ones to keep.
__global__ void k_cluster_need(int* __restrict__ touched, const int* __restrict__ same, int force, int K,
                               const unsigned long long* __restrict__ nk, int* __restrict__ need,
                               unsigned* __restrict__ counters, const unsigned* __restrict__ gate = nullptr)
{
    if (gate != nullptr && *gate == 0u) return; // (k_pick_form: this call's sums come from the events)
    __shared__ unsigned long long s_pts;
    if (threadIdx.x == 0) s_pts = 0ull;
    __syncthreads();
    unsigned long long mine = 0ull;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const int fresh = (force || touched[k]) ? 1 : 0;
        const int nd = (fresh || !same[k]) ? 1 : 0;
        need[k] = nd;
        touched[k] = fresh;
        if (nd) mine += nk[k];
    }
    if (mine) atomicAdd(&s_pts, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(counters + 32), s_pts);
        counters[13] = (unsigned)(s_pts > 0xffffffffull ? 0xffffffffull : s_pts); // points the exact pass processes in this call
    }
}

// after the exact pass: clusters whose membership changed (fresh[k], k_cluster_need) refresh the cache with their LOCAL
// sums / counts (before any all-reduce), all others get theirs from it
__global__ __launch_bounds__(256) void k_cluster_restore(const int* __restrict__ need, int K, int p,
                                                         double* __restrict__ sums, double* __restrict__ counts,
                                                         double* __restrict__ cache_s, double* __restrict__ cache_c,
                                                         const unsigned* __restrict__ gate = nullptr)
{
    if (gate != nullptr && *gate == 0u) return; // (k_pick_form: this call's sums come from the events)
    const size_t pk = (size_t)K * p;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < pk; t += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(t / p);
        if (need[k]) { cache_s[t] = sums[t]; cache_c[t] = counts[t]; }
        else { sums[t] = cache_s[t]; counts[t] = cache_c[t]; }
    }
}

// per-cluster statistics from the per-item ones of the processed clusters (items of a cluster are consecutive:
// ibeg[k], icnt[k] from k_plan_segments), cached for the others; then stats = { sum obj2, max distance, its first index }
__global__ __launch_bounds__(256) void k_cluster_stats(const int* __restrict__ need, int K, const int* __restrict__ ibeg,
                                                       const int* __restrict__ icnt, const double* __restrict__ it_obj,
                                                       const double* __restrict__ it_max,
                                                       const long long* __restrict__ it_imax, double* __restrict__ cl_obj,
                                                       double* __restrict__ cl_max, long long* __restrict__ cl_imax,
                                                       double* __restrict__ stats)
{
    __shared__ double s_o[256], s_m[256];
    __shared__ long long s_i[256];
    const int tid = threadIdx.x;
    double o = 0.0, m = -1.0;
    long long im = 0x7fffffffffffffffLL;
    for (int k = tid; k < K; k += 256) {
        if (need[k]) {
            double ko = 0.0, km = -1.0;
            long long ki = 0x7fffffffffffffffLL;
            for (int t = ibeg[k]; t < ibeg[k] + icnt[k]; t++) {
                ko += it_obj[t];
                if (it_max[t] > km || (it_max[t] == km && it_imax[t] < ki)) { km = it_max[t]; ki = it_imax[t]; }
            }
            cl_obj[k] = ko; cl_max[k] = km; cl_imax[k] = ki;
        }
        o += cl_obj[k];
        if (cl_max[k] > m || (cl_max[k] == m && cl_imax[k] < im)) { m = cl_max[k]; im = cl_imax[k]; }
    }
    s_o[tid] = o; s_m[tid] = m; s_i[tid] = im;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) {
            s_o[tid] += s_o[tid + off];
            if (s_m[tid + off] > s_m[tid] || (s_m[tid + off] == s_m[tid] && s_i[tid + off] < s_i[tid])) {
                s_m[tid] = s_m[tid + off];
                s_i[tid] = s_i[tid + off];
            }
        }
        __syncthreads();
    }
    if (tid == 0) { stats[0] = s_o[0]; stats[1] = s_m[0]; stats[2] = (double)s_i[0]; }
}
// The accumulation form of a lazy call, chosen ON THE DEVICE (api_lloyd.hip, `dual`): a call whose mover count the host cannot
// know yet -- a run's second call; the counters come back one call late -- queues BOTH forms, the incremental one
// (k_plan_segments / k_scatter_by_cluster / k_accumulate_events over the events) and the full sums-only pass
// (k_cluster_need / plan / scatter / k_exact_accumulate_rec<DIST = false> / k_cluster_restore over all points), and this
// kernel, queued behind k_assign_list -- when every event has been counted, counters[16] -- opens exactly one of them:
//   counters[18] = 1: the events (at most ev_cap of them: movers <= n / 3, policy.h few_movers)
//   counters[19] = 1: the full pass
// The kernels of the other form return at once (their `gate`) or find an empty work list: nitems[0] (full pass) and
// nitems[1] (events) are zeroed here, and only the plan kernel that runs fills its own.
__global__ void k_pick_form(unsigned* __restrict__ counters, unsigned ev_cap, int* __restrict__ nitems)
{
    const bool full = counters[16] > ev_cap;
    counters[18] = full ? 0u : 1u;
    counters[19] = full ? 1u : 0u;
    nitems[0] = 0;
    nitems[1] = 0;
    nitems[2] = 0; // (pair events: the first-level plan's chunk list)
}

// REGROUPING a shard whose 16-point steps mix clusters (data in arbitrary order; api_lloyd.hip, regroup_shard): the library's own
// order of the points -- the order of the screen copy and of everything it keeps per point -- becomes "by cluster, and inside
// a cluster the points that are sure of it first", so that a step's 16 points share their centroid and their prospects: a
// step is skipped on the carried bounds, or finished early by the hinted form, only if all 16 points allow it.  The records
// (and with them every per-point array of the caller) stay where they are; map[i] = the caller's index of the library's
// point i.  kmeans_sparsified.m:430-431 adds up find(assignments == k): no order of the points enters any result.
//   key = 2 * cluster + (1 if the runner-up bound is within 1.5x of the upper bound: the point may well move)
__global__ __launch_bounds__(256) void k_regroup_keys(const float* __restrict__ bnd, long long npad, long long n, int K,
                                                      const double* __restrict__ cum, int* __restrict__ keys)
{
    const double cum_now = *cum;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float ub = bnd[i];
        const double lb = (double)bnd[npad + i] - cum_now;
        int a = reinterpret_cast<const int*>(bnd)[2 * npad + i];
        if ((unsigned)a >= (unsigned)K) a = 0;
        keys[i] = 2 * a + ((lb >= 1.5 * (double)ub) ? 0 : 1); // (NaN: unsure)
    }
}
// the per-point state in the new order: new point j was point perm[j]; newmap[j] = oldmap[perm[j]] (oldmap null: identity)
__global__ __launch_bounds__(256) void k_regroup_apply(const int* __restrict__ perm, const int* __restrict__ oldmap,
                                                       int* __restrict__ newmap, const float* __restrict__ bnd_old,
                                                       float* __restrict__ bnd_new, long long npad, long long n)
{
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < npad; j += (long long)gridDim.x * blockDim.x) {
        if (j < n) {
            const int o = perm[j];
            newmap[j] = oldmap != nullptr ? oldmap[o] : o;
            bnd_new[j] = bnd_old[o];
            bnd_new[npad + j] = bnd_old[npad + o];
            bnd_new[2 * npad + j] = bnd_old[2 * npad + o];
        } else {
            bnd_new[j] = 0.f; bnd_new[npad + j] = 0.f; bnd_new[2 * npad + j] = 0.f;
        }
    }
}

// jc of a fixed-stride shard that was created from records: jc[i] = i * s
__global__ void k_fill_jc(long long* __restrict__ jc, long long n, long long s)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (long long)gridDim.x * blockDim.x) jc[i] = i * s;
}

// several small buffers zeroed by ONE launch (the per-call counters, flags and the reduce buffer of the fused
// iteration: six memsets were six launches in front of the first kernel that does work)
struct spkm_zero_jobs {
    unsigned* p[8];
    unsigned long long words[8]; // 4-byte words
    int n;
};
__global__ __launch_bounds__(256) void k_zero_many(const spkm_zero_jobs j)
{
    for (int q = 0; q < j.n; q++) {
        unsigned* d = j.p[q];
        for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < j.words[q];
             i += (unsigned long long)gridDim.x * blockDim.x)
            d[i] = 0u;
    }
}
__global__ void k_zero_u64_gated(unsigned long long* __restrict__ dst, int n, const unsigned* __restrict__ gate)
{
    if (gate != nullptr && *gate == 0u) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = 0ull;
}

// Per point: best / second-best estimate over the G tiles, certification, candidate assignment.
// Uncertified points are appended to list[] (count in *nlist); a tile that reports "no candidate"
// (+inf, +inf, -1) can never certify.
__global__ __launch_bounds__(256) void k_combine_screen(const float* __restrict__ scr_m1,
                                                        const float* __restrict__ scr_m2,
                                                        const int* __restrict__ scr_k, long long n, int G,
                                                        const float* __restrict__ xnr, int fixed_s,
                                                        const unsigned long long* __restrict__ cmax_bits,
                                                        int* __restrict__ assign, int* __restrict__ list,
                                                        unsigned int* __restrict__ nlist,
                                                        float* __restrict__ bnd, long long npad, int skipping,
                                                        const int* __restrict__ todo, int pt_mode,
                                                        const double* __restrict__ cum, int lib_valid,
                                                        int* __restrict__ touched, int K,
                                                        unsigned long long* __restrict__ nk,
                                                        int lazy, int* __restrict__ ev_pt, int* __restrict__ ev_k,
                                                        unsigned long long* __restrict__ nk_ev, unsigned ev_cap,
                                                        unsigned* __restrict__ wgstat, int* __restrict__ ev_o = nullptr,
                                                        const int* __restrict__ map = nullptr, int trusted = 0)
{
    // map != nullptr (a regrouped shard, api_lloyd.hip): point i of the library's order is the caller's point map[i] -- what is
    // written to the caller's assignment buffer and into the events (the records are in the caller's order) goes through it;
    // trusted: the caller's buffer already holds the library's copy (lazy statistics, same buffer as last call), so only
    // CHANGES are stored -- through a map every store is a scattered 4-byte write.
    // wgstat[4 b + 3] (calls over all points): 16-point steps whose points share one cluster -> nlist[21]; the host regroups a
    // shard whose steps are mixed (data in arbitrary order).
    // ev_o != nullptr: PAIR events -- ONE event per mover, (point, new cluster in ev_k, old cluster or -1 in ev_o), and the
    // histogram nk_ev over the K new clusters only: the events are then sorted by (new, old) pair and every mover's record
    // is read once, into its new cluster's sums and out of its old one's (api_lloyd.hip, k_accumulate_events<.., PAIR>); else
    // two events per mover, (point, K + old) and (point, new), over 2 K keys, each applied on its own.
    // wgstat[3 b .. 3 b + 2]: workgroup b's ambiguous points, "some assignment changed" flag and movers, as plain stores;
    // k_assign_list (the next launch) adds them up into nlist[1], nlist[5], nlist[14].  One atomic per workgroup and counter
    // on ONE cache line is served at ~10 ns apiece: 3500 workgroups of a settled call's point list (N = 1e8) spent 80 of
    // this kernel's 130 us queueing for them.
    // ev_cap: events appended at a position beyond it are counted but NOT stored -- the call's accumulation form is
    // chosen on the device from the count (k_pick_form: more than ev_cap events -> the full pass, which needs none of
    // them), so once the running count has passed the cap the stores (16 B per mover) would be wasted
    // nk_ev (with ev_pt): the histogram of the events over their 2 K keys, collected per workgroup next to the cluster-size
    // deltas -- the counting sort of the events then needs no histogram pass of its own
    // lazy != 0 (the exact pass will not run in this call, api_lloyd.hip): a certified point's upper bound is written here,
    // (r1 + eps1) rounded up -- rigorous, if a few 1e-6 looser than the exact distance the pass would have stored -- and
    // every point that changes cluster is recorded as two EVENTS, (point, K + old cluster) and (point, new cluster), for
    // the incremental update of the per-cluster sums (k_accumulate_events).  Events are staged in LDS and appended with
    // one global atomic per ~1500 (half the points move in a run's first iterations: an atomic per wave on one address
    // would take tens of milliseconds).  nlist[14] counts the movers in every mode, nlist[16] the events.
    constexpr int EVCAP = 2048;
    __shared__ int s_evp[EVCAP], s_evk[EVCAP];
    const bool pair = ev_o != nullptr; // (staged as new | (old + 1) << 16 in s_evk: pair events are for K <= 128)
    __shared__ unsigned s_evn, s_evbase, s_mov, s_over; // s_over: the running event count has passed ev_cap -- this workgroup stops collecting events
    const double cum_now = cum ? *cum : 0.0; // lower bounds are stored relative to the accumulated drift (k_bounds_steps)
    // The library's own copy of the assignment (bnd + 2 npad) is kept up to date here and in k_assign_list -- the only
    // two places an assignment can change: a CERTIFIED point's new cluster is written at once; an uncertified one keeps
    // the previous call's value until k_assign_list has the exact answer.  lib_valid: the copy holds the previous call's
    // final assignment, so that a change is known on the spot --
    //   nlist[5]: counts (per workgroup) that some assignment changed: the gate of the counting-sort reuse;
    //   touched[k] = 1 for every cluster a point left or entered (the unchanged-cluster shortcut, k_cluster_need);
    //   nk[k]: the cluster sizes, moved by the points that changed (collected per workgroup in K ints of dynamic LDS:
    //   half the points move in the first iterations of a run, and global atomics on K addresses would take 100 ms).
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    int* delta = reinterpret_cast<int*>(smem_c);
    unsigned* evc = reinterpret_cast<unsigned*>(smem_c) + (nk ? K : 0); // 2 K event counters (ev_pt != nullptr)
    int* alib = bnd ? reinterpret_cast<int*>(bnd + 2 * npad) : nullptr;
    bool changed = false;
    // bnd != nullptr: write each point's new lower bound (k_center_drift's comment)
    float* lbv = bnd ? bnd + npad : nullptr;
    const double cmax = __builtin_bit_cast(double, *cmax_bits);
    const double u = 0x1p-24;
    const double eu = (2.0 * u + u * u) * (1.0 + 1e-9);
    const double gacc = (double)(fixed_s + 1) * u * (1.0 + 1e-4);
    const double sqrt_s = sqrt((double)fixed_s) * (1.0 + 1e-15);
    const double nu = 0x1p-45;
    unsigned nambig = 0;
    // skipping: k_bounds_steps has settled the skipped steps; only the listed ones (nlist[4] of them) are looked at
    const long long total = skipping ? (pt_mode ? (long long)nlist[4] : (long long)nlist[4] * 16) : n;
    __shared__ unsigned s_amb, s_chg, s_hom;
    if ((long long)blockIdx.x * blockDim.x >= total) { // (whole workgroup)
        if (threadIdx.x == 0) { wgstat[4 * blockIdx.x] = 0u; wgstat[4 * blockIdx.x + 1] = 0u; wgstat[4 * blockIdx.x + 2] = 0u; wgstat[4 * blockIdx.x + 3] = 0u; }
        return;
    }
    if (threadIdx.x == 0) { s_amb = 0u; s_chg = 0u; s_evn = 0u; s_mov = 0u; s_over = 0u; s_hom = 0u; }
    if (nk) for (int k = threadIdx.x; k < K; k += blockDim.x) delta[k] = 0;
    if (ev_pt) for (int k = threadIdx.x; k < (pair ? K : 2 * K); k += blockDim.x) evc[k] = 0u;
    __syncthreads();
    unsigned nmov = 0, nhomog = 0;
    auto flush_events = [&]() { // (whole workgroup)
        if (threadIdx.x == 0) { s_evbase = atomicAdd(nlist + 16, s_evn); if (s_evbase > ev_cap) s_over = 1u; }
        __syncthreads();
        if (s_evbase <= ev_cap)
            for (unsigned j = threadIdx.x; j < s_evn; j += blockDim.x) {
                ev_pt[s_evbase + j] = s_evp[j];
                const int kv = s_evk[j];
                ev_k[s_evbase + j] = pair ? (kv & 0xffff) : kv;
                if (pair) ev_o[s_evbase + j] = (kv >> 16) - 1;
            }
        __syncthreads();
        if (threadIdx.x == 0) s_evn = 0u;
        __syncthreads();
    };
    for (long long q0 = (long long)blockIdx.x * blockDim.x; q0 < total; q0 += (long long)gridDim.x * blockDim.x) {
      const long long q = q0 + threadIdx.x;
      bool mover = false;
      int mv_old = -1, mv_new = 0, mine_k = -1;
      long long i = n;
      if (q < total) i = skipping ? (pt_mode ? (long long)todo[q] : (long long)todo[q >> 4] * 16 + (q & 15)) : q;
      if (i < n) {
        float b1 = __builtin_inff(), b2 = __builtin_inff();
        int bk = -1;
        for (int g = 0; g < G; g++) {
            const size_t at = (size_t)g * n + (size_t)((skipping && pt_mode) ? q : i); // (point lists: stored by list slot)
            const float m1 = scr_m1[at], m2 = scr_m2[at];
            const int k = scr_k[at];
            // m1 = estimate of the tile's leader, m2 = a LOWER bound of every other centroid of the tile (the
            // second smallest estimate, or with the two-phase screen the second smallest partial sum -- which
            // can lie below m1): every m2 and every m1 but the best one bound the competition from below
            if (m1 < b1) { b2 = fminf(b2, b1); b1 = m1; bk = k; }
            else { b2 = fminf(b2, m1); }
            b2 = fminf(b2, m2);
        }
        const double E = eu * ((double)xnr[i] + sqrt_s * cmax) * (1.0 + 1e-9) * (1.0 + 1e-9); // eu sqrt(W), W as in the header

        const double r1 = sqrt((double)b1), r2 = sqrt((double)b2);
        const double e1 = E + gacc * r1 + 1e-20, e2 = E + gacc * r2 + 1e-20;
        const bool certified = (bk >= 0) && ((r1 + e1) * (1.0 + nu) < (r2 - e2) * (1.0 - nu));
        const int newk = bk >= 0 ? bk : 0;
        const long long ic = map != nullptr ? (long long)map[i] : i; // the caller's index of this point
        const int old = (alib && lib_valid) ? alib[i] : -1;
        if (!(trusted && lib_valid) || old != newk) assign[ic] = newk; // (the caller's buffer; tentative for an uncertified point)
        mine_k = newk;
        if (alib && certified) {
            if (old != newk) {
                alib[i] = newk;
                if (lib_valid) {
                    changed = true;
                    const bool vo = (unsigned)old < (unsigned)K;
                    if (touched) { if (vo) touched[old] = 1; touched[newk] = 1; }
                    if (nk) { if (vo) atomicAdd(&delta[old], -1); atomicAdd(&delta[newk], 1); }
                    mover = true; mv_old = vo ? old : -1; mv_new = newk;
                    nmov++;
                }
            }
        }
        if (lazy && certified && bnd) bnd[i] = __double2float_ru((r1 + e1) * (1.0 + nu) * (1.0 + 1e-12));
        if (lbv) lbv[i] = __double2float_rd((certified ? fmax(0.0, (r2 - e2) * (1.0 - nu)) : 0.0) + cum_now);
        if (!certified) {
            // one atomic per wave, not per point: atomics on one address are served one after the other (~12 ns each --
            // 30 000 uncertified points of a cold iteration cost more than the rest of this kernel)
            const unsigned long long um = __ballot(1);
            const int ln = threadIdx.x & 63;
            unsigned at = 0;
            if (ln == __builtin_ctzll(um)) at = atomicAdd(nlist, (unsigned)__popcll(um));
            at = (unsigned)__builtin_amdgcn_readlane((int)at, __builtin_ctzll(um));
            list[at + __popcll(um & ((1ull << ln) - 1ull))] = (int)i;
        }
        // nlist[1]: points whose runner-up is within 2.25x of the winner -- the ones a partial-sum lower bound (a
        // quarter of the rounds: 5x in the squares leaves a margin) 
        // could not separate; the host decides from this count whether the next call may use the two-phase screen
        if (!(r2 >= 2.25 * r1)) nambig++;
      }
      if (!skipping) { // (q = i: the wave's lanes are four whole 16-point steps)
          const int kf = __shfl(mine_k, threadIdx.x & 48);
          const unsigned long long eq = __ballot(mine_k == kf && i < n);
          if ((threadIdx.x & 15) == 0 && i < n && (unsigned)((eq >> (threadIdx.x & 48)) & 0xffffull) == 0xffffu) nhomog++;
      }
      // (s_over: the full pass will run whatever else moves, k_pick_form -- no event of this workgroup is needed any more;
      //  the flag changes only inside flush_events, between barriers: the same for every thread of a trip)
      if (ev_pt && !s_over && __syncthreads_or(mover ? 1 : 0)) { // (every thread of the workgroup gets here in every trip; no mover: nothing to stage)
        const int cnt = mover ? ((mv_old >= 0 && !pair) ? 2 : 1) : 0;
        // exclusive prefix of cnt inside the wave, one LDS atomic per wave for its total
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if ((int)(threadIdx.x & 63) >= off) incl += t; }
        const int wtot = __shfl(incl, 63);
        unsigned wbase = 0;
        if (wtot) {
            if ((threadIdx.x & 63) == 0) wbase = atomicAdd(&s_evn, (unsigned)wtot);
            wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)wbase);
        }
        if (mover) {
            unsigned at = wbase + (unsigned)(incl - cnt);
            const int ie = map != nullptr ? map[i] : (int)i; // (the records the events are applied from are in the caller's order)
            if (!pair && mv_old >= 0) { s_evp[at] = ie; s_evk[at] = K + mv_old; at++; atomicAdd(&evc[K + mv_old], 1u); }
            s_evp[at] = ie; s_evk[at] = pair ? (mv_new | ((mv_old + 1) << 16)) : mv_new;
            atomicAdd(&evc[mv_new], 1u);
        }
        __syncthreads();
        if (s_evn + 2u * 256u > (unsigned)EVCAP) flush_events();
      }
    }
    if (ev_pt) { __syncthreads(); if (s_evn) flush_events(); }
    for (int off = 32; off > 0; off >>= 1) nmov += __shfl_down(nmov, off);
    if ((threadIdx.x & 63) == 0 && nmov) atomicAdd(&s_mov, nmov);
    // per workgroup: with a short list nearly every wave holds an ambiguous point (the listed points ARE the ones near a
    // boundary), and one atomic per wave on one address took longer than the rest of the kernel
    for (int off = 32; off > 0; off >>= 1) nambig += __shfl_down(nambig, off);
    if ((threadIdx.x & 63) == 0 && nambig) atomicAdd(&s_amb, nambig);
    if (__any(changed) && (threadIdx.x & 63) == 0) s_chg = 1u;
    for (int off = 32; off > 0; off >>= 1) nhomog += __shfl_down(nhomog, off);
    if ((threadIdx.x & 63) == 0 && nhomog) atomicAdd(&s_hom, nhomog);
    __syncthreads();
    if (threadIdx.x == 0) {
        wgstat[4 * blockIdx.x] = s_amb;
        wgstat[4 * blockIdx.x + 1] = s_chg;
        wgstat[4 * blockIdx.x + 2] = s_mov;
        wgstat[4 * blockIdx.x + 3] = s_hom;
    }
    if (nk)
        for (int k = threadIdx.x; k < K; k += blockDim.x)
            if (delta[k]) atomicAdd(&nk[k], (unsigned long long)(long long)delta[k]);
    if (ev_pt)
        for (int k = threadIdx.x; k < (pair ? K : 2 * K); k += blockDim.x)
            if (evc[k]) atomicAdd(&nk_ev[k], (unsigned long long)evc[k]);
}

// Listed points: exact reference arithmetic over all K centroids (row-major scaled centres Cs in
// global memory), sqrt, first-index min.  One wave per point.
template <typename IR>
__global__ __launch_bounds__(256) void k_assign_list(const long long* __restrict__ jc, const IR* __restrict__ ir,
                                                     const double* __restrict__ xval, const double* __restrict__ Cs,
                                                     int K, int fixed_s, const int* __restrict__ list,
                                                     const unsigned int* __restrict__ nlist,
                                                     int* __restrict__ assign, int* __restrict__ alib, int lib_valid,
                                                     unsigned* __restrict__ changed, int* __restrict__ touched,
                                                     unsigned long long* __restrict__ nk,
                                                     float* __restrict__ ubv, int* __restrict__ ev_pt,
                                                     int* __restrict__ ev_k, unsigned* __restrict__ counters,
                                                     const char* __restrict__ rec, int rec_R,
                                                     unsigned long long* __restrict__ nk_ev, unsigned ev_cap,
                                                     const unsigned* __restrict__ wgstat, int nwg,
                                                     int* __restrict__ ev_o = nullptr, const int* __restrict__ map = nullptr)
{
    // map != nullptr (a regrouped shard): listed point i of the library's order is the caller's point map[i] -- its record,
    // its place in the caller's assignment buffer, its id in the events
    // ev_o != nullptr: pair events, one per mover (k_combine_screen)
    // wgstat / nwg: k_combine_screen's per-workgroup statistics (ambiguous points, changed flag, movers); the LAST
    // workgroup of this launch adds them into counters[1], *changed (counters[5]) and counters[14] -- nobody reads those
    // before this kernel has finished
    if (blockIdx.x == gridDim.x - 1) {
        unsigned a = 0u, c = 0u, m = 0u, h = 0u;
        for (int b = threadIdx.x; b < nwg; b += blockDim.x) { a += wgstat[4 * b]; c += wgstat[4 * b + 1]; m += wgstat[4 * b + 2]; h += wgstat[4 * b + 3]; }
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); c += __shfl_down(c, off); m += __shfl_down(m, off); h += __shfl_down(h, off); }
        if ((threadIdx.x & 63) == 0) {
            if (a) atomicAdd(counters + 1, a);
            if (c) atomicAdd(changed, c);
            if (m) atomicAdd(counters + 14, m);
            if (h) atomicAdd(counters + 21, h);
        }
    }
    // alib / lib_valid / touched / nk: the library's copy of the assignment and what follows from a change, as in
    // k_combine_screen (these points kept their previous value there); few points: global atomics
    // ubv != nullptr (lazy calls): the point's upper bound = its exact distance, rounded up; ev_pt / ev_k: the two
    // events of a point that changes cluster (k_combine_screen); counters[14] movers, counters[16] events
    // rec != nullptr: the point's entries come from the record layout (x | ir side by side, R bytes per point) -- the
    // only copy of the exact entries once the shard's CSC arrays have been released (spkm_shard_release_csc)
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long cnt = *nlist;
    for (long long q = wave; q < cnt; q += nwaves) {
        const long long i = list[q];
        const long long ic = map != nullptr ? (long long)map[i] : i;
        const double* xq;
        const IR* rq;
        int ne;
        if (rec != nullptr) {
            const char* b = rec + (size_t)ic * (size_t)rec_R;
            xq = reinterpret_cast<const double*>(b);
            rq = reinterpret_cast<const IR*>(b + (size_t)fixed_s * 8);
            ne = fixed_s;
        } else {
            const long long j0 = fixed_s > 0 ? ic * fixed_s : jc[ic];
            xq = xval + j0;
            rq = ir + j0;
            ne = fixed_s > 0 ? fixed_s : (int)(jc[ic + 1] - j0);
        }
        double best = __builtin_inf();
        int bk = 0x7fffffff;
        // two centroids per lane at a time (k and k + 64): their gathers are in flight together -- with K = 100 a wave used to
        // walk the column twice, one dependent batch of loads after the other (a short list is all latency); each sum keeps
        // its own storage-order additions, and the candidates are compared in ascending k
        for (int k = lane; k < K; k += 128) {
            const bool two = k + 64 < K;
            const int kb = two ? k + 64 : k;
            double acc = 0.0, accb = 0.0;
            int j = 0;
            for (; j + 8 <= ne; j += 8) { // eight (sixteen) independent gathers in flight; the additions stay in storage order
                double c[8], d[8], cb[8], db[8];
                int r[8];
#pragma unroll
                for (int u = 0; u < 8; u++) r[u] = (int)rq[j + u];
#pragma unroll
                for (int u = 0; u < 8; u++) { c[u] = Cs[(size_t)r[u] * K + k]; cb[u] = Cs[(size_t)r[u] * K + kb]; }
#pragma unroll
                for (int u = 0; u < 8; u++) { const double xv = xq[j + u]; d[u] = xv - c[u]; db[u] = xv - cb[u]; }
#pragma unroll
                for (int u = 0; u < 8; u++) { acc = acc + d[u] * d[u]; accb = accb + db[u] * db[u]; }
            }
            for (; j < ne; j++) {
                const double xv = xq[j];
                const size_t r = (size_t)rq[j] * K;
                const double d = xv - Cs[r + k], db = xv - Cs[r + kb];
                acc = acc + d * d;
                accb = accb + db * db;
            }
            const double dd = sqrt(acc), ddb = sqrt(accb);
            if (dd < best) { best = dd; bk = k; }
            if (two && ddb < best) { best = ddb; bk = kb; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off);
            const int ok = __shfl_xor(bk, off);
            if (ob < best || (ob == best && ok < bk)) { best = ob; bk = ok; }
        }
        if (lane == 0) {
            if ((unsigned)bk >= (unsigned)K) bk = 0; // non-finite distances: MATLAB's min() gives index 1; never an out-of-range cluster
            assign[ic] = bk;
            if (ubv) ubv[i] = __double2float_ru(best * (1.0 + 1e-12)); // (inf / NaN: the point fails every later test)
            if (alib) {
                const int old = lib_valid ? alib[i] : -1;
                if (old != bk) {
                    alib[i] = bk;
                    if (lib_valid) {
                        atomicAdd(changed, 1u);
                        atomicAdd(counters + 14, 1u);
                        const bool vo = (unsigned)old < (unsigned)K;
                        if (touched) { if (vo) touched[old] = 1; touched[bk] = 1; }
                        if (nk) { if (vo) atomicAdd(&nk[old], ~0ull); atomicAdd(&nk[bk], 1ull); }
                        if (ev_pt && ev_o != nullptr) {
                            const unsigned at = atomicAdd(counters + 16, 1u);
                            if (at <= ev_cap) { ev_pt[at] = (int)ic; ev_k[at] = bk; ev_o[at] = vo ? old : -1; }
                            atomicAdd(&nk_ev[bk], 1ull);
                        } else if (ev_pt) {
                            const unsigned at = atomicAdd(counters + 16, vo ? 2u : 1u);
                            if (at <= ev_cap) { // (k_combine_screen: past the cap the events are only counted)
                                if (vo) { ev_pt[at] = (int)ic; ev_k[at] = K + old; }
                                ev_pt[at + (vo ? 1u : 0u)] = (int)ic; ev_k[at + (vo ? 1u : 0u)] = bk;
                            }
                            if (vo) atomicAdd(&nk_ev[K + old], 1ull);
                            atomicAdd(&nk_ev[bk], 1ull);
                        }
                    }
                }
            }
        }
    }
}

// gate != nullptr: the kernel does nothing when *gate == 0 (no assignment changed since the call whose counting sort
// is still in the context's buffers -- spkm_assign_accumulate_dev)
// n_dev != nullptr: the number of entries is *n_dev (an event list whose length only the device knows)
__global__ __launch_bounds__(256) void k_hist(const int* __restrict__ assign, long long n, int K,
                                              unsigned long long* __restrict__ nk, const unsigned* __restrict__ gate,
                                              const unsigned* __restrict__ n_dev = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate != nullptr && *gate == 0u) return;
    if (n_dev != nullptr) n = (long long)*n_dev;
    unsigned int* hist = reinterpret_cast<unsigned int*>(smem);
    for (int k = threadIdx.x; k < K; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = assign[i];
        const unsigned long long act = __ballot(1);
        const int k0 = __builtin_amdgcn_readfirstlane(k);
        if (__ballot(k == k0) == act) { // the whole wave holds one cluster: one atomic for all lanes
            if (__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u)) == 0)
                atomicAdd(&hist[k0], (unsigned)__builtin_popcountll(act));
        } else
            atomicAdd(&hist[k], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x)
        if (hist[k]) atomicAdd(&nk[k], (unsigned long long)hist[k]);
}

// Phase 2: per (cluster k, segment) work item -- the segment's points are all assigned to k.
//  (a) lanes <-> entries (coalesced): m_j = RN(RN(x_j - c_k[r_j])^2) against the f64 column -c_k/gamma held in
//      LDS -- the reference's separately rounded subtract and multiply -- is staged in LDS, and the
//      per-cluster sums / counts are updated with LDS atomics (as k_accumulate_sorted does);
//  (b) lane q then adds point q's staged m_j IN STORAGE ORDER (the reference's accumulation order):
//      mind[i] = sqrt(acc) is exactly the value the reference's min() returns for this point.
// LDS: negc f64[p] | ssum f64[p] | scnt u32[p] | per wave: ms f64[PTS*S1]   (S1 = s|1: odd stride, conflict-free)
// Also per-block partial statistics: sum mind^2, max mind and its first index.
// Record layout (optional, `rec` != nullptr): the library's private copy of a fixed-stride shard with each point's
// values and row ids side by side in ONE record of R = align16(s * (8 + sizeof(IR))) bytes -- x[0..s) then ir[0..s) --
// so that a point is a whole number of aligned 16-B pieces in one place (s = 51, 16-bit ids: exactly 512 B = four
// 128-B lines; in the two separate arrays the same point straddles 5 + 2 lines of two DRAM pages).  In cluster order
// the points of a segment are gathered through `perm`: with data in arbitrary order every point is a random access,
// and the record halves the pages and removes the partially used lines.
template <typename IR>
__global__ __launch_bounds__(256) void k_build_records(const IR* __restrict__ ir, const double* __restrict__ x,
                                                       long long n, int s, int R, char* __restrict__ rec)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long i = wave; i < n; i += nwaves) {
        char* dst = rec + (size_t)i * R;
        double* dx = reinterpret_cast<double*>(dst);
        IR* dr = reinterpret_cast<IR*>(dst + (size_t)s * 8);
        for (int j = lane; j < s; j += 64) {
            dx[j] = x[(size_t)i * s + j];
            dr[j] = ir[(size_t)i * s + j];
        }
        // (the few pad bytes behind the row ids are never read)
    }
}

// The inverse of k_build_records: the CSC value / row-id arrays of a fixed-stride shard from its record layout (entry points
// that read CSC after spkm_shard_release_csc re-materialise them first: api_lloyd.hip, ensure_csc).
template <typename IR>
__global__ __launch_bounds__(256) void k_unpack_records(const char* __restrict__ rec, long long n, int s, int R,
                                                        IR* __restrict__ ir, double* __restrict__ x)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long i = wave; i < n; i += nwaves) {
        const char* src = rec + (size_t)i * R;
        const double* sx = reinterpret_cast<const double*>(src);
        const IR* sr = reinterpret_cast<const IR*>(src + (size_t)s * 8);
        for (int j = lane; j < s; j += 64) {
            x[(size_t)i * s + j] = sx[j];
            ir[(size_t)i * s + j] = sr[j];
        }
    }
}

// distances(i) = the reference's distance of point i to centroid assign[i], squared terms added in storage order
// (SparseMatrixMinusCluster.c:173-180 for that one centroid): what spkm_distances_dev returns when the library's kept
// counting sort does not describe `assign`.  One wave per point, any column length; Cs = row-major scaled centres.
template <typename IR>
__global__ __launch_bounds__(256) void k_point_distances(const long long* __restrict__ jc, const IR* __restrict__ ir,
                                                         const double* __restrict__ xval, const double* __restrict__ Cs,
                                                         int K, long long n, int fixed_s, const int* __restrict__ assign,
                                                         double* __restrict__ mind)
{
    __shared__ double stage[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long i = wave; i < n; i += nwaves) {
        const long long j0 = fixed_s > 0 ? i * fixed_s : jc[i];
        const long long j1 = fixed_s > 0 ? j0 + fixed_s : jc[i + 1];
        int k = assign[i];
        if ((unsigned)k >= (unsigned)K) k = 0;
        double acc = 0.0;
        for (long long jb = j0; jb < j1; jb += 64) {
            const long long j = jb + lane;
            if (j < j1) {
                const double d = xval[j] - Cs[(size_t)ir[j] * K + k];
                stage[w][lane] = d * d;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const int cnt = (int)((j1 - jb < 64) ? j1 - jb : 64);
                for (int t = 0; t < cnt; t++) acc = acc + stage[w][t];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) mind[i] = sqrt(acc);
    }
}

// (map != nullptr: b is in a regrouped shard's own order, b[i] belongs to a[map[i]])
__global__ __launch_bounds__(256) void k_count_diff_i32(const int* __restrict__ a, const int* __restrict__ b, long long n,
                                                        unsigned* __restrict__ out, const int* __restrict__ map = nullptr)
{
    unsigned c = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        c += a[map != nullptr ? map[i] : i] != b[i];
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

template <typename IR, int U, int WPE, bool NT, bool REC>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_exact_accumulate(
                                                          const char* __restrict__ rec, int R,
                                                          const IR* __restrict__ ir, const double* __restrict__ x,
                                                          const int* __restrict__ perm,
                                                          const long long* __restrict__ offs,
                                                          const int4* __restrict__ items,
                                                          const int* __restrict__ nitems,
                                                          const double* __restrict__ C, double gamma, int p,
                                                          int fixed_s, int pts, double* __restrict__ mind,
                                                          float* __restrict__ ubv, // carried bounds: ub[i] >= true distance (or null)
                                                          double* __restrict__ sums, double* __restrict__ counts,
                                                          double* __restrict__ blk_obj2,
                                                          double* __restrict__ blk_max,
                                                          long long* __restrict__ blk_imax)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* negc = reinterpret_cast<double*>(smem);
    double* ssum = negc + p;
    unsigned int* scnt = reinterpret_cast<unsigned int*>(ssum + p);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int S1 = fixed_s | 1;
    char* wbase = smem + (size_t)p * 20 + (size_t)((p & 1) ? 4 : 0);
    double* ms = reinterpret_cast<double*>(wbase) + (size_t)wave * pts * S1;
    __shared__ double s_obj[16], s_max[16];
    __shared__ long long s_imax[16];

    double obj2 = 0.0, dmax = -1.0;
    long long imax = 0x7fffffffffffffffLL;
    for (int item = blockIdx.x; item < *nitems; item += gridDim.x) {
        const int4 it = items[item];
        const int k = it.x;
        const long long start = offs[k] + it.y;
        const int len = it.z;
        for (int r = tid; r < p; r += blockDim.x) {
            double c = C[(size_t)k * p + r];
            if (gamma > 0.0) c = c / gamma;
            negc[r] = -c;
            ssum[r] = 0.0;
            scnt[r] = 0u;
        }
        __syncthreads();
        // the point ids of a wave's next batch are fetched one batch ahead (takes their latency off the chain)
        long long my_next = 0;
        if (wave * pts < len && lane < pts && wave * pts + lane < len) my_next = perm[start + wave * pts + lane];
        for (int qb = wave * pts; qb < len; qb += nwaves * pts) {
            const int have = (len - qb < pts) ? len - qb : pts;
            const long long my_i = my_next;
            {
                const int qn = qb + nwaves * pts;
                if (qn < len && lane < pts && qn + lane < len) my_next = perm[start + qn + lane];
            }
            // (a) U points' loads in flight per lane
            const int lanec = lane < fixed_s ? lane : fixed_s - 1;
            const unsigned offx = (unsigned)lanec * 8u;                                              // record path: byte offsets
            const unsigned offr = (unsigned)fixed_s * 8u + (unsigned)lanec * (unsigned)sizeof(IR);   // inside a point's record
            for (int u = 0; u < have; u += U) {
                double xv[U];
                int rv[U];
                if constexpr (REC) {
                    // one uniform base per point (an SGPR pair), two loop-invariant lane offsets
#pragma unroll
                    for (int v = 0; v < U; v++) {
                        const int src = (u + v < have) ? u + v : u;
                        const long long i = (long long)(unsigned)__builtin_amdgcn_readlane((int)my_i, src);
                        const bool ok = (u + v < have) && lane < fixed_s;
                        const char* b = rec + (size_t)i * (size_t)R;
                        int r;
                        if constexpr (NT) { // streamed once per launch
                            xv[v] = __builtin_nontemporal_load(reinterpret_cast<const double*>(b + offx));
                            r = (int)__builtin_nontemporal_load(reinterpret_cast<const IR*>(b + offr));
                        } else {
                            xv[v] = *reinterpret_cast<const double*>(b + offx);
                            r = (int)*reinterpret_cast<const IR*>(b + offr);
                        }
                        rv[v] = ok ? r : -1;
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < U; v++) {
                        // unconditional loads through a uniform per-point base + a clamped 32-bit lane offset (no
                        // divergent control flow around the loads); validity is applied to the row id afterwards
                        const int src = (u + v < have) ? u + v : u;
                        const long long i = (long long)(unsigned)__builtin_amdgcn_readlane((int)my_i, src);
                        const bool ok = (u + v < have) && lane < fixed_s;
                        const double* xb = x + i * fixed_s;
                        const IR* rb = ir + i * fixed_s;
                        int r;
                        if constexpr (NT) {
                            xv[v] = __builtin_nontemporal_load(xb + lanec);
                            r = (int)__builtin_nontemporal_load(rb + lanec);
                        } else {
                            xv[v] = xb[lanec];
                            r = (int)rb[lanec];
                        }
                        rv[v] = ok ? r : -1;
                    }
                }
#pragma unroll
                for (int v = 0; v < U; v++) {
                    if (rv[v] >= 0) {
                        const double d = xv[v] + negc[rv[v]]; // RN(x - c): the reference's subtraction
                        ms[(size_t)(u + v) * S1 + lane] = d * d;
                        unsafeAtomicAdd(&ssum[rv[v]], xv[v]);
                        atomicAdd(&scnt[rv[v]], 1u);
                    }
                    if (fixed_s > 64 && u + v < have) { // columns longer than one wave
                        const long long i = (long long)(unsigned)__builtin_amdgcn_readlane((int)my_i, u + v);
                        for (int e = 64 + lane; e < fixed_s; e += 64) {
                            const double xe = REC ? reinterpret_cast<const double*>(rec + (size_t)i * R)[e] : x[i * fixed_s + e];
                            const int re = REC ? (int)reinterpret_cast<const IR*>(rec + (size_t)i * R + (size_t)fixed_s * 8)[e]
                                               : (int)ir[i * fixed_s + e];
                            const double d = xe + negc[re];
                            ms[(size_t)(u + v) * S1 + e] = d * d;
                            unsafeAtomicAdd(&ssum[re], xe);
                            atomicAdd(&scnt[re], 1u);
                        }
                    }
                }
            }
            // (b) one lane per point, squared terms added in storage order (wave-synchronous: the
            // staged data was written by this wave; LDS ops of a wave complete in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < have) {
                double acc = 0.0;
                const double* mq = ms + (size_t)lane * S1;
                // the additions are a dependent chain in storage order; the LDS reads are not: eight in flight
                int j = 0;
                for (; j + 8 <= fixed_s; j += 8) {
                    const double m0 = mq[j], m1 = mq[j + 1], m2 = mq[j + 2], m3 = mq[j + 3], m4 = mq[j + 4],
                                 m5 = mq[j + 5], m6 = mq[j + 6], m7 = mq[j + 7];
                    acc = acc + m0; acc = acc + m1; acc = acc + m2; acc = acc + m3;
                    acc = acc + m4; acc = acc + m5; acc = acc + m6; acc = acc + m7;
                }
                for (; j < fixed_s; j++) acc = acc + mq[j];
                const double dist = sqrt(acc);
                if (mind) mind[my_i] = dist; // (null: the caller does not need per-point distances from this call)
                if (ubv) ubv[my_i] = __double2float_ru(dist * (1.0 + 1e-12)); // the reference value is within 2^-45 of the true one
                obj2 += dist * dist;
                if (dist > dmax || (dist == dmax && my_i < imax)) { dmax = dist; imax = my_i; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        for (int r = tid; r < p; r += blockDim.x) {
            const unsigned int c = scnt[r];
            if (c) {
                unsafeAtomicAdd(&sums[(size_t)k * p + r], ssum[r]);
                unsafeAtomicAdd(&counts[(size_t)k * p + r], (double)c);
            }
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) {
        obj2 += __shfl_down(obj2, off);
        const double om = __shfl_down(dmax, off);
        const long long oi = __shfl_down(imax, off);
        if (om > dmax || (om == dmax && oi < imax)) { dmax = om; imax = oi; }
    }
    if (lane == 0) { s_obj[wave] = obj2; s_max[wave] = dmax; s_imax[wave] = imax; }
    __syncthreads();
    if (tid == 0) {
        double o = 0.0, m = -1.0;
        long long im = 0x7fffffffffffffffLL;
        for (int w = 0; w < nwaves; w++) {
            o += s_obj[w];
            if (s_max[w] > m || (s_max[w] == m && s_imax[w] < im)) { m = s_max[w]; im = s_imax[w]; }
        }
        blk_obj2[blockIdx.x] = o;
        blk_max[blockIdx.x] = m;
        blk_imax[blockIdx.x] = im;
    }
}

// The same pass over the record layout, software-pipelined: a wave's batch is P = 16 points whose loads (one 8-B and
// one id load per lane and point, 48 VGPRs) are issued BEFORE the storage-order sums of the previous batch, so that a
// wave has its next 8 KB in flight while it works through phase (b) -- in k_exact_accumulate a wave's loads are only
// in flight while it waits for them.  (a2) consumes the registers, then the next batch is issued into the same
// registers, then (b) runs on what (a2) staged in LDS.  Columns of up to 64 entries; outputs bit-identical to
// k_exact_accumulate (same arithmetic, same order).
// ACCUM = false: the distance-only variant of spkm_distances_dev -- no sum / count atomics, no flush; `sums` and `counts`
// may be null.
// DIST = false: the sums-only variant of a LAZY call's full pass (spkm_shard_set_lazy_stats: no distance, objective or
// largest distance is wanted, the upper bounds come from the screen's certificate): no centroid reads, no staging of the
// squared terms, no per-point sums; mind / ubv / it_* are not touched.
template <typename IR, int WPE, bool ACCUM = true, bool DIST = true>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_exact_accumulate_rec(
    const char* __restrict__ rec, int R, const int* __restrict__ perm, const long long* __restrict__ offs,
    const int4* __restrict__ items, const int* __restrict__ nitems, const double* __restrict__ C, double gamma, int p,
    int fixed_s, double* __restrict__ mind, float* __restrict__ ubv, double* __restrict__ sums,
    double* __restrict__ counts, double* __restrict__ it_obj2, double* __restrict__ it_max,
    long long* __restrict__ it_imax) // per work ITEM: sum of squared distances, largest distance, its first point index
{
    constexpr int P = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* negc = reinterpret_cast<double*>(smem);
    double* ssum = negc + p;
    unsigned int* scnt = reinterpret_cast<unsigned int*>(ssum + p);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int S1 = fixed_s | 1;
    char* wbase = smem + (size_t)p * 20 + (size_t)((p & 1) ? 4 : 0);
    double* ms = reinterpret_cast<double*>(wbase) + (size_t)wave * P * S1;
    __shared__ double s_obj[16], s_max[16];
    __shared__ long long s_imax[16];
    const int lanec = lane < fixed_s ? lane : fixed_s - 1;
    const unsigned offx = (unsigned)lanec * 8u;
    const unsigned offr = (unsigned)fixed_s * 8u + (unsigned)lanec * (unsigned)sizeof(IR);
    const bool lane_ok = lane < fixed_s;

    double obj2 = 0.0, dmax = -1.0;
    long long imax = 0x7fffffffffffffffLL;
    double xv[P];
    int rv[P];
    for (int item = blockIdx.x; item < *nitems; item += gridDim.x) {
        const int4 it = items[item];
        const int k = it.x;
        const long long start = offs[k] + it.y;
        const int len = it.z;
        for (int r = tid; r < p; r += blockDim.x) {
            if constexpr (DIST) {
                double c = C[(size_t)k * p + r];
                if (gamma > 0.0) c = c / gamma;
                negc[r] = -c;
            }
            ssum[r] = 0.0;
            scnt[r] = 0u;
        }
        const int stride = nwaves * P;
        int qb = wave * P;
        // point ids of a wave's batches are fetched two batches ahead, UNCONDITIONALLY from a clamped position (a load
        // inside a divergent branch is waited for at the end of the branch -- together with every load issued before it)
        const int lane_p = lane < P ? lane : P - 1;
        int my_i = perm[start + min(qb + lane_p, len - 1)];
        int my_next = perm[start + min(qb + stride + lane_p, len - 1)];
        // prologue: the first batch's loads
        if (qb < len) {
            const int have = (len - qb < P) ? len - qb : P;
#pragma unroll
            for (int v = 0; v < P; v++) {
                const int src = (v < have) ? v : 0;
                const long long i = (long long)(unsigned)__builtin_amdgcn_readlane(my_i, src);
                const char* b = rec + (size_t)i * (size_t)R;
                xv[v] = __builtin_nontemporal_load(reinterpret_cast<const double*>(b + offx));
                rv[v] = (int)__builtin_nontemporal_load(reinterpret_cast<const IR*>(b + offr));
            }
        }
        __syncthreads(); // slab ready
        for (; qb < len; qb += stride) {
            const int have = (len - qb < P) ? len - qb : P;
            // (a2) squared terms to LDS in lanes <-> entries order; per-cluster sums / counts with LDS atomics
            // The centroid reads run one group of four points AHEAD of the group being processed, unconditionally (every
            // lane holds a valid row id, clamped lanes and slots past `have` included): LDS operations complete in
            // order, so a read issued behind a group's staging writes and atomics would wait for all of them; issued in
            // front of them it is only ever waited for together with older reads.
            double cg[2][4];
            if constexpr (DIST) {
#pragma unroll
                for (int t = 0; t < 4; t++) cg[0][t] = negc[rv[t]];
            }
#pragma unroll
            for (int g = 0; g < P; g += 4) {
                if constexpr (DIST) {
                    if (g + 4 < P) {
#pragma unroll
                        for (int t = 0; t < 4; t++) cg[((g >> 2) + 1) & 1][t] = negc[rv[g + 4 + t]];
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int v = g + t;
                    if (v < have && lane_ok) {
                        const int r = rv[v];
                        if constexpr (DIST) {
                            const double d = xv[v] + cg[(g >> 2) & 1][t]; // RN(x - c): the reference's subtraction
                            ms[(size_t)v * S1 + lane] = d * d;
                        }
                        if constexpr (ACCUM) {
                            unsafeAtomicAdd(&ssum[r], xv[v]);
                            atomicAdd(&scnt[r], 1u);
                        }
                    }
                }
            }
            // the next batch: ids were fetched one batch ahead; its loads fly during (b)
            const long long ids_done = (long long)(unsigned)my_i;
            const int qn = qb + stride;
            my_i = my_next;
            if (qn < len) {
                const int have_n = (len - qn < P) ? len - qn : P;
#pragma unroll
                for (int v = 0; v < P; v++) {
                    const int src = (v < have_n) ? v : 0;
                    const long long i = (long long)(unsigned)__builtin_amdgcn_readlane(my_i, src);
                    const char* b = rec + (size_t)i * (size_t)R;
                    xv[v] = __builtin_nontemporal_load(reinterpret_cast<const double*>(b + offx));
                    rv[v] = (int)__builtin_nontemporal_load(reinterpret_cast<const IR*>(b + offr));
                }
                my_next = perm[start + min(qn + stride + lane_p, len - 1)];
            }
            // (b) one lane per point, squared terms added in storage order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (DIST && lane < have) {
                double acc = 0.0;
                const double* mq = ms + (size_t)lane * S1;
                int j = 0;
                for (; j + 8 <= fixed_s; j += 8) {
                    const double m0 = mq[j], m1 = mq[j + 1], m2 = mq[j + 2], m3 = mq[j + 3], m4 = mq[j + 4],
                                 m5 = mq[j + 5], m6 = mq[j + 6], m7 = mq[j + 7];
                    acc = acc + m0; acc = acc + m1; acc = acc + m2; acc = acc + m3;
                    acc = acc + m4; acc = acc + m5; acc = acc + m6; acc = acc + m7;
                }
                for (; j < fixed_s; j++) acc = acc + mq[j];
                const double dist = sqrt(acc);
                if (mind) mind[ids_done] = dist; // (null: the caller does not need per-point distances from this call)
                if (ubv) ubv[ids_done] = __double2float_ru(dist * (1.0 + 1e-12));
                obj2 += dist * dist;
                if (dist > dmax || (dist == dmax && ids_done < imax)) { dmax = dist; imax = ids_done; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // this item's statistics: wave -> LDS here, combined by thread 0 behind the barrier the flush needs anyway
        for (int off = 32; off > 0; off >>= 1) {
            obj2 += __shfl_down(obj2, off);
            const double om = __shfl_down(dmax, off);
            const long long oi = __shfl_down(imax, off);
            if (om > dmax || (om == dmax && oi < imax)) { dmax = om; imax = oi; }
        }
        if (lane == 0) { s_obj[wave] = obj2; s_max[wave] = dmax; s_imax[wave] = imax; }
        obj2 = 0.0; dmax = -1.0; imax = 0x7fffffffffffffffLL;
        __syncthreads();
        if (DIST && tid == 0) {
            double o = 0.0, m = -1.0;
            long long im = 0x7fffffffffffffffLL;
            for (int w = 0; w < nwaves; w++) {
                o += s_obj[w];
                if (s_max[w] > m || (s_max[w] == m && s_imax[w] < im)) { m = s_max[w]; im = s_imax[w]; }
            }
            it_obj2[item] = o;
            it_max[item] = m;
            it_imax[item] = im;
        }
        if constexpr (ACCUM) {
            for (int r = tid; r < p; r += blockDim.x) {
                const unsigned int c = scnt[r];
                if (c) {
                    unsafeAtomicAdd(&sums[(size_t)k * p + r], ssum[r]);
                    unsafeAtomicAdd(&counts[(size_t)k * p + r], (double)c);
                }
            }
        }
        __syncthreads();
    }
}

template __global__ void k_screen_tile<unsigned short>(const unsigned short*, const float*, const float*, int, int,
    int, int, const spkm_blockmap*, int, float*, float*, int*);
template __global__ void k_screen_tile<unsigned int>(const unsigned int*, const float*, const float*, int, int, int,
    int, const spkm_blockmap*, int, float*, float*, int*);


// ============================================================================================
// K = 1: the distance of every point to ONE centre (the k-means++ rounds, Arthur_initialization.m:39 through
// SparseMatrixMinusCluster.c:133-141).  Nothing to minimise over, so the work is a stream over X: the same
// lanes <-> entries load + LDS transposition as k_exact_accumulate (separately rounded subtract and multiply,
// squared terms added in storage order), without the counting sort and the accumulation slabs.  HBM bound.
// LDS: negc f64[p] | per wave ms f64[pts * S1]
template <typename IR, int U>
__global__ __launch_bounds__(1024) void k_exact_dist1(const IR* __restrict__ ir, const double* __restrict__ x,
                                                      const double* __restrict__ C, double gamma, int p, long long n,
                                                      int fixed_s, int pts, int* __restrict__ assign,
                                                      double* __restrict__ mind, double* __restrict__ blk_obj2,
                                                      double* __restrict__ blk_max, long long* __restrict__ blk_imax)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* negc = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int S1 = fixed_s | 1;
    double* ms = negc + p + (size_t)wave * pts * S1;
    __shared__ double s_obj[16], s_max[16];
    __shared__ long long s_imax[16];
    for (int r = tid; r < p; r += blockDim.x) {
        double c = C[r];
        if (gamma > 0.0) c = c / gamma;
        negc[r] = -c;
    }
    __syncthreads();
    double obj2 = 0.0, dmax = -1.0;
    long long imax = 0x7fffffffffffffffLL;
    const long long stride = (long long)gridDim.x * nwaves * pts;
    for (long long b0 = ((long long)blockIdx.x * nwaves + wave) * pts; b0 < n; b0 += stride) {
        const int have = (n - b0 < pts) ? (int)(n - b0) : pts;
        // uniform base pointers + 32-bit lane offsets; loads are unconditional (clamped), validity is applied to
        // the staging store only -- no divergent control flow around the loads
        const double* xb = x + b0 * fixed_s;
        const IR* rb = ir + b0 * fixed_s;
        const int lanec = lane < fixed_s ? lane : fixed_s - 1;
        for (int u = 0; u < have; u += U) {
            double xv[U];
            int rv[U];
#pragma unroll
            for (int v = 0; v < U; v++) {
                const int pc = (u + v < have) ? u + v : have - 1;
                const int off = pc * fixed_s + lanec;
                xv[v] = xb[off];
                rv[v] = (int)rb[off];
            }
#pragma unroll
            for (int v = 0; v < U; v++) {
                const double d = xv[v] + negc[rv[v]]; // RN(x - c): the reference's subtraction
                if (u + v < have && lane < fixed_s) ms[(size_t)(u + v) * S1 + lane] = d * d;
                if (fixed_s > 64 && u + v < have) { // columns longer than one wave
                    for (int e = 64 + lane; e < fixed_s; e += 64) {
                        const double de = xb[(u + v) * fixed_s + e] + negc[(int)rb[(u + v) * fixed_s + e]];
                        ms[(size_t)(u + v) * S1 + e] = de * de;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < have) {
            double acc = 0.0;
            const double* mq = ms + (size_t)lane * S1;
            int j = 0; // dependent chain of additions in storage order, eight independent LDS reads in flight
            for (; j + 8 <= fixed_s; j += 8) {
                const double m0 = mq[j], m1 = mq[j + 1], m2 = mq[j + 2], m3 = mq[j + 3], m4 = mq[j + 4], m5 = mq[j + 5],
                             m6 = mq[j + 6], m7 = mq[j + 7];
                acc = acc + m0; acc = acc + m1; acc = acc + m2; acc = acc + m3;
                acc = acc + m4; acc = acc + m5; acc = acc + m6; acc = acc + m7;
            }
            for (; j < fixed_s; j++) acc = acc + mq[j];
            const double dist = sqrt(acc);
            const long long i = b0 + lane;
            mind[i] = dist;
            assign[i] = 0;
            obj2 += dist * dist;
            if (dist > dmax || (dist == dmax && i < imax)) { dmax = dist; imax = i; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int off = 32; off > 0; off >>= 1) {
        obj2 += __shfl_down(obj2, off);
        const double om = __shfl_down(dmax, off);
        const long long oi = __shfl_down(imax, off);
        if (om > dmax || (om == dmax && oi < imax)) { dmax = om; imax = oi; }
    }
    if (lane == 0) { s_obj[wave] = obj2; s_max[wave] = dmax; s_imax[wave] = imax; }
    __syncthreads();
    if (tid == 0) {
        double o = 0.0, m = -1.0;
        long long im = 0x7fffffffffffffffLL;
        for (int w = 0; w < nwaves; w++) {
            o += s_obj[w];
            if (s_max[w] > m || (s_max[w] == m && s_imax[w] < im)) { m = s_max[w]; im = s_imax[w]; }
        }
        blk_obj2[blockIdx.x] = o;
        blk_max[blockIdx.x] = m;
        blk_imax[blockIdx.x] = im;
    }
}

__global__ void k_set_u64(unsigned long long* __restrict__ dst, unsigned long long v) { dst[0] = v; }
