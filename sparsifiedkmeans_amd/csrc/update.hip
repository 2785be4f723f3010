// Lloyd-iteration support kernels around the tiled assignment kernel (gfx950).
//
//   k_prep_tiles        centers/gamma (findClusterAssignments.m:78), negated + transposed into LDS tiles
//   k_combine           [d,a]=min(...) across tiles (findClusterAssignments.m:169) + per-shard statistics
//   k_reduce_stats      fixed-order reduction of the per-block statistics
//   k_accumulate_*      per-cluster sums / counts            (kmeans_sparsified.m:430-431,447-448)
//   k_finalize_centers  gamma*S./(Cnt+1e-16), dff            (kmeans_sparsified.m:448,470)
#include "common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

// negCt[g][r][kk] = -(C[(g*KT+kk)*p + r] / gamma)   r < p, g*KT+kk < K;  0 elsewhere (row p is the zero row).
// gamma <= 0 means "no scaling" (findClusterAssignments.m:80).  The divide is a true IEEE divide.
__global__ void k_prep_tiles(const double* __restrict__ C, int p, int K, int KT, int G, double gamma,
                             double* __restrict__ negCt)
{
    const size_t total = (size_t)G * (p + 1) * KT;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(t % KT);
        const size_t rest = t / KT;
        const int r = (int)(rest % (p + 1));
        const int g = (int)(rest / (p + 1));
        const int k = g * KT + kk;
        double v = 0.0;
        if (r < p && k < K) {
            v = C[(size_t)k * p + r];
            if (gamma > 0.0) v = v / gamma;
            v = -v;
        }
        negCt[t] = v;
    }
}

// Per point: winner over the G tile partials under the reference's order
// "smallest sqrt(acc), first index on ties".  Tiles are ordered by k, so a strict
// '<' keeps the lowest k.  Writes assign (0-based) and mind = sqrt(acc_win), and
// per-block partial statistics (deterministic for a fixed launch geometry):
//   blk_obj2[b] = sum mind^2,  blk_max[b], blk_imax[b] = max mind and its first index,
// plus the cluster histogram nk[K] (integer atomics: order-independent, exact).
__global__ __launch_bounds__(256) void k_combine(const double* __restrict__ part_acc,
                                                 const int* __restrict__ part_k, long long n, int G, int K,
                                                 int* __restrict__ assign, double* __restrict__ mind,
                                                 double* __restrict__ blk_obj2, double* __restrict__ blk_max,
                                                 long long* __restrict__ blk_imax,
                                                 unsigned long long* __restrict__ nk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int* hist = reinterpret_cast<unsigned int*>(smem); // K entries
    __shared__ double s_obj[4], s_max[4];
    __shared__ long long s_imax[4];
    const int tid = threadIdx.x;
    for (int k = tid; k < K; k += blockDim.x) hist[k] = 0;
    __syncthreads();

    double obj2 = 0.0, dmax = -1.0;
    long long imax = 0x7fffffffffffffffLL;
    // contiguous slab per block so that per-block partials are independent of gridDim ordering games
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per;
    const long long hi = (lo + per < n) ? lo + per : n;
    for (long long i = lo + tid; i < hi; i += blockDim.x) {
        double best = sqrt(part_acc[i]);
        int bk = part_k[i];
        for (int g = 1; g < G; g++) {
            const double d = sqrt(part_acc[(size_t)g * n + i]);
            if (d < best) { best = d; bk = part_k[(size_t)g * n + i]; }
        }
        // non-finite input (Inf / NaN distances to every centroid): no '<' ever held and the tile kernels left their
        // sentinel.  MATLAB's min() returns index 1 there; what matters on this side of the ABI is that everything
        // downstream (histogram, counting sort, accumulation) indexes with a valid cluster.
        if ((unsigned)bk >= (unsigned)K) bk = 0;
        assign[i] = bk;
        mind[i] = best;
        obj2 += best * best;
        if (best > dmax) { dmax = best; imax = i; } // i ascending per thread: first index kept
        atomicAdd(&hist[bk], 1u);
    }
    // wave reduce (sum is order-fixed by the shuffle tree)
    for (int off = 32; off > 0; off >>= 1) {
        obj2 += __shfl_down(obj2, off);
        const double om = __shfl_down(dmax, off);
        const long long oi = __shfl_down(imax, off);
        if (om > dmax || (om == dmax && oi < imax)) { dmax = om; imax = oi; }
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) { s_obj[wave] = obj2; s_max[wave] = dmax; s_imax[wave] = imax; }
    __syncthreads();
    if (tid == 0) {
        double o = 0.0, m = -1.0;
        long long im = 0x7fffffffffffffffLL;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
            o += s_obj[w];
            if (s_max[w] > m || (s_max[w] == m && s_imax[w] < im)) { m = s_max[w]; im = s_imax[w]; }
        }
        blk_obj2[blockIdx.x] = o;
        blk_max[blockIdx.x] = m;
        blk_imax[blockIdx.x] = im;
    }
    for (int k = tid; k < K; k += blockDim.x)
        if (hist[k]) atomicAdd(&nk[k], (unsigned long long)hist[k]);
}

// ---------------- k-means++ seeding on the device (private/Arthur_initialization.m:38-69) ----------------
// A round evaluates the distance of every point to the NEWEST centre only (spkm_assign_dev, K = 1) and keeps a running
// minimum -- min() is exact and a (point, centre) distance does not depend on the other centres, so the vector equals
// the reference's full recomputation bit for bit -- then draws the next centre with probability proportional to
// dist.^2 (randsample(n,1,true,dist.^2), :50): inclusive prefix sums of dist.^2 in a fixed order (block partial sums,
// one-block scan of the partials, per-block scan) and a search for the first index whose cumulative weight exceeds
// u x total, u the HOST's uniform random number (the random stream stays the host's).
#define KPP_BLOCK 1024
// The prefix sums must be MONOTONE (the draw is a binary search), so every partial result is built by the same chain of
// additions wherever it is needed: a thread owns 4 consecutive elements (a0 <= a1 <= a2 <= a3 their running sums), thread
// 0 walks the 256 thread totals serially (e[t+1] = fl(e[t] + a3[t])), a block's total is fl(e[255] + a3[255]), and the
// block offsets are one serial chain part[b+1] = fl(part[b] + total[b]).  Then cum = fl(part[b] + fl(e[t] + a_j)) never
// decreases: floating-point addition is monotone in each argument.
__device__ __forceinline__ void kpp_block_prefix(const double* sh /* KPP_BLOCK squares */, double* seg /* 256 */,
                                                 double& a0, double& a1, double& a2, double& a3, double& e)
{
    const int e0 = threadIdx.x * 4;
    a0 = sh[e0]; a1 = a0 + sh[e0 + 1]; a2 = a1 + sh[e0 + 2]; a3 = a2 + sh[e0 + 3];
    seg[threadIdx.x] = a3;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int t = 0; t < 256; t++) { const double v = seg[t]; seg[t] = run; run = run + v; }
    }
    __syncthreads();
    e = seg[threadIdx.x];
}
// run[i] = min(run[i], dnew[i]) (first round: run = dnew); total[b] = sum over block b of run[i]^2
__global__ __launch_bounds__(256) void k_kpp_min_partial(const double* __restrict__ dnew, double* __restrict__ run,
                                                         long long n, int first_round, double* __restrict__ total)
{
    __shared__ double sh[KPP_BLOCK];
    __shared__ double seg[256];
    const long long base = (long long)blockIdx.x * KPP_BLOCK;
    for (int t = threadIdx.x; t < KPP_BLOCK; t += 256) {
        const long long i = base + t;
        double v = 0.0;
        if (i < n) {
            v = dnew[i];
            if (!first_round) { const double r = run[i]; v = r < v ? r : v; }
            run[i] = v;
        }
        sh[t] = v * v;
    }
    __syncthreads();
    double a0, a1, a2, a3, e;
    kpp_block_prefix(sh, seg, a0, a1, a2, a3, e);
    if (threadIdx.x == 255) total[blockIdx.x] = e + a3;
}
// part[b] = exclusive serial prefix of the block totals, in place; part[nb] = grand total (one thread: nb <= 1e5 additions)
__global__ void k_kpp_scan_partials(double* __restrict__ part, int nb)
{
    if (blockIdx.x || threadIdx.x) return;
    double run = 0.0;
    for (int b = 0; b < nb; b++) { const double v = part[b]; part[b] = run; run = run + v; }
    part[nb] = run;
}
// cum[i] = part[block] + (e[thread] + a_j)
__global__ __launch_bounds__(256) void k_kpp_block_scan(const double* __restrict__ run, long long n,
                                                        const double* __restrict__ part, double* __restrict__ cum)
{
    __shared__ double sh[KPP_BLOCK];
    __shared__ double seg[256];
    const long long base = (long long)blockIdx.x * KPP_BLOCK;
    for (int t = threadIdx.x; t < KPP_BLOCK; t += 256) { const long long i = base + t; const double v = i < n ? run[i] : 0.0; sh[t] = v * v; }
    __syncthreads();
    double a0, a1, a2, a3, e;
    kpp_block_prefix(sh, seg, a0, a1, a2, a3, e);
    const double off = part[blockIdx.x];
    const long long i = base + threadIdx.x * 4;
    if (i < n) cum[i] = off + (e + a0);
    if (i + 1 < n) cum[i + 1] = off + (e + a1);
    if (i + 2 < n) cum[i + 2] = off + (e + a2);
    if (i + 3 < n) cum[i + 3] = off + (e + a3);
}
// out[0] = first index i with cum[i] > target (clamped to n - 1): searchsorted(cum, target, right)
__global__ void k_kpp_search(const double* __restrict__ cum, long long n, double target, long long* __restrict__ out)
{
    if (blockIdx.x || threadIdx.x) return;
    long long lo = 0, hi = n; // first index in [lo, hi) with cum > target
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (cum[mid] > target) hi = mid; else lo = mid + 1;
    }
    out[0] = lo < n ? lo : n - 1;
}

// Per-block statistics of a vector of distances (spkm_distances_stats_dev on shards without a record layout): the same
// partials k_combine produces -- sum of squares, largest value, its first index -- over contiguous slabs.
__global__ __launch_bounds__(256) void k_mind_stats(const double* __restrict__ mind, long long n,
                                                    double* __restrict__ blk_obj2, double* __restrict__ blk_max,
                                                    long long* __restrict__ blk_imax)
{
    __shared__ double s_obj[4], s_max[4];
    __shared__ long long s_imax[4];
    const int tid = threadIdx.x;
    double obj2 = 0.0, dmax = -1.0;
    long long imax = 0x7fffffffffffffffLL;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per;
    const long long hi = (lo + per < n) ? lo + per : n;
    for (long long i = lo + tid; i < hi; i += blockDim.x) {
        const double d = mind[i];
        obj2 += d * d;
        if (d > dmax) { dmax = d; imax = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        obj2 += __shfl_down(obj2, off);
        const double om = __shfl_down(dmax, off);
        const long long oi = __shfl_down(imax, off);
        if (om > dmax || (om == dmax && oi < imax)) { dmax = om; imax = oi; }
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) { s_obj[wave] = obj2; s_max[wave] = dmax; s_imax[wave] = imax; }
    __syncthreads();
    if (tid == 0) {
        double o = 0.0, m = -1.0;
        long long im = 0x7fffffffffffffffLL;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) {
            o += s_obj[w];
            if (s_max[w] > m || (s_max[w] == m && s_imax[w] < im)) { m = s_max[w]; im = s_imax[w]; }
        }
        blk_obj2[blockIdx.x] = o;
        blk_max[blockIdx.x] = m;
        blk_imax[blockIdx.x] = im;
    }
}

// stats[0] = sum_b blk_obj2[b] (fixed order: lane-strided partials, then a fixed shuffle tree),
// stats[1] = max mind, stats[2] = its first index (as double).  One wave.
__global__ void k_reduce_stats(const double* __restrict__ blk_obj2, const double* __restrict__ blk_max,
                               const long long* __restrict__ blk_imax, int nblk, double* __restrict__ stats)
{
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    double o = 0.0, m = -1.0;
    long long im = 0x7fffffffffffffffLL;
    for (int b = lane; b < nblk; b += 64) {
        o += blk_obj2[b];
        if (blk_max[b] > m || (blk_max[b] == m && blk_imax[b] < im)) { m = blk_max[b]; im = blk_imax[b]; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        o += __shfl_down(o, off);
        const double om = __shfl_down(m, off);
        const long long oi = __shfl_down(im, off);
        if (om > m || (om == m && oi < im)) { m = om; im = oi; }
    }
    if (lane == 0) { stats[0] = o; stats[1] = m; stats[2] = (double)im; }
}

// the same over a number of entries that lives on the device (the work items the plan emitted)
__global__ void k_reduce_stats_n(const double* __restrict__ blk_obj2, const double* __restrict__ blk_max,
                                 const long long* __restrict__ blk_imax, const int* __restrict__ nblk_dev,
                                 double* __restrict__ stats)
{
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const int lane = threadIdx.x, nblk = *nblk_dev;
    double o = 0.0, m = -1.0;
    long long im = 0x7fffffffffffffffLL;
    for (int b = lane; b < nblk; b += 64) {
        o += blk_obj2[b];
        if (blk_max[b] > m || (blk_max[b] == m && blk_imax[b] < im)) { m = blk_max[b]; im = blk_imax[b]; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        o += __shfl_down(o, off);
        const double om = __shfl_down(m, off);
        const long long oi = __shfl_down(im, off);
        if (om > m || (om == m && oi < im)) { m = om; im = oi; }
    }
    if (lane == 0) { stats[0] = o; stats[1] = m; stats[2] = (double)im; }
}

// Fallback accumulation straight into the global p x K tables with f64 hardware atomics
// (used when the per-cluster LDS slab of the sorted path does not fit).  One wave per point.
template <typename IR>
__global__ __launch_bounds__(256) void k_accumulate_atomic(const long long* __restrict__ jc,
                                                           const IR* __restrict__ ir,
                                                           const double* __restrict__ x,
                                                           const int* __restrict__ assign, int p, long long n,
                                                           double* __restrict__ sums, double* __restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long i = wave; i < n; i += nwaves) {
        const int k = assign[i];
        const long long j0 = jc[i], j1 = jc[i + 1];
        for (long long j = j0 + lane; j < j1; j += 64) {
            const size_t at = (size_t)k * p + (size_t)ir[j];
            unsafeAtomicAdd(&sums[at], x[j]);
            unsafeAtomicAdd(&counts[at], 1.0);
        }
    }
}

// ---------------- sorted accumulation: counting sort by cluster + LDS slabs ----------------
// offs[k] = exclusive prefix of nk; cursor[k] = offs[k]; builds the work-item list: one item per
// (cluster, segment of <= seg points).  items[t] = {k, start, len}.  One 256-thread block: thread k
// owns cluster k (strided for K > 256); two small scans give the point and item offsets.
__global__ __launch_bounds__(256) void k_plan_segments(const unsigned long long* __restrict__ nk, int K, int seg,
                                                       long long* __restrict__ offs,
                                                       unsigned long long* __restrict__ cursor,
                                                       int4* __restrict__ items, int* __restrict__ nitems,
                                                       const unsigned* __restrict__ gate,
                                                       const int* __restrict__ need = nullptr,  // items only for need[k] != 0
                                                       int* __restrict__ ibeg = nullptr, int* __restrict__ icnt = nullptr,
                                                       unsigned long long* __restrict__ zero = nullptr, int zero_n = 0)
{
    __shared__ long long s_pts[256];
    if (gate != nullptr && *gate == 0u) return; // nothing changed: the previous plan stands (see k_hist)
    // zero: a table the NEXT kernels of the stream accumulate into (pair events: the second-level histogram), cleared on the way
    for (int t = threadIdx.x; t < zero_n; t += blockDim.x) zero[t] = 0ull;
    __shared__ int s_items[256];
    __shared__ long long s_run_pts;
    __shared__ int s_run_items;
    const int tid = threadIdx.x;
    if (tid == 0) { s_run_pts = 0; s_run_items = 0; }
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += 256) {
        const int k = k0 + tid;
        const long long cnt = (k < K) ? (long long)nk[k] : 0;
        const int nseg = (k < K && need != nullptr && !need[k]) ? 0 : (int)((cnt + seg - 1) / seg);
        s_pts[tid] = cnt;
        s_items[tid] = nseg;
        __syncthreads();
        // Hillis-Steele inclusive scans over the 256 slots
        for (int off = 1; off < 256; off <<= 1) {
            const long long a = (tid >= off) ? s_pts[tid - off] : 0;
            const int b = (tid >= off) ? s_items[tid - off] : 0;
            __syncthreads();
            s_pts[tid] += a;
            s_items[tid] += b;
            __syncthreads();
        }
        const long long pbase = s_run_pts + s_pts[tid] - cnt;
        const int ibase = s_run_items + s_items[tid] - nseg;
        if (k < K) {
            offs[k] = pbase;
            cursor[k] = (unsigned long long)pbase;
            if (ibeg) { ibeg[k] = ibase; icnt[k] = nseg; }
            for (int s = 0; s < nseg; s++) {
                const long long st = (long long)s * seg;
                const long long len = (cnt - st < seg) ? cnt - st : seg;
                items[ibase + s] = make_int4(k, (int)st, (int)len, 0);
            }
        }
        __syncthreads();
        if (tid == 255) { s_run_pts += s_pts[255]; s_run_items += s_items[255]; }
        __syncthreads();
    }
    if (tid == 0) { offs[K] = s_run_pts; *nitems = s_run_items; }
}

// The same plan for MANY keys (pair events: K (K + 1) of them, 10^4 at K = 100).  One workgroup of 1024 threads; the counts
// are staged in LDS with coalesced loads (u32: an event list holds fewer than 2^31), every thread owns a contiguous range of
// keys, ONE scan over the 1024 range totals, and the offsets go back through LDS so that the stores are coalesced too
// (a scan per 256 keys took 0.2-0.3 ms per call, a strided walk over global memory 0.12 -- fixed costs that an eighth of
// the data pays in full).  Dynamic LDS: 4 K bytes.
__global__ __launch_bounds__(1024) void k_plan_segments_wide(const unsigned long long* __restrict__ nk, int K, int seg,
                                                             long long* __restrict__ offs,
                                                             unsigned long long* __restrict__ cursor,
                                                             int4* __restrict__ items, int* __restrict__ nitems,
                                                             const unsigned* __restrict__ gate)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned s_pts[1024];
    __shared__ int s_items[1024];
    if (gate != nullptr && *gate == 0u) return;
    unsigned* cnt = reinterpret_cast<unsigned*>(smem); // K counts, then K offsets in place
    const int tid = threadIdx.x;
    for (int k = tid; k < K; k += 1024) cnt[k] = (unsigned)nk[k];
    __syncthreads();
    const int per = (K + 1023) / 1024;
    const int k_lo = min(K, tid * per), k_hi = min(K, k_lo + per);
    unsigned pts = 0;
    int its = 0;
    for (int k = k_lo; k < k_hi; k++) {
        pts += cnt[k];
        its += (int)((cnt[k] + (unsigned)seg - 1u) / (unsigned)seg);
    }
    s_pts[tid] = pts;
    s_items[tid] = its;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned a = (tid >= off) ? s_pts[tid - off] : 0u;
        const int b = (tid >= off) ? s_items[tid - off] : 0;
        __syncthreads();
        s_pts[tid] += a;
        s_items[tid] += b;
        __syncthreads();
    }
    unsigned pbase = s_pts[tid] - pts;
    int ibase = s_items[tid] - its;
    for (int k = k_lo; k < k_hi; k++) {
        const unsigned c = cnt[k];
        const int nseg = (int)((c + (unsigned)seg - 1u) / (unsigned)seg);
        for (int sg = 0; sg < nseg; sg++) {
            const unsigned st = (unsigned)sg * (unsigned)seg;
            const unsigned len = (c - st < (unsigned)seg) ? c - st : (unsigned)seg;
            items[ibase + sg] = make_int4(k, (int)st, (int)len, 0);
        }
        cnt[k] = pbase; // (the count is not needed again: the slot becomes the key's offset)
        pbase += c;
        ibase += nseg;
    }
    __syncthreads();
    for (int k = tid; k < K; k += 1024) {
        offs[k] = (long long)cnt[k];
        cursor[k] = (unsigned long long)cnt[k];
    }
    if (tid == 1023) { offs[K] = (long long)s_pts[1023]; *nitems = s_items[1023]; }
}

// perm[cursor[assign[i]]++] = i, with one global atomic per (block, cluster) via an LDS histogram.
// VEC: the assignment is read four points at a time (16-B loads; the pointer must be 16-B aligned) -- two passes over
// 4 B per point are latency bound with one 4-B load per lane in flight.
template <bool VEC>
__global__ __launch_bounds__(256) void k_scatter_by_cluster(const int* __restrict__ assign, long long n, int K,
                                                            unsigned long long* __restrict__ cursor,
                                                            int* __restrict__ perm, const unsigned* __restrict__ gate,
                                                            const int* __restrict__ need,
                                                            const unsigned* __restrict__ n_dev = nullptr,
                                                            const int* __restrict__ ids = nullptr,
                                                            const int* __restrict__ ids2 = nullptr,
                                                            int* __restrict__ perm2 = nullptr)
{
    // ids2 / perm2: a second payload placed like the first (pair events: the mover's old cluster travels with its point)
    // n_dev != nullptr: the number of entries is *n_dev (an event list whose length only the device knows);
    // ids != nullptr: entry i stands for point ids[i] (perm receives ids[i], not i)
    if (n_dev != nullptr) n = (long long)*n_dev;
    // need != nullptr: only the points of clusters with need[k] != 0 are placed (the exact pass will not read the
    // others' part of the permutation: screen.hip, k_cluster_need); the rest of perm[] is then stale
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate != nullptr && *gate == 0u) return; // nothing changed: the previous permutation stands (see k_hist)
    unsigned int* cnt = reinterpret_cast<unsigned int*>(smem);             // K
    unsigned long long* base = reinterpret_cast<unsigned long long*>(cnt + ((K + 1) & ~1)); // K
    // the need flags as an LDS table (a global load per point -- dependent on the point's assignment -- made this pass
    // latency bound: 0.48 ms to look at 4 B per point of 1e8 points)
    int* needl = reinterpret_cast<int*>(base + K);                                          // K
    const int tid = threadIdx.x;
    constexpr int W = VEC ? 4 : 1;
    long long per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + W - 1) / W * W;
    const long long lo = (long long)blockIdx.x * per;
    const long long hi = (lo + per < n) ? lo + per : n;
    // `per` is rounded up to a multiple of W, so trailing workgroups can start past the end: with lo > n the
    // "whole groups of four" bound below would round a NEGATIVE length down and the scalar tail would place the
    // last n % 4 points a second time.  The whole workgroup leaves together, before any barrier.
    if (lo >= hi) return;
    for (int k = tid; k < K; k += blockDim.x) { cnt[k] = 0; needl[k] = need != nullptr ? need[k] : 1; }
    __syncthreads();
    // Neighbouring points very often share a cluster (any dataset stored roughly by class, and every dataset once
    // the counting sort of the previous iteration is reflected in its order): when all active lanes of a wave
    // hold the same k, one lane adds the wave's population and the lanes take consecutive ranks.
    auto count_one = [&](int k) {
        if (!needl[k]) return;
        const unsigned long long act = __ballot(1);
        const int k0 = __builtin_amdgcn_readfirstlane(k);
        if (__ballot(k == k0) == act) {
            if (__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u)) == 0)
                atomicAdd(&cnt[k0], (unsigned)__builtin_popcountll(act));
        } else
            atomicAdd(&cnt[k], 1u);
    };
    auto place_one = [&](int k, long long i) {
        if (!needl[k]) return;
        const unsigned long long act = __ballot(1);
        const int k0 = __builtin_amdgcn_readfirstlane(k);
        unsigned int r;
        if (__ballot(k == k0) == act) {
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
            unsigned first = 0;
            if (rank == 0) first = atomicAdd(&cnt[k0], (unsigned)__builtin_popcountll(act));
            r = (unsigned)__builtin_amdgcn_readfirstlane((int)first) + rank;
        } else
            r = atomicAdd(&cnt[k], 1u);
        perm[base[k] + r] = ids != nullptr ? ids[i] : (int)i;
        if (ids2 != nullptr) perm2[base[k] + r] = ids2[i];
    };
    if constexpr (VEC) {
        const long long hv = lo + ((hi - lo) & ~3LL); // whole groups of four
        for (long long i = lo + 4LL * tid; i < hv; i += 4LL * blockDim.x) {
            const int4 a = *reinterpret_cast<const int4*>(assign + i);
            count_one(a.x); count_one(a.y); count_one(a.z); count_one(a.w);
        }
        for (long long i = hv + tid; i < hi; i += blockDim.x) count_one(assign[i]);
    } else {
        for (long long i = lo + tid; i < hi; i += blockDim.x) count_one(assign[i]);
    }
    __syncthreads();
    for (int k = tid; k < K; k += blockDim.x) {
        base[k] = cnt[k] ? atomicAdd(&cursor[k], (unsigned long long)cnt[k]) : 0ull;
        cnt[k] = 0;
    }
    __syncthreads();
    if constexpr (VEC) {
        const long long hv = lo + ((hi - lo) & ~3LL);
        for (long long i = lo + 4LL * tid; i < hv; i += 4LL * blockDim.x) {
            const int4 a = *reinterpret_cast<const int4*>(assign + i);
            place_one(a.x, i); place_one(a.y, i + 1); place_one(a.z, i + 2); place_one(a.w, i + 3);
        }
        for (long long i = hv + tid; i < hi; i += blockDim.x) place_one(assign[i], i);
    } else {
        for (long long i = lo + tid; i < hi; i += blockDim.x) place_one(assign[i], i);
    }
}

// One workgroup per item: accumulate the item's points into an LDS slab
// (sums f64[p] + counts u32[p]) with LDS atomics, then add the slab's touched rows into the
// global p x K tables with one hardware f64 atomic per touched row.  HBM-latency bound unless many
// points are in flight: a wave fetches 64 point ids (and their column bounds) with one coalesced
// load each, then walks them four at a time so that eight entry loads are outstanding per lane.
template <typename IR>
__global__ __launch_bounds__(256) void k_accumulate_sorted(const long long* __restrict__ jc,
                                                           const IR* __restrict__ ir,
                                                           const double* __restrict__ x,
                                                           const int* __restrict__ perm,
                                                           const long long* __restrict__ offs,
                                                           const int4* __restrict__ items,
                                                           const int* __restrict__ nitems, int p, int fixed_s,
                                                           double* __restrict__ sums, double* __restrict__ counts)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ssum = reinterpret_cast<double*>(smem);
    unsigned int* scnt = reinterpret_cast<unsigned int*>(ssum + p);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    for (int item = blockIdx.x; item < *nitems; item += gridDim.x) {
        const int4 it = items[item];
        const int k = it.x;
        const long long start = offs[k] + it.y;
        const int len = it.z;
        for (int r = tid; r < p; r += blockDim.x) { ssum[r] = 0.0; scnt[r] = 0u; }
        __syncthreads();
        for (int qb = wave * 64; qb < len; qb += nwaves * 64) {
            const int have = (len - qb < 64) ? len - qb : 64;
            long long my_j0 = 0;
            int my_cnt = 0;
            if (lane < have) {
                const long long i = perm[start + qb + lane];
                if (fixed_s > 0) { my_j0 = i * fixed_s; my_cnt = fixed_s; }
                else { my_j0 = jc[i]; my_cnt = (int)(jc[i + 1] - my_j0); }
            }
            for (int u = 0; u < have; u += 4) {
                long long j0[4];
                int cn[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int src = (u + v < have) ? u + v : u; // clamp: duplicates are masked by cn = 0
                    const int lo = __builtin_amdgcn_readlane((int)my_j0, src);
                    const int hi = __builtin_amdgcn_readlane((int)(my_j0 >> 32), src);
                    j0[v] = ((long long)hi << 32) | (unsigned)lo;
                    cn[v] = (u + v < have) ? __builtin_amdgcn_readlane(my_cnt, src) : 0;
                }
                double xv[4];
                int rv[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const bool ok = lane < cn[v];
                    xv[v] = ok ? x[j0[v] + lane] : 0.0;
                    rv[v] = ok ? (int)ir[j0[v] + lane] : -1;
                }
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    if (rv[v] >= 0) {
                        unsafeAtomicAdd(&ssum[rv[v]], xv[v]);
                        atomicAdd(&scnt[rv[v]], 1u);
                    }
                    // columns longer than one wave: the rest, 64 entries at a time
                    for (long long j = j0[v] + 64 + lane; j < j0[v] + cn[v]; j += 64) {
                        const int r = (int)ir[j];
                        unsafeAtomicAdd(&ssum[r], x[j]);
                        atomicAdd(&scnt[r], 1u);
                    }
                }
            }
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
}

// Incremental update of the per-cluster sums and counts (kmeans_sparsified.m:447-448's S and Cnt) by the points that
// CHANGED cluster: an event (point, key) with key = k adds the point to cluster k, key = K + k takes it out.  The events
// were sorted by key (k_hist / k_plan_segments / k_scatter_by_cluster over 2K keys), so a work item is a run of points
// that enter -- or leave -- one cluster: accumulated in an LDS slab exactly as k_accumulate_sorted does, then added to
// or subtracted from the cluster's row of the table with one f64 atomic per touched row.  Counts are integers held in
// doubles: exact under + and -.  Sums pick up one rounding per update, relative to the table entry at that time -- the
// same order of magnitude as the summation-order noise of a full pass (1e-16 relative per operation; bar: 1e-6).
// A cluster no point entered or left is not touched at all: its sums, and with them its centroid, stay bitwise the same.
// Fixed-stride shards only; rec != nullptr: the record layout (x | ir side by side), else the two arrays.
// PAIR (round 4): the events were sorted by (new, old) PAIR -- key = new (K + 1) + old, old = K for a mover without a
// valid old cluster -- so an item is a run of points that all go from one cluster to one other: its slab is added to the
// new cluster's rows and subtracted from the old one's, and every mover's record is read ONCE (two events per mover, each
// applied on its own, read it twice: 34 GB of gathers for a third of 1e8 points).
template <typename IR, bool PAIR = false>
__global__ __launch_bounds__(256) void k_accumulate_events(const char* __restrict__ rec, int R,
                                                           const IR* __restrict__ ir, const double* __restrict__ x,
                                                           const int* __restrict__ perm,
                                                           const long long* __restrict__ offs,
                                                           const int4* __restrict__ items,
                                                           const int* __restrict__ nitems, int p, int s, int K,
                                                           double* __restrict__ sums, double* __restrict__ counts)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ssum = reinterpret_cast<double*>(smem);
    unsigned int* scnt = reinterpret_cast<unsigned int*>(ssum + p);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    for (int item = blockIdx.x; item < *nitems; item += gridDim.x) {
        const int4 it = items[item];
        const int key = it.x;
        const int k = PAIR ? key / (K + 1) : (key >= K ? key - K : key);  // (PAIR: the cluster entered)
        const int kold = PAIR ? key - k * (K + 1) : -1;                   // (PAIR: the cluster left; K: none)
        const double sign = (!PAIR && key >= K) ? -1.0 : 1.0;
        const long long start = offs[key] + it.y;
        const int len = it.z;
        for (int r = tid; r < p; r += blockDim.x) { ssum[r] = 0.0; scnt[r] = 0u; }
        __syncthreads();
        for (int qb = wave * 64; qb < len; qb += nwaves * 64) {
            const int have = (len - qb < 64) ? len - qb : 64;
            int my_i = 0;
            if (lane < have) my_i = perm[start + qb + lane];
            for (int u = 0; u < have; u += 4) {
                double xv[4];
                int rv[4];
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int src = (u + v < have) ? u + v : u;
                    const long long i = (long long)(unsigned)__builtin_amdgcn_readlane(my_i, src);
                    const bool ok = (u + v < have) && lane < s;
                    if (rec != nullptr) {
                        const char* b = rec + (size_t)i * (size_t)R;
                        xv[v] = ok ? reinterpret_cast<const double*>(b)[lane] : 0.0;
                        rv[v] = ok ? (int)reinterpret_cast<const IR*>(b + (size_t)s * 8)[lane] : -1;
                    } else {
                        xv[v] = ok ? x[i * s + lane] : 0.0;
                        rv[v] = ok ? (int)ir[i * s + lane] : -1;
                    }
                }
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    if (rv[v] >= 0) {
                        unsafeAtomicAdd(&ssum[rv[v]], xv[v]);
                        atomicAdd(&scnt[rv[v]], 1u);
                    }
                    if (s > 64 && u + v < have) { // columns longer than one wave: the rest, 64 entries at a time
                        const long long i = (long long)(unsigned)__builtin_amdgcn_readlane(my_i, u + v);
                        for (int j = 64 + lane; j < s; j += 64) {
                            double xe;
                            int re;
                            if (rec != nullptr) {
                                const char* b = rec + (size_t)i * (size_t)R;
                                xe = reinterpret_cast<const double*>(b)[j];
                                re = (int)reinterpret_cast<const IR*>(b + (size_t)s * 8)[j];
                            } else { xe = x[i * s + j]; re = (int)ir[i * s + j]; }
                            unsafeAtomicAdd(&ssum[re], xe);
                            atomicAdd(&scnt[re], 1u);
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int r = tid; r < p; r += blockDim.x) {
            const unsigned int c = scnt[r];
            if (c) {
                unsafeAtomicAdd(&sums[(size_t)k * p + r], sign * ssum[r]);
                unsafeAtomicAdd(&counts[(size_t)k * p + r], sign * (double)c);
                if (PAIR && kold < K) {
                    unsafeAtomicAdd(&sums[(size_t)kold * p + r], -ssum[r]);
                    unsafeAtomicAdd(&counts[(size_t)kold * p + r], -(double)c);
                }
            }
        }
        __syncthreads();
    }
}

// Pair events, second level of the sort.  After the placement by NEW cluster (k_scatter_by_cluster over K keys; the old
// clusters travel as the second payload) the events of bucket b lie together; the work items of that first plan are
// chunks of one bucket.  k_pair_hist counts, per chunk, the old clusters in an LDS table of K + 1 bins and adds them to
// hist2[b (K + 1) + old]; k_plan_segments over those K (K + 1) keys gives every pair its range and the accumulation its
// items; k_pair_scatter walks the same chunks again and places the points: one global atomic per (chunk, old cluster) in
// each pass -- a one-level sort over 10^4 keys would need an LDS table of that size per workgroup and 10^7 of them.
__global__ __launch_bounds__(256) void k_pair_hist(const int* __restrict__ olds, const long long* __restrict__ offs1,
                                                   const int4* __restrict__ items1, const int* __restrict__ nitems1,
                                                   int K, unsigned long long* __restrict__ hist2,
                                                   const unsigned* __restrict__ gate)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate != nullptr && *gate == 0u) return;
    unsigned* cnt = reinterpret_cast<unsigned*>(smem); // K + 1
    for (int item = blockIdx.x; item < *nitems1; item += gridDim.x) {
        const int4 it = items1[item];
        const int b = it.x;
        const long long start = offs1[b] + it.y;
        const int len = it.z;
        for (int o = threadIdx.x; o <= K; o += blockDim.x) cnt[o] = 0u;
        __syncthreads();
        for (int j = threadIdx.x; j < len; j += blockDim.x) {
            const int o = olds[start + j];
            atomicAdd(&cnt[(unsigned)o < (unsigned)K ? o : K], 1u);
        }
        __syncthreads();
        for (int o = threadIdx.x; o <= K; o += blockDim.x)
            if (cnt[o]) atomicAdd(&hist2[(size_t)b * (K + 1) + o], (unsigned long long)cnt[o]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_pair_scatter(const int* __restrict__ pts, const int* __restrict__ olds,
                                                      const long long* __restrict__ offs1,
                                                      const int4* __restrict__ items1, const int* __restrict__ nitems1,
                                                      int K, unsigned long long* __restrict__ cursor2,
                                                      int* __restrict__ perm2, const unsigned* __restrict__ gate)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (gate != nullptr && *gate == 0u) return;
    unsigned* cnt = reinterpret_cast<unsigned*>(smem);                                     // K + 1
    unsigned long long* base = reinterpret_cast<unsigned long long*>(cnt + ((K + 2) & ~1)); // K + 1
    for (int item = blockIdx.x; item < *nitems1; item += gridDim.x) {
        const int4 it = items1[item];
        const int b = it.x;
        const long long start = offs1[b] + it.y;
        const int len = it.z;
        for (int o = threadIdx.x; o <= K; o += blockDim.x) cnt[o] = 0u;
        __syncthreads();
        for (int j = threadIdx.x; j < len; j += blockDim.x) {
            const int o = olds[start + j];
            atomicAdd(&cnt[(unsigned)o < (unsigned)K ? o : K], 1u);
        }
        __syncthreads();
        for (int o = threadIdx.x; o <= K; o += blockDim.x) {
            base[o] = cnt[o] ? atomicAdd(&cursor2[(size_t)b * (K + 1) + o], (unsigned long long)cnt[o]) : 0ull;
            cnt[o] = 0u;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < len; j += blockDim.x) {
            const int oo = olds[start + j];
            const int o = (unsigned)oo < (unsigned)K ? oo : K;
            const unsigned r = atomicAdd(&cnt[o], 1u);
            perm2[base[o] + r] = pts[start + j];
        }
        __syncthreads();
    }
}

// FEW events (a settled run moves some hundred points per call; api_lloyd.hip takes this form when the previous call counted
// fewer than spkm_policy::direct_events_below movers): no counting sort, no slab -- a wave per event adds the point's
// entries to (key < K) or takes them out of (key >= K) its cluster's rows of the table, one f64 atomic per entry and
// table.  Three launches (plan, placement, k_accumulate_events: 27 us of latency for 200 events) become one of 6 us.
// The events are read where k_combine_screen / k_assign_list appended them; their number is read on the device.
template <typename IR>
__global__ __launch_bounds__(256) void k_events_direct(const char* __restrict__ rec, int R, const IR* __restrict__ ir,
                                                       const double* __restrict__ x, const int* __restrict__ ev_pt,
                                                       const int* __restrict__ ev_k, const unsigned* __restrict__ n_ev,
                                                       int p, int s, int K, double* __restrict__ sums,
                                                       double* __restrict__ counts, const int* __restrict__ ev_o = nullptr)
{
    // ev_o != nullptr: pair events (one per mover: into cluster ev_k, out of cluster ev_o unless that is -1)
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long cnt = (long long)*n_ev;
    for (long long e = wave; e < cnt; e += nwaves) {
        const long long i = (long long)(unsigned)ev_pt[e];
        const int key = ev_k[e];
        const int k = (ev_o == nullptr && key >= K) ? key - K : key;
        const double sign = (ev_o == nullptr && key >= K) ? -1.0 : 1.0;
        const int kold = ev_o != nullptr ? ev_o[e] : -1;
        for (int j = lane; j < s; j += 64) {
            double xe;
            int re;
            if (rec != nullptr) {
                const char* b = rec + (size_t)i * (size_t)R;
                xe = reinterpret_cast<const double*>(b)[j];
                re = (int)reinterpret_cast<const IR*>(b + (size_t)s * 8)[j];
            } else { xe = x[i * s + j]; re = (int)ir[i * s + j]; }
            unsafeAtomicAdd(&sums[(size_t)k * p + re], sign * xe);
            unsafeAtomicAdd(&counts[(size_t)k * p + re], sign);
            if ((unsigned)kold < (unsigned)K) {
                unsafeAtomicAdd(&sums[(size_t)kold * p + re], -xe);
                unsafeAtomicAdd(&counts[(size_t)kold * p + re], -1.0);
            }
        }
    }
}

// centers(:,k) = (gamma*S(:,k)) ./ (Cnt(:,k) + 1e-16) for non-empty clusters (kmeans_sparsified.m:448);
// empty clusters keep their old column (the host applies EmptyAction, :432-445).
// blk_dff2[b] = partial sum of (old - new)^2, reduced in fixed order by k_reduce_dff.
__global__ __launch_bounds__(256) void k_finalize_centers(const double* __restrict__ sums,
                                                          const double* __restrict__ counts,
                                                          const double* __restrict__ nk_f64, int p, int K,
                                                          double gamma, double* __restrict__ centers,
                                                          double* __restrict__ blk_dff2, unsigned* __restrict__ ticket,
                                                          double* __restrict__ out, const double* __restrict__ obj2,
                                                          double* __restrict__ host_res = nullptr, unsigned long long seq = 0ull)
{
    // host_res != nullptr (spkm_lloyd_iter_host): pinned host memory, device-mapped -- [seq | dff^2 | obj^2 | nk[0..K-1]]; the
    // results go there as well, then the call's sequence number (system-scope release): the host waits for the number and
    // needs neither a copy nor a stream synchronisation to decide whether to iterate again (kmeans_sparsified.m:470-487).
    // blk_dff2[b] = this workgroup's partial sum of (old - new)^2.  The workgroup that finishes LAST (a ticket counter, reset
    // for the next call) adds the partials up in a fixed order -- lane l takes blocks l, l + 64, ...; then a shuffle tree --
    // and writes out = [dff^2, obj^2] (kmeans_sparsified.m:470-471 before sqrt): what used to be a launch of its own.
    __shared__ double s_part[4];
    __shared__ unsigned s_last;
    const size_t total = (size_t)p * K;
    double acc = 0.0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(t / p);
        const double oldv = centers[t];
        double newv = oldv;
        if (nk_f64[k] > 0.0) newv = (gamma * sums[t]) / (counts[t] + 1e-16);
        centers[t] = newv;
        const double d = oldv - newv;
        acc += d * d;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double o = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) o += s_part[w];
        blk_dff2[blockIdx.x] = o;
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || threadIdx.x >= 64) return;
    __threadfence();
    double o = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 64) o += __builtin_nontemporal_load(blk_dff2 + b);
    for (int off = 32; off > 0; off >>= 1) o += __shfl_down(o, off);
    if (threadIdx.x == 0) {
        out[0] = o;
        if (obj2) out[1] = *obj2;
        *ticket = 0u;
    }
    if (host_res != nullptr) { // (one wave: its stores are complete behind the fence, whichever lane made them)
        for (int k = threadIdx.x; k < K; k += 64) host_res[3 + k] = nk_f64[k];
        if (threadIdx.x == 0) {
            host_res[1] = o;
            host_res[2] = obj2 ? *obj2 : __longlong_as_double(0x7ff8000000000000LL);
        }
        __threadfence_system();
        if (threadIdx.x == 0)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_res), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// nk (u64 counters) -> f64 slots of the reduce buffer, so that one SUM all-reduce covers everything.
__global__ void k_nk_to_f64(const unsigned long long* __restrict__ nk, int K, double* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) out[k] = (double)nk[k];
}

// ... together with the other small hand-overs at the end of the fused call (each was a launch of its own): obj^2 into
// the reduce buffer, the statistics and cluster sizes into the caller's buffers (either may be null)
#define SPKM_REPORT_WORDS 24 // counters a fused call reports to the host ([19]: the device opened the full pass, k_pick_form)
__global__ void k_call_tail(const unsigned long long* __restrict__ nk, int K, double* __restrict__ nk_f,
                            const double* __restrict__ stats, double* __restrict__ obj2, double* __restrict__ d_stats,
                            unsigned long long* __restrict__ d_nk, const unsigned* __restrict__ bstat, int bstat_n,
                            unsigned* __restrict__ counters, int lazy = 0, double* __restrict__ cache_s = nullptr,
                            const double* __restrict__ cache_c = nullptr, size_t pk = 0, double* __restrict__ sums = nullptr,
                            double* __restrict__ counts = nullptr, unsigned* __restrict__ host_out = nullptr, unsigned seq = 0u,
                            unsigned long long work_steps = 0ull, int work_tiles = 0, int work_nr = 0, int work_a = 0,
                            int work_flags = 0)
{
    // work_*: what this call's 4-lanes-per-point screen launch did, in ROUNDS (4 stored entries of 16 points against the
    // centroids of one tile) -- counters[34..35] += rounds executed for all centroids of a tile, counters[36..37] += rounds of
    // a launch that does all the work (work_steps = ceil(n / 16) steps x work_tiles x work_nr); running totals, read by
    // spkm_screen_work_totals (bench.py weights the window's algorithmic bytes by their ratio).  work_flags: 1 = the
    // unconditional two-phase form (every step stops after work_a rounds), 2 = the launch ran over a list (counters[4]
    // entries), 4 = of points; the hinted form's early-finished (step, tile) pairs are counters[2].
    if (work_tiles > 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long steps = work_steps;
        if (work_flags & 2) steps = (work_flags & 4) ? ((unsigned long long)counters[4] + 15ull) / 16ull : (unsigned long long)counters[4];
        unsigned long long r = steps * (unsigned long long)work_tiles * (unsigned long long)((work_flags & 1) ? work_a : work_nr);
        if (!(work_flags & 1) && work_a < work_nr) {
            const unsigned long long saved = (unsigned long long)counters[2] * (unsigned long long)(work_nr - work_a);
            r -= saved < r ? saved : r;
        }
        *reinterpret_cast<unsigned long long*>(counters + 34) += r;
        *reinterpret_cast<unsigned long long*>(counters + 36) += work_steps * (unsigned long long)work_tiles * (unsigned long long)work_nr;
    }
    // host_out != nullptr: pinned host memory, device-mapped -- the call's first SPKM_REPORT_WORDS counters for the host policy go there, then
    // the report's number `seq` with a system-scope release (api_lloyd.hip reads them one call later, if the number is there)
    // cache_s != nullptr (incremental calls): the call's sums and counts ARE the cache.  One repair on the way: a row of a
    // cluster that no member stores any more (count 0) must have the sum EXACTLY 0 -- a fresh summation gives that, an
    // add-and-subtract history leaves a residual of rounding noise, and kmeans_sparsified.m:448 divides it by 1e-16.
    if (cache_s != nullptr) {
        for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < pk; t += (size_t)gridDim.x * blockDim.x) {
            const double c = cache_c[t];
            double v = cache_s[t];
            if (c == 0.0 && v != 0.0) { v = 0.0; cache_s[t] = 0.0; }
            sums[t] = v;
            counts[t] = c;
        }
    }
    // lazy != 0: this call did not evaluate the objective, the largest distance and its index (spkm_shard_set_lazy_stats):
    // NaN in their places, so that a caller that reads them anyway cannot mistake them for values
    // bstat: (points kept, steps skipped) per workgroup of k_bounds_steps -> counters[12], counters[3] and the running
    // total at counters[8..9] (read by the host one call later)
    if (blockIdx.x == 0 && bstat_n > 0) {
        __shared__ unsigned s_k[256], s_s[256];
        unsigned a = 0, b = 0;
        for (int t = threadIdx.x; t < bstat_n; t += blockDim.x) { a += bstat[2 * t]; b += bstat[2 * t + 1]; }
        s_k[threadIdx.x] = a; s_s[threadIdx.x] = b;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) { s_k[threadIdx.x] += s_k[threadIdx.x + off]; s_s[threadIdx.x] += s_s[threadIdx.x + off]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            counters[12] += s_k[0];
            counters[3] += s_s[0];
            *reinterpret_cast<unsigned long long*>(counters + 8) += (unsigned long long)s_s[0];
        }
    }
    if (host_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { // (thread 0 made the last changes to the counters itself)
        for (int j = 0; j < SPKM_REPORT_WORDS; j++) host_out[j] = counters[j];
        __hip_atomic_store(host_out + SPKM_REPORT_WORDS, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) {
        const unsigned long long v = nk[k];
        nk_f[k] = (double)v;
        if (d_nk) d_nk[k] = v;
    }
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (k == 0) *obj2 = lazy ? nan : stats[0];
    if (k < 3 && d_stats) d_stats[k] = lazy ? nan : stats[k];
}

template __global__ void k_accumulate_atomic<unsigned short>(const long long*, const unsigned short*, const double*,
    const int*, int, long long, double*, double*);
template __global__ void k_accumulate_atomic<unsigned int>(const long long*, const unsigned int*, const double*,
    const int*, int, long long, double*, double*);
template __global__ void k_accumulate_sorted<unsigned short>(const long long*, const unsigned short*, const double*,
    const int*, const long long*, const int4*, const int*, int, int, double*, double*);
template __global__ void k_accumulate_sorted<unsigned int>(const long long*, const unsigned int*, const double*,
    const int*, const long long*, const int4*, const int*, int, int, double*, double*);
