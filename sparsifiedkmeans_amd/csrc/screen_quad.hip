// The 4-lanes-per-point f32 screen kernel (k_screen_quad) -- see the header of screen.hip for what it certifies.
// Its own translation unit, compiled once per (row-id width, list granularity): the 16 round counts x 3 forms x 4 tile
// widths of hand-scheduled rounds are most of the library's build time, and four TUs build in parallel.
//   -DSPKM_SQ_IRBITS=16|32   row ids of the screen copy
//   -DSPKM_SQ_PTS=0|1        the list names 16-point steps / points
// Each TU exports one dispatcher, spkm_sq_kernel_<bits>_<pts>(rounds, a_rounds) -> kernel (host stub address).
#include "common.h"
#include "quad_steps.inc"
#include <type_traits>

#ifndef SPKM_SQ_IRBITS
#error "build with -DSPKM_SQ_IRBITS=16|32 -DSPKM_SQ_PTS=0|1 (csrc/build.sh)"
#endif

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

// ============================================================================================
// Second screen kernel: 4 lanes per point, 16 points per wave, 8 centroids per lane.
//
// Same tile as k_screen_tile (32 centroids, 128-B rows).  Lane (point slot ps, l4) keeps entries l4, l4+4, ...
// of its point in registers; at step j the owner's value and row offset are broadcast inside the quad
// (quad_perm DPP, one instruction each) and every lane reads its 2 x 16 B of the row:
//     xs   = quad_bcast(x)                     v_mov_b32_dpp
//     addr = quad_bcast(rowoff) + laneoff      v_add_u32_dpp
//     q0, q1 = LDS[addr], LDS[addr + 64]       ds_read_b128 x2
//     acc[0..3] += (q + xs)^2                  v_pk_add_f32 x4, v_pk_fma_f32 x4
// 10 issue slots per entry for 16 points x 32 centroids (the 16-lane layout above: 4 slots for 4 points),
// no cross-lane reduction of the sums (each lane owns its 8 centroids), a 2-stage quad min for the winner.
__device__ __forceinline__ float raw_min_f32(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float raw_max_f32(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int quad_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false)); // quad_perm:[1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false)); // quad_perm:[2,3,0,1]
    return v;
}
__device__ __forceinline__ float quad_sum_f32(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    return v;
}
template <int SEL> __device__ __forceinline__ int quad_bcast_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, SEL * 0x55, 0xf, 0xf, false); // quad_perm:[SEL,SEL,SEL,SEL]
}
__device__ __forceinline__ float quad_min_f32(float v)
{
    asm("s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}

template <int NR, typename IR, int PL0, int A, bool PTS>
__device__ __forceinline__ void screen_quad_body(const IR* __restrict__ ir, const float* __restrict__ xval, int p, int n,
                                                 int nv, int fixed_s, int K, const spkm_blockmap bm, int chunk_points,
                                                 float* __restrict__ m1o, float* __restrict__ m2o,
                                                 int* __restrict__ ko, char* smem, unsigned* ticket, int extra_base,
                                                 int extra_k0,
                                                 const float* __restrict__ hint, float hint_c,
                                                 unsigned* __restrict__ counters, const int* __restrict__ todo,
                                                 int todo_pts, const char* __restrict__ rec, int rec_R,
                                                 const int* __restrict__ recmap)
{
    // recmap != nullptr (a regrouped shard, api_lloyd.hip): point q of this shard's order is record recmap[q]
    // PTS (its own kernel instantiation): the list names POINTS; rec != nullptr: their entries are read from the record
    // layout of the exact pass (k_build_records: one point = R contiguous bytes, f64 values then row ids) instead of the
    // step-major f32 copy, where the entries of ONE point are 13 pieces of 16 + 8 B in 26 different cache lines
    constexpr int PPS = 16;
    const int lane = threadIdx.x & 63;
    const int ps = lane >> 2, l4 = lane & 3;
    const int k0 = bm.tile * SCREEN_KT;
    // The two point pairs of a 16-lane LDS phase start on opposite halves of their rows (PL = 4), or read
    // different copies of a narrow tile's row (PL < 4).
    const bool swp = (ps & 2) != 0;
    // PL = 5: the lane's extra centroid extra_k0 + l4 sits in a table of 16-B rows at extra_base:
    // its address is (a >> 7) * 16 + ce for a = row * 128 + ...
    const int ce = extra_base + l4 * 4;
    // ROTATING REMAINDER (PL0 = 5 with rot > 0, build_blockmap_quad's team layout): the <= 4 centroids that have no tile
    // of their own are carried by the rot tiles IN TURN -- tile g carries them for the chunks ci of its team with
    // ci % rot == g -- so that every tile costs the same per chunk: the workgroups of a team (one per tile, same XCD)
    // stay in lock-step, a chunk is fetched from HBM once and met in L2 by the others, and the clock that this
    // power-bound kernel sustains goes up with the halved traffic.  A step is then evaluated by the PL = 5 code when
    // this tile carries its chunk's remainder and by the PL = 4 code otherwise.
    const int rot = PL0 == 5 ? (bm.pad >> 24) & 0xff : 0;
    // todo != nullptr: only the steps listed there are processed (k_bounds_steps: the others were skipped on the
    // carried bounds); tickets and chunks then number the LIST -- nv = 16 x its length stands in for n
    const int nchunks = (nv + chunk_points - 1) / chunk_points;
    const int R = chunk_points / PPS;
    // chunk ids of this workgroup: (stream + ci * nstreams) * mul + add   (mul/add: XCD-local numbering)
    const int mul = bm.pad & 0xff, add = (bm.pad >> 8) & 0xff;
    const int local_chunks = nchunks > add ? (nchunks - add + mul - 1) / mul : 0;
    const int my_chunks = (local_chunks > bm.stream) ? (local_chunks - bm.stream + bm.nstreams - 1) / bm.nstreams : 0;
    const int Tn = my_chunks * R;
    auto draw = [&]() {
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(ticket, 1u);
        return (int)__builtin_amdgcn_readfirstlane(v);
    };
    auto point_of = [&](int t) {
        if (t >= Tn) return nv;
        const int ci = t / R, rr = t - ci * R;
        const int base = ((bm.stream + ci * bm.nstreams) * mul + add) * chunk_points + rr * PPS;
        return (base < 0 || base > nv) ? nv : base;
    };
    // a step's entries: NR rounds of 4 (entries past the column become x = 0 on the zero row p).  No software
    // prefetch across steps: the other three waves of the SIMD cover the load latency.
    const int nvl = fixed_s - 4 * (NR - 1);
    unsigned npruned = 0; // steps this wave finished in the hinted two-phase form
    for (int t = draw(); t < Tn; t = draw()) {
        const int vbase = point_of(t);
        if (vbase < nv) {
            // step-major screen copy (k_screen_reorder): round r of a step is 64 consecutive elements, element
            // 4 * slot + l4 belongs to the step's point `slot`
            int i;
            const float* xp = nullptr;
            const IR* rp = nullptr;
            const double* xd = nullptr; // record mode: the point's values / row ids, entry e at xd[e] / rd[e]
            const IR* rd = nullptr;
            if constexpr (PTS) {
                // the list names POINTS (k_bounds_steps, point mode): 16 unrelated points share this wave's step, each
                // quad fetches its own point's elements (16 B per quad and round instead of one 256-B row per wave --
                // affordable only while few points are listed, which is when the host selects this mode)
                const int slot = vbase + ps;
                const int q = slot < todo_pts ? todo[slot] : n;       // n: an empty slot (no output)
                const int qc = q < n ? q : n - 1;
                i = q;
                if (rec != nullptr) {
                    const char* rb = rec + (size_t)(recmap != nullptr ? recmap[qc] : qc) * (size_t)rec_R;
                    xd = reinterpret_cast<const double*>(rb) + l4;
                    rd = reinterpret_cast<const IR*>(rb + (size_t)fixed_s * 8) + l4;
                } else {
                    xp = xval + (size_t)(qc >> 4) * (NR * 64) + ((qc & 15) << 2) + l4;
                    rp = ir + (size_t)(qc >> 4) * (NR * 64) + ((qc & 15) << 2) + l4;
                }
            } else {
                const int base = todo != nullptr ? todo[vbase >> 4] << 4 : vbase;
                i = base + ps;
                xp = xval + (size_t)(base >> 4) * (NR * 64) + lane;
                rp = ir + (size_t)(base >> 4) * (NR * 64) + lane;
            }
            // the hint is needed only after the first evaluation, but its load must not wait until then (a second
            // exposed memory latency per step): issued first, pinned in a register before the rounds
            float hraw = 0.f;
            if (A < NR && hint != nullptr) hraw = hint[i < n ? i : n - 1];
            // scalars, not arrays: the compiler turns constant-indexed arrays into 16-wide register tuples and spills them
#define SPKM_QUAD_DECL(r) float x##r = 0.f; int o##r = 0;
            SPKM_QUAD_DECL(0) SPKM_QUAD_DECL(1) SPKM_QUAD_DECL(2) SPKM_QUAD_DECL(3) SPKM_QUAD_DECL(4) SPKM_QUAD_DECL(5)
            SPKM_QUAD_DECL(6) SPKM_QUAD_DECL(7) SPKM_QUAD_DECL(8) SPKM_QUAD_DECL(9) SPKM_QUAD_DECL(10)
            SPKM_QUAD_DECL(11) SPKM_QUAD_DECL(12) SPKM_QUAD_DECL(13) SPKM_QUAD_DECL(14) SPKM_QUAD_DECL(15)
#undef SPKM_QUAD_DECL
#define SPKM_QUAD_LOAD(r)                                                              \
    if constexpr (NR > r) { x##r = xp[r * 64]; o##r = (int)rp[r * 64]; }
            // record mode: entry 4 r + l4 of the point, converted and encoded as k_screen_reorder does (x~ = fl32(x); row id
            // -> row * 8 ^ swizzle; slots past the column: x = 0 on the all-zero row p).  Storage order, not partitioned by
            // row parity: more LDS bank conflicts per round, in a mode that is bound by its gathers
#define SPKM_QUAD_LOAD_REC(r)                                                          \
    if constexpr (NR > r) {                                                            \
        const bool okr = (r < NR - 1) || (4 * r + l4 < fixed_s);                       \
        const double xv_ = okr ? xd[4 * r] : 0.0;                                      \
        const unsigned row_ = okr ? (unsigned)rd[4 * r] : (unsigned)p;                 \
        x##r = (float)xv_;                                                             \
        o##r = (int)((row_ << 3) ^ ((row_ >> 1) & 3u));                                \
    }
            if (PTS && rec != nullptr) {
                SPKM_QUAD_LOAD_REC(0) SPKM_QUAD_LOAD_REC(1) SPKM_QUAD_LOAD_REC(2) SPKM_QUAD_LOAD_REC(3) SPKM_QUAD_LOAD_REC(4)
                SPKM_QUAD_LOAD_REC(5) SPKM_QUAD_LOAD_REC(6) SPKM_QUAD_LOAD_REC(7) SPKM_QUAD_LOAD_REC(8) SPKM_QUAD_LOAD_REC(9)
                SPKM_QUAD_LOAD_REC(10) SPKM_QUAD_LOAD_REC(11) SPKM_QUAD_LOAD_REC(12) SPKM_QUAD_LOAD_REC(13)
                SPKM_QUAD_LOAD_REC(14) SPKM_QUAD_LOAD_REC(15)
            } else {
                SPKM_QUAD_LOAD(0) SPKM_QUAD_LOAD(1) SPKM_QUAD_LOAD(2) SPKM_QUAD_LOAD(3) SPKM_QUAD_LOAD(4) SPKM_QUAD_LOAD(5)
                SPKM_QUAD_LOAD(6) SPKM_QUAD_LOAD(7) SPKM_QUAD_LOAD(8) SPKM_QUAD_LOAD(9) SPKM_QUAD_LOAD(10)
                SPKM_QUAD_LOAD(11) SPKM_QUAD_LOAD(12) SPKM_QUAD_LOAD(13) SPKM_QUAD_LOAD(14) SPKM_QUAD_LOAD(15)
            }
#undef SPKM_QUAD_LOAD
#undef SPKM_QUAD_LOAD_REC
            auto evaluate_step = [&](auto pl_tag) {
            constexpr int PL = decltype(pl_tag)::value;
            // The two point pairs of a 16-lane LDS phase start on opposite halves of their rows (PL = 4), or read
            // different copies of a narrow tile's row (PL < 4).
            const int off0 = l4 * (PL == 1 ? 8 : 16) + (swp ? 64 : 0), off1 = l4 * 16 + (swp ? 0 : 64);
            const bool tile_full = (PL >= 4 ? k0 + SCREEN_KT <= K : k0 + 8 * PL <= K) && (PL != 5 || extra_k0 + 4 <= K);
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0; // two f32 sums each (bit pattern 0 = (0.f, 0.f))
            float acc4 = 0.f;                                       // PL = 5: the lane's extra centroid
            // the last round broadcasts only the nvl = fixed_s - 4 (NR - 1) entries the column still has
#define SPKM_GUARD_A(r) ((r) < A)
#define SPKM_GUARD_B(r) ((r) >= A)
#define SPKM_QUAD_ROUND_G(r, COND)                                                                          \
    if constexpr (NR > r && COND(r)) {                                                                      \
        const int xi = __builtin_bit_cast(int, x##r);                                                       \
        const int ro = o##r << 4; /* stored: row * 8 ^ swizzle -> row * 128 | swizzle * 16 */              \
        if (r < NR - 1 || nvl == 4) quad_round<4, PL>(xi, ro, off0, off1 - off0, ce, acc0, acc1, acc2, acc3, acc4);        \
        else if (nvl == 3) quad_round<3, PL>(xi, ro, off0, off1 - off0, ce, acc0, acc1, acc2, acc3, acc4);            \
        else if (nvl == 2) quad_round<2, PL>(xi, ro, off0, off1 - off0, ce, acc0, acc1, acc2, acc3, acc4);            \
        else quad_round<1, PL>(xi, ro, off0, off1 - off0, ce, acc0, acc1, acc2, acc3, acc4);                          \
    }
#define SPKM_QUAD_ROUNDS(COND)                                                                              \
    SPKM_QUAD_ROUND_G(0, COND) SPKM_QUAD_ROUND_G(1, COND) SPKM_QUAD_ROUND_G(2, COND) SPKM_QUAD_ROUND_G(3, COND)   \
    SPKM_QUAD_ROUND_G(4, COND) SPKM_QUAD_ROUND_G(5, COND) SPKM_QUAD_ROUND_G(6, COND) SPKM_QUAD_ROUND_G(7, COND)   \
    SPKM_QUAD_ROUND_G(8, COND) SPKM_QUAD_ROUND_G(9, COND) SPKM_QUAD_ROUND_G(10, COND) SPKM_QUAD_ROUND_G(11, COND) \
    SPKM_QUAD_ROUND_G(12, COND) SPKM_QUAD_ROUND_G(13, COND) SPKM_QUAD_ROUND_G(14, COND) SPKM_QUAD_ROUND_G(15, COND)
            asm volatile("" : "+v"(hraw));
            SPKM_QUAD_ROUNDS(SPKM_GUARD_A)
            // lane's centroids.  PL = 4: first read -> k0 + off0/4 + 0..3, second read -> k0 + off1/4 + 0..3;
            // PL < 4: k0 + 2 PL l4 + 0 .. 2 PL - 1 (either copy)
            // branch-free smallest / second smallest / argmin over the lane's values (ascending k, first wins).
            // Raw v_min / v_max: the compiler's fminf / fmaxf add a canonicalising v_max per operand.  A NaN
            // estimate never wins (v < lo is false) and drags `hi` down to `lo`: the point goes to the list.
            float lo, hi, m1, m2;
            int klo, first;
            unsigned seg;
            constexpr int NPAIR = PL == 5 ? 4 : PL;
            auto evaluate = [&]() {
                const f2v acc[4] = {__builtin_bit_cast(f2v, acc0), __builtin_bit_cast(f2v, acc1),
                                    __builtin_bit_cast(f2v, acc2), __builtin_bit_cast(f2v, acc3)};
                lo = __builtin_inff();
                hi = __builtin_inff();
                klo = -1;
                auto consider = [&](float v, int k) {
                    const bool less = v < lo;
                    hi = raw_min_f32(hi, raw_max_f32(lo, v));
                    klo = less ? k : klo;
                    lo = less ? v : lo;
                };
                if (tile_full) { // every slot of this tile is a real centroid (uniform): no masking
#pragma unroll
                    for (int a = 0; a < NPAIR; a++) {
#pragma unroll
                        for (int h = 0; h < 2; h++)
                            consider(h ? acc[a].y : acc[a].x, PL >= 4 ? k0 + ((a < 2 ? off0 : off1) >> 2) + 2 * (a & 1) + h
                                                                      : k0 + 2 * PL * l4 + 2 * a + h);
                    }
                    if (PL == 5) consider(acc4, extra_k0 + l4);
                } else {
#pragma unroll
                    for (int a = 0; a < NPAIR; a++) {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int k = PL >= 4 ? k0 + ((a < 2 ? off0 : off1) >> 2) + 2 * (a & 1) + h
                                                  : k0 + 2 * PL * l4 + 2 * a + h;
                            const float v = h ? acc[a].y : acc[a].x;
                            consider((k < K) ? v : __builtin_inff(), k);
                        }
                    }
                    if (PL == 5) {
                        const int k = extra_k0 + l4;
                        consider((k < K) ? acc4 : __builtin_inff(), k);
                    }
                }
                m1 = quad_min_f32(lo);
                const bool win = (lo == m1);
                seg = (unsigned)(__ballot(win) >> (ps * 4)) & 0xfu;
                first = seg ? __builtin_ctz(seg) : 0;
                m2 = quad_min_f32((l4 == first) ? hi : lo);
            };
            evaluate();
            // Hinted two-phase screen: `hint` holds, per point, an estimate of its distance to the centroid it had in
            // the previous call (from the library's carried bound and that centroid's drift, k_bounds_steps).  If, for
            // every point of this step, all non-leading centroids of the tile are already (by their partial sums)
            // more than sqrt(hint_c) times that far away, the step is finished for the leaders only; otherwise the
            // remaining rounds are run for all centroids.  The hint steers the work, never a result.
            int a_eff = A;
            if (A < NR && hint != nullptr) {
                const float hv = hraw;
                // (stale hints -- the first iterations of a run, a reused buffer -- send steps to the exact list; the host
                // sees the count one call later and pauses the hints, api_lloyd.hip)
                const bool fine = !(i < n) || m2 >= hint_c * hv * hv; // false for NaN
                if (!__all(fine)) {
                    SPKM_QUAD_ROUNDS(SPKM_GUARD_B)
                    evaluate();
                    a_eff = NR;
                } else
                    npruned++;
            }
            // Phase B (A < NR): the sums above cover only the first 4 A entries of each column.  They are
            // LOWER bounds of the full sums (every term is >= 0 and f32 addition is monotone), which is all the
            // certificate needs for the centroids that lose; only the tile's leader by partial sum is finished:
            // each lane adds ITS OWN remaining entries (no broadcast) for that one centroid, the quad adds up.
            //   m1 = full estimate of the leader, m2 = smallest partial sum among the others (<= their full sums)
            float full = m1;
            if (a_eff < NR) {
                const int kwin = quad_min_i32((l4 == first && seg != 0u) ? klo : 0x7fffffff);
                const bool is_extra = PL == 5 && kwin >= extra_k0;
                // tile: float c = kwin - k0 of the row sits in 16-B piece (c >> 2) ^ swizzle(row), and the stored row id
                // (o << 4 = row * 128 | swizzle * 16) carries the swizzle: address = ((o << 4) ^ piece * 16) + (c & 3) * 4.
                // The 16 points of a step mostly share their leader: without the swizzle all 64 lanes of such a read
                // would hit the two banks (row parity) of one column.  Extra table (rare leader): 16-B rows, no swizzle.
                const int cpiece = ((kwin - k0) >> 2) << 4, celem = ((kwin - k0) & 3) << 2;
                const int ebase = extra_base + (kwin - extra_k0) * 4;
                float accb = 0.f;
#define SPKM_QUAD_FINISH(r, EXTRA)                                                                          \
    if constexpr (NR > r && r >= A) {                                                                       \
        const bool okr = (r < NR - 1) || l4 < nvl;                                                          \
        const int t4 = o##r << 4;                                                                           \
        const int adr = (EXTRA && is_extra) ? ((t4 >> 7) << 4) + ebase : (t4 ^ cpiece) + celem;             \
        const float cv = *reinterpret_cast<const float*>(smem + adr);                                       \
        const float tv = okr ? cv + x##r : 0.f;                                                             \
        accb = __builtin_fmaf(tv, tv, accb);                                                                \
    }
#define SPKM_QUAD_FINISH_ALL(EXTRA)                                                                         \
    SPKM_QUAD_FINISH(0, EXTRA) SPKM_QUAD_FINISH(1, EXTRA) SPKM_QUAD_FINISH(2, EXTRA) SPKM_QUAD_FINISH(3, EXTRA)     \
    SPKM_QUAD_FINISH(4, EXTRA) SPKM_QUAD_FINISH(5, EXTRA) SPKM_QUAD_FINISH(6, EXTRA) SPKM_QUAD_FINISH(7, EXTRA)     \
    SPKM_QUAD_FINISH(8, EXTRA) SPKM_QUAD_FINISH(9, EXTRA) SPKM_QUAD_FINISH(10, EXTRA) SPKM_QUAD_FINISH(11, EXTRA)   \
    SPKM_QUAD_FINISH(12, EXTRA) SPKM_QUAD_FINISH(13, EXTRA) SPKM_QUAD_FINISH(14, EXTRA) SPKM_QUAD_FINISH(15, EXTRA)
                if (PL == 5 && __any(is_extra)) { SPKM_QUAD_FINISH_ALL(true) } else { SPKM_QUAD_FINISH_ALL(false) }
#undef SPKM_QUAD_FINISH_ALL
#undef SPKM_QUAD_FINISH
#undef SPKM_QUAD_ROUNDS
#undef SPKM_QUAD_ROUND_G
#undef SPKM_GUARD_A
#undef SPKM_GUARD_B
                full = m1 + quad_sum_f32(kwin == 0x7fffffff ? 0.f : accb);
            }
            if (l4 == first && i < n) {
                const bool none = seg == 0u;
                // (point lists: the results are stored by LIST SLOT, not by point id -- the screen writes and
                //  k_combine_screen reads them contiguously instead of at 2 % random places of three n-sized arrays)
                const int at = PTS ? vbase + ps : i;
                m1o[at] = none ? __builtin_inff() : full;
                m2o[at] = none ? __builtin_inff() : m2;
                ko[at] = none ? -1 : klo;
            }
            }; // evaluate_step
            if constexpr (PL0 == 5) {
                // (t / R = the team's chunk counter: the same for the workgroups of a team, whatever the tile)
                if (rot > 0 && (t / R) % rot != bm.tile) evaluate_step(std::integral_constant<int, 4>{});
                else evaluate_step(std::integral_constant<int, 5>{});
            } else
                evaluate_step(std::integral_constant<int, PL0>{});
        }
    }
    // (per workgroup, not per wave: 4096 waves finishing together queued 4096 atomics on one address -- ~50 us at the
    //  end of every launch; the kernel adds ticket[1] to counters[2] behind its last barrier)
    if (counters != nullptr && lane == 0 && npruned) atomicAdd(ticket + 1, npruned);
}

template <int NR, typename IR, int TWO, bool PTS> // TWO: 0 all rounds for all centroids, 1 split at quad_split(NR, PTS), 2 at quad_split_late(NR, PTS); PTS: the list names points
__global__ __launch_bounds__(1024) void k_screen_quad(
    const IR* __restrict__ ir, const float* __restrict__ xval, const float* __restrict__ T32, int p, int n, int fixed_s,
    int K, const spkm_blockmap* __restrict__ bmap, int chunk_points, float* __restrict__ scr_m1,
    float* __restrict__ scr_m2, int* __restrict__ scr_k, int extra_tile,
    const float* __restrict__ hint, float hint_c, unsigned* __restrict__ counters, const int* __restrict__ todo,
    int todo_points, // todo_points != 0 (the PTS instantiation): the list holds point ids (counters[4] of them), not 16-point steps
    const char* __restrict__ rec, int rec_R, // record layout for the listed points (PTS; may be null)
    const int* __restrict__ recmap)          // ... and which record a point of this shard's order is (null: its own)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const spkm_blockmap bm = bmap[blockIdx.x];
    if (bm.tile < 0) return;
    if (todo != nullptr && counters[4] == 0u) return; // an empty list: not even the tile is loaded
    const int tid = threadIdx.x;
    const int pl = (bm.pad >> 16) & 0xff; // centroid pairs per lane in this workgroup's tile (5: 4 + one extra centroid)
    const size_t tile_bytes = (size_t)(p + 1) * SCREEN_KT * 4;
    const size_t extra_bytes = pl == 5 ? (size_t)(p + 1) * 16 : 0;
    {
        const float4* src = reinterpret_cast<const float4*>(T32 + (size_t)bm.tile * (p + 1) * SCREEN_KT);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (size_t t = tid; t < tile_bytes / 16; t += blockDim.x) dst[t] = src[t];
        if (pl == 5) {
            const float4* esrc = reinterpret_cast<const float4*>(T32 + (size_t)extra_tile * (p + 1) * SCREEN_KT);
            float4* edst = reinterpret_cast<float4*>(smem + tile_bytes);
            for (size_t t = tid; t < extra_bytes / 16; t += blockDim.x) edst[t] = esrc[t];
        }
    }
    unsigned* ticket = reinterpret_cast<unsigned*>(smem + tile_bytes + extra_bytes);
    if (tid == 0) { ticket[0] = 0u; ticket[1] = 0u; }
    __syncthreads();
    float* m1o = scr_m1 + (size_t)bm.tile * n;
    float* m2o = scr_m2 + (size_t)bm.tile * n;
    int* ko = scr_k + (size_t)bm.tile * n;
    const int eb = (int)tile_bytes, ek = extra_tile * SCREEN_KT;
    constexpr int A = TWO == 0 ? NR : (TWO == 2 && quad_split_late(NR, PTS) > 0 ? quad_split_late(NR, PTS) : quad_split(NR, PTS));
    int nv = n, chunk_v = chunk_points, tp = 0;
    if (todo != nullptr) { // counters[4] = length of the list; chunks small enough that every workgroup gets several
        if (PTS) { tp = (int)counters[4]; nv = (tp + 15) & ~15; }
        else nv = (int)counters[4] * 16;
        chunk_v = max(256, min(chunk_points, (nv / (int)(gridDim.x * 2)) & ~255));
    }
    if (pl == 4) screen_quad_body<NR, IR, 4, A, PTS>(ir, xval, p, n, nv, fixed_s, K, bm, chunk_v, m1o, m2o, ko, smem, ticket, eb, ek, hint, hint_c, counters, todo, tp, rec, rec_R, recmap);
    else if (pl == 5) screen_quad_body<NR, IR, 5, A, PTS>(ir, xval, p, n, nv, fixed_s, K, bm, chunk_v, m1o, m2o, ko, smem, ticket, eb, ek, hint, hint_c, counters, todo, tp, rec, rec_R, recmap);
    else if (pl == 2) screen_quad_body<NR, IR, 2, A, PTS>(ir, xval, p, n, nv, fixed_s, K, bm, chunk_v, m1o, m2o, ko, smem, ticket, eb, ek, hint, hint_c, counters, todo, tp, rec, rec_R, recmap);
    else screen_quad_body<NR, IR, 1, A, PTS>(ir, xval, p, n, nv, fixed_s, K, bm, chunk_v, m1o, m2o, ko, smem, ticket, eb, ek, hint, hint_c, counters, todo, tp, rec, rec_R, recmap);
    if (counters != nullptr) {
        __syncthreads();
        if (tid == 0 && ticket[1]) atomicAdd(counters + 2, ticket[1]);
    }
}


#if SPKM_SQ_IRBITS == 16
typedef unsigned short sq_ir_t;
#else
typedef unsigned int sq_ir_t;
#endif
#define SPKM_SQ_CAT2(a, b, c) a##b##_##c
#define SPKM_SQ_CAT(a, b, c) SPKM_SQ_CAT2(a, b, c)

// one kernel per round count (a switch inside one kernel makes the register allocator spill)
// a_rounds: rounds evaluated for all centroids (>= rounds: the plain form; else one of the two compiled splits)
const void* SPKM_SQ_CAT(spkm_sq_kernel_, SPKM_SQ_IRBITS, SPKM_SQ_PTS)(int rounds, int a_rounds)
{
    constexpr bool PTS = SPKM_SQ_PTS != 0;
    typedef sq_ir_t IR;
    const bool late = a_rounds < rounds && quad_split_late(rounds, PTS) > 0 && a_rounds == quad_split_late(rounds, PTS);
    const bool two = a_rounds < rounds;
    switch (rounds) {
#define SPKM_QUAD_CASE(N)                                                                                   \
    case N:                                                                                                 \
        if (late) return (const void*)k_screen_quad<N, IR, (quad_split_late(N, PTS) > 0 ? 2 : 0), PTS>;     \
        return two && quad_split(N, PTS) < N ? (const void*)k_screen_quad<N, IR, (quad_split(N, PTS) < N ? 1 : 0), PTS> : (const void*)k_screen_quad<N, IR, 0, PTS>;
        SPKM_QUAD_CASE(1) SPKM_QUAD_CASE(2) SPKM_QUAD_CASE(3) SPKM_QUAD_CASE(4) SPKM_QUAD_CASE(5) SPKM_QUAD_CASE(6)
        SPKM_QUAD_CASE(7) SPKM_QUAD_CASE(8) SPKM_QUAD_CASE(9) SPKM_QUAD_CASE(10) SPKM_QUAD_CASE(11) SPKM_QUAD_CASE(12)
        SPKM_QUAD_CASE(13) SPKM_QUAD_CASE(14) SPKM_QUAD_CASE(15) SPKM_QUAD_CASE(16)
#undef SPKM_QUAD_CASE
    default: return nullptr;
    }
}
