// libspkm.so -- C ABI implementation (include/spkm.h), part 2 of 2: the device-resident Lloyd engine -- shard layouts,
// exact assignment and accumulation, the fused call with its certified screen (run_screen), distances on demand,
// finalisation, k-means++ helpers, the dense two-pass outputs.  Unity build of the kernels it launches.
#include "assign.hip"
#include "update.hip"
#include "screen.hip"
#include "dense.hip"

#include "api_internal.h"

// k_screen_quad lives in its own translation units (screen_quad.hip, one per row-id width and list granularity)
const void* spkm_sq_kernel_16_0(int rounds, int a_rounds);
const void* spkm_sq_kernel_16_1(int rounds, int a_rounds);
const void* spkm_sq_kernel_32_0(int rounds, int a_rounds);
const void* spkm_sq_kernel_32_1(int rounds, int a_rounds);
template <typename IR> static const void* screen_quad_kernel(int rounds, int a_rounds, bool pts = false)
{
    if (sizeof(IR) == 2) return pts ? spkm_sq_kernel_16_1(rounds, a_rounds) : spkm_sq_kernel_16_0(rounds, a_rounds);
    return pts ? spkm_sq_kernel_32_1(rounds, a_rounds) : spkm_sq_kernel_32_0(rounds, a_rounds);
}

// A shard over RECORDS the caller holds on the device (spkm_mix_sample_rec_dev's output): n points of exactly s entries.
// The library adopts the buffer (the caller keeps it alive); it is the only copy of the entries -- the state a CSC shard
// reaches through spkm_shard_release_csc, without ever having held the arrays.  jc is the library's.
extern "C" int spkm_shard_create_rec_dev(spkm_ctx* ctx, uint64_t p, uint64_t n, uint64_t s_entries, int ir_bits,
                                         const void* d_rec, spkm_shard** out)
{
    if (!ctx || !out || (n && !d_rec)) return SPKM_ERR_NULL_ARG;
    *out = nullptr;
    if (p == 0 || p > 0x7fffffffull || n > 0x7ff00000ull) return SPKM_ERR_UNSUPPORTED;
    if (ir_bits != 16 && ir_bits != 32) return SPKM_ERR_BAD_VALUE;
    if ((ir_bits == 16 && p > 65536) || s_entries == 0 || s_entries > 64 || s_entries > p) return SPKM_ERR_BAD_VALUE;
    HIP_TRY(hipSetDevice(ctx->device));
    spkm_shard* s = new spkm_shard();
    s->ctx = ctx; s->p = p; s->n = n; s->nnz = n * s_entries; s->ir_bits = ir_bits;
    s->fixed_s = (int)s_entries;
    s->rec = (char*)d_rec; s->rec_R = (int)spkm_record_bytes(s_entries, ir_bits); s->rec_owned = false; s->rec_tried = true;
    s->slack = 48;            // (what ensure_csc gives the arrays it re-materialises)
    s->csc_released = true;
    s->owned = true;
    if (hipMalloc((void**)&s->jc, (n + 1) * 8) != hipSuccess) { delete s; return SPKM_ERR_NO_DEVICE; }
    hipLaunchKernelGGL(k_fill_jc, dim3((unsigned)std::min<uint64_t>((n + 256) / 256, 4096)), dim3(256), 0, ctx->stream, s->jc,
                       (long long)n, (long long)s_entries);
    *out = s;
    return SPKM_OK;
}

static bool screen_use_quad(const spkm_ctx* ctx, const spkm_shard* s);
template <typename IR> static int build_records(spkm_ctx* ctx, spkm_shard* sm);
template <typename IR> static int build_screen_copy(spkm_ctx* ctx, spkm_shard* sm);

extern "C" int spkm_shard_release_csc(spkm_ctx* ctx, spkm_shard* s)
{
    if (!ctx || !s) return SPKM_ERR_NULL_ARG;
    if (s->csc_released || s->nnz == 0) return SPKM_OK;
    if (s->fixed_s <= 0 || s->fixed_s > 64 || s->slack < 48 || ctx->sw.no_rec) return SPKM_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    // everything that is derived from the CSC arrays, now: the records (from here on the only copy of the exact entries)
    // and the screen's f32 copy + the certificate's norms
    s->rec_tried = false;
    rc = s->ir_bits == 16 ? build_records<unsigned short>(ctx, s) : build_records<unsigned int>(ctx, s);
    if (rc) return rc;
    if (!s->rec) return SPKM_ERR_UNSUPPORTED; // (no room for the records beside the arrays: nothing released)
    if (screen_use_quad(ctx, s)) {
        rc = s->ir_bits == 16 ? build_screen_copy<unsigned short>(ctx, s) : build_screen_copy<unsigned int>(ctx, s);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // the builders read x / ir
    if (s->owned_csc) {
        if (s->ir) (void)hipFree(s->ir);
        if (s->x) (void)hipFree(s->x);
    }
    s->ir = nullptr;
    s->x = nullptr;
    s->owned_csc = false;
    s->csc_released = true;
    return SPKM_OK;
}

// ------------------------------------------------------------------------------------------
// one-time layouts of a fixed-stride shard (built from its CSC arrays) and the way back
// ------------------------------------------------------------------------------------------
template <typename IR>
static int build_records(spkm_ctx* ctx, spkm_shard* sm)
{
    // Record layout of the exact entries (screen.hip, k_build_records): built once per shard, when the device has room
    // for it (n * R bytes: 51 GB at N = 1e8, s = 51).  SPKM_NO_REC=1: never (A/B switch, and what runs when memory is short).
    if (sm->rec || sm->rec_tried || ctx->sw.no_rec || sm->fixed_s <= 0 || !sm->x) return SPKM_OK;
    sm->rec_tried = true;
    const long long n = (long long)sm->n;
    const int R = (int)(((size_t)sm->fixed_s * (8 + sizeof(IR)) + 15) / 16 * 16);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > (size_t)n * R + ((size_t)4 << 30) &&
        hipMalloc((void**)&sm->rec, (size_t)n * R + 256) == hipSuccess) {
        hipLaunchKernelGGL((k_build_records<IR>), dim3((unsigned)std::min<long long>((n + 3) / 4, 65536)), dim3(256), 0,
                           ctx->stream, (const IR*)sm->ir, (const double*)sm->x, n, sm->fixed_s, R, sm->rec);
        sm->rec_R = R;
    } else {
        (void)hipGetLastError();
        sm->rec = nullptr;
    }
    return SPKM_OK;
}

template <typename IR>
static int build_screen_copy(spkm_ctx* ctx, spkm_shard* sm)
{
    // f32 values + row ids in the 4-lanes-per-point kernel's step-major lane order, columns partitioned by row parity
    // (k_screen_reorder), and the certificate's per-point norms on the same pass
    const long long n = (long long)sm->n;
    const int p = (int)sm->p;
    if (!sm->xnr) HIP_TRY(hipMalloc((void**)&sm->xnr, (size_t)n * 4));
    if (sm->xfs) return SPKM_OK;
    const size_t isz = sizeof(IR);
    const size_t slots = (size_t)((n + 15) / 16) * ((sm->fixed_s + 3) / 4) * 64; // steps x rounds x lanes
    HIP_TRY(hipMalloc((void**)&sm->xfs, slots * 4));
    HIP_TRY(hipMalloc((void**)&sm->irs, slots * isz));
    // (from the CSC arrays, or -- a shard created from records, or one that has released its arrays -- from the records)
    hipLaunchKernelGGL((k_screen_reorder<IR>), dim3((unsigned)std::min<long long>((n + 15) / 16, 16384)), dim3(256),
                       0, ctx->stream, (const IR*)sm->ir, (const double*)sm->x, n, sm->fixed_s, p, sm->xfs, (IR*)sm->irs,
                       sm->norms_done ? (float*)nullptr : sm->xnr,
                       sm->x == nullptr ? (const char*)sm->rec : (const char*)nullptr, sm->rec_R, (const int*)nullptr);
    sm->norms_done = true;
    return SPKM_OK;
}

// Regroup a lazy shard by cluster (screen.hip, k_regroup_keys): called at the start of a fused call whose predecessor -- a
// call over every point -- found most 16-point steps mixing clusters.  Counting sort of the library's points by (cluster,
// unsure) from the bounds that call left; the bounds and the library's copy of the assignment move with the points, the
// screen copy and the certificate's norms are rebuilt from the records in the new order (one gather of the records, one
// write of the copy), the block summaries start over.  Scratch of its own: the context's kept counting sort is untouched.
template <typename IR>
static int regroup_shard(spkm_ctx* ctx, spkm_shard* sm, int K)
{
    const long long n = (long long)sm->n, npad = sm->hb_npad;
    const int K2 = 2 * K;
    int *keys = nullptr, *perm = nullptr, *newmap = nullptr;
    float* hb_new = nullptr;
    auto fail = [&](hipError_t e) {
        (void)hipGetLastError();
        if (keys) (void)hipFree(keys);
        if (perm) (void)hipFree(perm);
        if (newmap) (void)hipFree(newmap);
        if (hb_new) (void)hipFree(hb_new);
        return e == hipErrorOutOfMemory ? SPKM_OK : (int)e; // (no room: the shard simply stays as it is)
    };
    hipError_t e;
    if ((e = hipMalloc((void**)&keys, (size_t)n * 4 + 64)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&perm, (size_t)n * 4 + 64)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&newmap, (size_t)n * 4 + 64)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&hb_new, ((size_t)3 * npad + HB_TAIL) * 4)) != hipSuccess) return fail(e);
    int rc;
    if ((rc = ensure(ctx, ctx->hist2, (size_t)K2 * 8))) return rc;
    if ((rc = ensure(ctx, ctx->offs2, (size_t)(K2 + 1) * 8))) return rc;
    if ((rc = ensure(ctx, ctx->cursor2, (size_t)K2 * 8))) return rc;
    constexpr int RG_SEG = 2048; // (the plan's items are not used: only its offsets and cursors)
    if ((rc = ensure(ctx, ctx->items2, (size_t)((n / RG_SEG) + K2 + 1) * 16))) return rc;
    if ((rc = ensure(ctx, ctx->nitems, 64))) return rc;
    const unsigned g1 = (unsigned)std::min<long long>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(k_regroup_keys, dim3(g1), dim3(256), 0, ctx->stream, (const float*)sm->hb, npad, n, K,
                       (const double*)(sm->hb_cum + sm->cum_par), keys);
    HIP_TRY(hipMemsetAsync(ctx->hist2.p, 0, (size_t)K2 * 8, ctx->stream));
    hipLaunchKernelGGL(k_hist, dim3((unsigned)std::min<long long>(1024, (n + 1023) / 1024)), dim3(256), (size_t)K2 * 4, ctx->stream,
                       (const int*)keys, n, K2, (unsigned long long*)ctx->hist2.p, (const unsigned*)nullptr);
    hipLaunchKernelGGL(k_plan_segments, dim3(1), dim3(256), 0, ctx->stream, (const unsigned long long*)ctx->hist2.p, K2, RG_SEG,
                       (long long*)ctx->offs2.p, (unsigned long long*)ctx->cursor2.p, (int4*)ctx->items2.p, (int*)ctx->nitems.p + 4,
                       (const unsigned*)nullptr);
    {
        const int sb = (int)std::max<long long>(std::min<long long>(1024, (n + 1023) / 1024), std::min<long long>(8192, n / 4096));
        const size_t sc = (size_t)((K2 + 1) & ~1) * 4 + (size_t)K2 * 12;
        if (sc > 48 * 1024) (void)allow_lds(ctx, (const void*)k_scatter_by_cluster<true>, sc);
        hipLaunchKernelGGL(k_scatter_by_cluster<true>, dim3(sb), dim3(256), sc, ctx->stream, (const int*)keys, n, K2,
                           (unsigned long long*)ctx->cursor2.p, perm, (const unsigned*)nullptr, (const int*)nullptr);
    }
    hipLaunchKernelGGL(k_regroup_apply, dim3(g1), dim3(256), 0, ctx->stream, (const int*)perm, (const int*)sm->map, newmap,
                       (const float*)sm->hb, hb_new, npad, n);
    HIP_TRY(hipMemcpyAsync(hb_new + 3 * npad, sm->hb + 3 * npad, (size_t)HB_TAIL * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipGetLastError());
    // the screen copy and the norms in the new order, over the old ones (their source is the records)
    hipLaunchKernelGGL((k_screen_reorder<IR>), dim3((unsigned)std::min<long long>((n + 15) / 16, 16384)), dim3(256), 0, ctx->stream,
                       (const IR*)nullptr, (const double*)nullptr, n, sm->fixed_s, (int)sm->p, sm->xfs, (IR*)sm->irs, sm->xnr,
                       (const char*)sm->rec, sm->rec_R, (const int*)newmap);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // (once per shard and run: the old arrays go back now)
    (void)hipFree(keys);
    (void)hipFree(perm);
    (void)hipFree(sm->hb);
    if (sm->map) (void)hipFree(sm->map);
    sm->hb = hb_new;
    sm->map = newmap;
    sm->sp_clean = false;
    return SPKM_OK;
}

// CSC arrays back from the records, library-owned (an entry point that reads CSC was called after
// spkm_shard_release_csc): one streaming pass.  The shard stays "released" in spirit -- the next release frees them again.
static int ensure_csc(spkm_ctx* ctx, const spkm_shard* s)
{
    spkm_shard* sm = const_cast<spkm_shard*>(s);
    if (sm->x != nullptr || sm->nnz == 0) return SPKM_OK;
    if (!sm->rec) return SPKM_ERR_BAD_VALUE; // (cannot happen: release requires the records)
    const size_t irb = (size_t)sm->ir_bits / 8;
    HIP_TRY(hipMalloc(&sm->ir, (sm->nnz + 48) * irb));
    HIP_TRY(hipMalloc((void**)&sm->x, (sm->nnz + 48) * sizeof(double)));
    HIP_TRY(hipMemsetAsync((char*)sm->ir + sm->nnz * irb, 0, 48 * irb, ctx->stream));
    HIP_TRY(hipMemsetAsync(sm->x + sm->nnz, 0, 48 * sizeof(double), ctx->stream));
    const long long n = (long long)sm->n;
    const unsigned grid = (unsigned)std::min<long long>((n + 3) / 4, 65536);
    if (sm->ir_bits == 16)
        hipLaunchKernelGGL((k_unpack_records<unsigned short>), dim3(grid), dim3(256), 0, ctx->stream, (const char*)sm->rec, n,
                           sm->fixed_s, sm->rec_R, (unsigned short*)sm->ir, sm->x);
    else
        hipLaunchKernelGGL((k_unpack_records<unsigned int>), dim3(grid), dim3(256), 0, ctx->stream, (const char*)sm->rec, n,
                           sm->fixed_s, sm->rec_R, (unsigned int*)sm->ir, sm->x);
    HIP_TRY(hipGetLastError());
    sm->owned_csc = true;
    sm->slack = 48;
    sm->csc_released = false;
    return SPKM_OK;
}

// ------------------------------------------------------------------------------------------
// assignment
// ------------------------------------------------------------------------------------------
static int pick_kt(const spkm_ctx* ctx, uint64_t p, uint64_t K)
{
    int best = 0;
    uint64_t best_slots = ~0ull;
    for (int kt : {16, 32, 64}) {
        if ((p + 1) * (uint64_t)kt * 8 + 16 > ctx->lds_max) continue;
        const uint64_t slots = ((K + kt - 1) / kt) * kt;
        if (slots < best_slots || (slots == best_slots && kt > best)) { best = kt; best_slots = slots; }
    }
    return best; // 0: no tile fits -> generic kernel
}

// Workgroup -> (tile, chunk stream).  Workgroup b is observed to run on XCD b % 8, so the G
// workgroups that stream the same chunks (one per tile) are given ids that share an XCD and
// its L2: the chunk is fetched from HBM once and re-read from L2 by the other tiles.  This is
// a speed heuristic only -- any placement gives the same results.
static int build_blockmap(spkm_ctx* ctx, int G)
{
    const int NB = ctx->num_cus > 0 ? ctx->num_cus : 256;
    if (ctx->bmap_G == G && ctx->bmap_blocks == NB) return SPKM_OK;
    std::vector<spkm_blockmap> bm(NB, spkm_blockmap{-1, 0, 1, 0});
    const int NX = (NB % 8 == 0) ? 8 : 1;
    const int per_xcd = NB / NX;
    const int local_streams = per_xcd / G;
    int nstreams = 0;
    std::vector<int> spare;
    for (int b = 0; b < NB; b++) {
        const int xcd = b % NX, i = b / NX;
        if (i < local_streams * G) {
            bm[b].tile = i % G;
            bm[b].stream = xcd * local_streams + i / G;
        } else spare.push_back(b);
    }
    nstreams = NX * local_streams;
    const int extra = (int)spare.size() / G; // floating streams built from the left-over workgroups
    for (int t = 0; t < extra * G; t++) {
        bm[spare[t]].tile = t % G;
        bm[spare[t]].stream = nstreams + t / G;
    }
    nstreams += extra;
    if (nstreams == 0) return SPKM_ERR_UNSUPPORTED; // more tiles than workgroups
    for (auto& e : bm) e.nstreams = nstreams;
    int rc = ensure(ctx, ctx->bmap, NB * sizeof(spkm_blockmap));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->bmap.p, bm.data(), NB * sizeof(spkm_blockmap), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->bmap_G = G; ctx->bmap_blocks = NB; ctx->bmap_streams = nstreams;
    return SPKM_OK;
}

// Block map of the 4-lanes-per-point screen.  The last tile may be narrow (pl_last centroid pairs per lane
// instead of 4) and then costs a fraction of a full tile per point, so the workgroups of each XCD are split
// over the tiles in proportion to the tiles' cost; all tiles sweep the XCD's chunks (chunk c belongs to XCD
// c % NX) in the same order, so a chunk is fetched from HBM once and re-read from that XCD's L2.
//   entry: tile, stream = index among the tile's workgroups on this XCD, nstreams = their number,
//          pad = NX | xcd << 8 | pairs-per-lane << 16
static int build_blockmap_quad(spkm_ctx* ctx, int G, int pl_last, int rounds)
{
    devbuf& buf = ctx->bmapq;
    int& slot_key = ctx->bmapq_key;
    int& slot_blocks = ctx->bmapq_blocks;
    const int NB = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const int key = G * 64 + pl_last * 8 + 1000003 * rounds + (ctx->sw.no_teams ? 7 : 0);
    if (slot_key == key && slot_blocks == NB) return SPKM_OK;
    const int NX = (NB % 8 == 0) ? 8 : 1;
    const int per = NB / NX;
    if (per < G) return SPKM_ERR_UNSUPPORTED;
    // issue cycles per 16-point step: rounds * (32 + 32 pl (+12 for the second address)) + overhead
    // (overhead fitted to K = 37 / 100 / 200 timings: a narrow tile costs about half a full one at 13 rounds)
    auto cost = [&](int pl) {
        // full tile + one extra centroid per lane: 32 issue cycles per round by count, ~60 measured (its 4-B LDS
        // reads of 16 random 16-B rows conflict)
        if (pl == 5) return (double)rounds * (32.0 + 128.0 + 12.0 + 60.0) + 580.0;
        return (double)rounds * (32.0 + 32.0 * pl + (pl == 4 ? 12.0 : 0.0)) + 560.0;
    };
    std::vector<spkm_blockmap> bm(NB, spkm_blockmap{-1, 0, 1, 0});
    if (pl_last >= 4 && !ctx->sw.no_teams) {
        // TEAMS: every tile costs the same per chunk (full tiles; with pl_last = 5 the remainder of <= 4 centroids is
        // carried by the G tiles in turn, chunk by chunk -- screen_quad.hip, `rot`).  A team is one workgroup per tile on
        // ONE XCD; team t takes chunks t, t + nteams, ...; its members sweep them in the same order at the same pace, so
        // a chunk is fetched from HBM once and met in that XCD's L2 by the other tiles (half the traffic of tiles that
        // drift apart, and with it a higher sustained clock for this power-bound kernel: 1.77 -> 1.98 GHz measured).
        // The per % G workgroups an XCD has left over form teams ACROSS XCDs (no L2 sharing, a few per cent of the
        // chunks) instead of idling.
        const int T = per / G, spare = per - T * G;
        const int F = (NX * spare) / G;          // floating teams
        const int nteams = NX * T + F;
        const int rot = pl_last == 5 ? G : 0;
        std::vector<int> spares;
        for (int x = 0; x < NX; x++)
            for (int i = 0; i < per; i++) {
                const int b = i * NX + x; // workgroup b runs on XCD b % NX
                if (i < T * G) {
                    spkm_blockmap& e = bm[b];
                    e.tile = i % G;
                    e.stream = x + NX * (i / G);
                    e.nstreams = nteams;
                    e.pad = 1 | (0 << 8) | (pl_last << 16) | (rot << 24);
                } else
                    spares.push_back(b);
            }
        for (int f = 0; f < F; f++)
            for (int g = 0; g < G; g++) {
                spkm_blockmap& e = bm[spares[f * G + g]];
                e.tile = g;
                e.stream = NX * T + f;
                e.nstreams = nteams;
                e.pad = 1 | (0 << 8) | (pl_last << 16) | (rot << 24);
            }
    } else {
    std::vector<double> w(G, cost(4));
    w[G - 1] = cost(pl_last);
    // apportionment of the XCD's workgroups, at least one per tile, minimising the makespan
    std::vector<int> cnt(G, 1);
    for (int left = per - G; left > 0; left--) {
        int best = 0;
        double worst = -1;
        for (int g = 0; g < G; g++)
            if (w[g] / cnt[g] > worst) { worst = w[g] / cnt[g]; best = g; }
        cnt[best]++;
    }
    for (int x = 0; x < NX; x++) {
        int i = 0;
        for (int g = 0; g < G; g++)
            for (int j = 0; j < cnt[g]; j++, i++) {
                spkm_blockmap& e = bm[i * NX + x]; // workgroup b runs on XCD b % NX
                e.tile = g;
                e.stream = j;
                e.nstreams = cnt[g];
                e.pad = NX | (x << 8) | ((g == G - 1 ? pl_last : 4) << 16);
            }
    }
    }
    int rc = ensure(ctx, buf, NB * sizeof(spkm_blockmap));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(buf.p, bm.data(), NB * sizeof(spkm_blockmap), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    slot_key = key; slot_blocks = NB;
    return SPKM_OK;
}

// Events bracket the dominant kernel on the context's own stream.  With the log enabled each
// launch gets its own pair, so bench.py can read all durations after its timed region without
// adding a host sync inside it.
static hipError_t timing_begin(spkm_ctx* ctx)
{
    if (ctx->tlog_on) {
        if (ctx->tlog_used == ctx->tlog.size()) {
            hipEvent_t a, b;
            hipError_t e = hipEventCreate(&a);
            if (e != hipSuccess) return e;
            e = hipEventCreate(&b);
            if (e != hipSuccess) return e;
            ctx->tlog.emplace_back(a, b);
        }
        return hipEventRecord(ctx->tlog[ctx->tlog_used].first, ctx->stream);
    }
    return hipEventRecord(ctx->ev0, ctx->stream);
}
static hipError_t timing_end(spkm_ctx* ctx)
{
    if (ctx->tlog_on) return hipEventRecord(ctx->tlog[ctx->tlog_used++].second, ctx->stream);
    ctx->ev_valid = true;
    return hipEventRecord(ctx->ev1, ctx->stream);
}

template <int KT, typename IR, bool FIXED>
static int launch_tile2(spkm_ctx* ctx, const spkm_shard* s, int K, int G, int chunk)
{
    const size_t lds = (size_t)(s->p + 1) * KT * 8 + 16; // tile + work-ticket counter
    auto kern = k_assign_tile<KT, IR, FIXED>;
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(timing_begin(ctx));
    hipLaunchKernelGGL(kern, dim3(ctx->bmap_blocks), dim3(1024), lds, ctx->stream, (const long long*)s->jc,
                       (const IR*)s->ir, (const double*)s->x, (const double*)ctx->tiles.p, (int)s->p, (int)s->n,
                       (long long)s->nnz, s->fixed_s, K, (const spkm_blockmap*)ctx->bmap.p, chunk,
                       (double*)ctx->part_acc.p, (int*)ctx->part_k.p, (long long*)ctx->dbg.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(timing_end(ctx));
    return SPKM_OK;
}

template <int KT, typename IR>
static int launch_tile(spkm_ctx* ctx, const spkm_shard* s, int K, int G, int chunk)
{
    // the fixed-stride kernel reads up to 15 entries past a column's end: needs slack after nnz
    if (s->fixed_s > 0 && s->slack >= 16) return launch_tile2<KT, IR, true>(ctx, s, K, G, chunk);
    return launch_tile2<KT, IR, false>(ctx, s, K, G, chunk);
}

static constexpr int COMBINE_BLOCKS = 1024;
static constexpr int FIN_BLOCKS_MAX = 256;

static int combine_partials(spkm_ctx* ctx, long long n, int G, int K, int32_t* d_assign, double* d_mind,
                            double* d_stats, uint64_t* d_nk_u64)
{
    int rc;
    const int cb = (int)std::min<long long>(COMBINE_BLOCKS, (n + 255) / 256);
    if ((rc = ensure(ctx, ctx->blk_obj, (size_t)cb * 8))) return rc;
    if ((rc = ensure(ctx, ctx->blk_max, (size_t)cb * 8))) return rc;
    if ((rc = ensure(ctx, ctx->blk_imax, (size_t)cb * 8))) return rc;
    hipLaunchKernelGGL(k_combine, dim3(cb), dim3(256), (size_t)K * 4, ctx->stream, (const double*)ctx->part_acc.p,
                       (const int*)ctx->part_k.p, n, G, K, (int*)d_assign, d_mind, (double*)ctx->blk_obj.p,
                       (double*)ctx->blk_max.p, (long long*)ctx->blk_imax.p, (unsigned long long*)ctx->nk.p);
    hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(64), 0, ctx->stream, (const double*)ctx->blk_obj.p,
                       (const double*)ctx->blk_max.p, (const long long*)ctx->blk_imax.p, cb, (double*)ctx->stats.p);
    HIP_TRY(hipGetLastError());
    if (d_stats) HIP_TRY(hipMemcpyAsync(d_stats, ctx->stats.p, 3 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (d_nk_u64) HIP_TRY(hipMemcpyAsync(d_nk_u64, ctx->nk.p, (size_t)K * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return SPKM_OK;
}

extern "C" int spkm_assign_dev(spkm_ctx* ctx, const spkm_shard* s, uint64_t K64, const double* d_centers,
                               double gamma, int32_t* d_assign, double* d_mind, double* d_stats,
                               uint64_t* d_nk_u64)
{
    if (!ctx || !s || !d_centers || !d_assign || !d_mind) return SPKM_ERR_NULL_ARG;
    if (K64 == 0 || K64 > 65536) return SPKM_ERR_UNSUPPORTED;
    ctx->sort_owner = nullptr; // this call overwrites (some of) the buffers a kept counting sort lives in
    HIP_TRY(hipSetDevice(ctx->device));
    const_cast<spkm_shard*>(s)->sp_clean = false; // (this call writes d_assign: the block summaries' claim on that buffer ends)
    if (int rcc = ensure_csc(ctx, s)) return rcc;
    const int K = (int)K64, p = (int)s->p;
    const long long n = (long long)s->n;
    int rc;
    if ((rc = ensure(ctx, ctx->nk, (size_t)K * 8))) return rc;
    if ((rc = ensure(ctx, ctx->stats, 4 * 8))) return rc;
    HIP_TRY(hipMemsetAsync(ctx->nk.p, 0, (size_t)K * 8, ctx->stream));
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(ctx->stats.p, 0, 4 * 8, ctx->stream));
        if (d_stats) HIP_TRY(hipMemsetAsync(d_stats, 0, 3 * 8, ctx->stream));
        if (d_nk_u64) HIP_TRY(hipMemsetAsync(d_nk_u64, 0, (size_t)K * 8, ctx->stream));
        return SPKM_OK;
    }
    // K = 1 on a fixed-stride shard (the k-means++ rounds): a plain stream over X, no tiles, no partials
    if (K == 1 && s->fixed_s > 0 && s->nnz > 0) {
        const int threads = 1024, nw = threads / 64;
        const size_t per_pt = (size_t)(s->fixed_s | 1) * 8;
        const size_t fixed_lds = (size_t)p * 8;
        if (fixed_lds + 1024 + (size_t)nw * 16 * per_pt <= ctx->lds_max) {
            int pts = (int)std::min<size_t>(64, (ctx->lds_max - fixed_lds - 1024) / nw / per_pt);
            pts = std::max(16, pts & ~15);
            const size_t lds1 = fixed_lds + (size_t)nw * pts * per_pt;
            const int nb = (int)std::min<long long>(std::max(1, ctx->num_cus), (n + (long long)nw * pts - 1) / ((long long)nw * pts));
            if ((rc = ensure(ctx, ctx->blk_obj, (size_t)nb * 8))) return rc;
            if ((rc = ensure(ctx, ctx->blk_max, (size_t)nb * 8))) return rc;
            if ((rc = ensure(ctx, ctx->blk_imax, (size_t)nb * 8))) return rc;
            ctx->ev_valid = false;
            HIP_TRY(timing_begin(ctx));
            if (s->ir_bits == 16) {
                auto k1 = k_exact_dist1<unsigned short, 8>;
                HIP_TRY(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
                hipLaunchKernelGGL(k1, dim3(nb), dim3(threads), lds1, ctx->stream, (const unsigned short*)s->ir,
                                   (const double*)s->x, d_centers, gamma, p, n, s->fixed_s, pts, (int*)d_assign, d_mind,
                                   (double*)ctx->blk_obj.p, (double*)ctx->blk_max.p, (long long*)ctx->blk_imax.p);
            } else {
                auto k1 = k_exact_dist1<unsigned int, 8>;
                HIP_TRY(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
                hipLaunchKernelGGL(k1, dim3(nb), dim3(threads), lds1, ctx->stream, (const unsigned int*)s->ir,
                                   (const double*)s->x, d_centers, gamma, p, n, s->fixed_s, pts, (int*)d_assign, d_mind,
                                   (double*)ctx->blk_obj.p, (double*)ctx->blk_max.p, (long long*)ctx->blk_imax.p);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(timing_end(ctx));
            hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(64), 0, ctx->stream, (const double*)ctx->blk_obj.p,
                               (const double*)ctx->blk_max.p, (const long long*)ctx->blk_imax.p, nb, (double*)ctx->stats.p);
            hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, ctx->stream, (unsigned long long*)ctx->nk.p,
                               (unsigned long long)n);
            HIP_TRY(hipGetLastError());
            ctx->assign_KT = 0;
            ctx->assign_G = 1;
            if (d_stats) HIP_TRY(hipMemcpyAsync(d_stats, ctx->stats.p, 3 * 8, hipMemcpyDeviceToDevice, ctx->stream));
            if (d_nk_u64) HIP_TRY(hipMemcpyAsync(d_nk_u64, ctx->nk.p, 8, hipMemcpyDeviceToDevice, ctx->stream));
            return SPKM_OK;
        }
    }
    const int KT = (s->nnz > 0) ? pick_kt(ctx, s->p, K64) : 0;
    int G = 1;
    ctx->ev_valid = false;
    bool tiled = KT > 0;
    if (tiled) {
        G = (K + KT - 1) / KT;
        rc = build_blockmap(ctx, G);
        if (rc == SPKM_ERR_UNSUPPORTED) tiled = false; // more tiles than workgroups
        else if (rc) return rc;
    }
    if (tiled) {
        const size_t tile_doubles = (size_t)G * (p + 1) * KT;
        if ((rc = ensure(ctx, ctx->tiles, tile_doubles * 8))) return rc;
        if ((rc = ensure(ctx, ctx->part_acc, (size_t)G * n * 8))) return rc;
        if ((rc = ensure(ctx, ctx->part_k, (size_t)G * n * 4))) return rc;
        hipLaunchKernelGGL(k_prep_tiles, dim3(std::min<size_t>((tile_doubles + 255) / 256, 2048)), dim3(256), 0,
                           ctx->stream, d_centers, p, K, KT, G, gamma, (double*)ctx->tiles.p);
        // chunk: multiple of the points one workgroup covers per sweep; small enough that every
        // stream gets several chunks, large enough to amortise the loop overhead
        const int ppw = 64 / KT, sweep = 16 * 2 * ppw;
        long long chunk = n / ((long long)ctx->bmap_streams * 8);
        chunk = std::max<long long>(sweep, std::min<long long>(chunk, 16 * sweep));
        chunk = (chunk / sweep) * sweep;
        if (s->ir_bits == 16) {
            if (KT == 16) rc = launch_tile<16, unsigned short>(ctx, s, K, G, (int)chunk);
            else if (KT == 32) rc = launch_tile<32, unsigned short>(ctx, s, K, G, (int)chunk);
            else rc = launch_tile<64, unsigned short>(ctx, s, K, G, (int)chunk);
        } else {
            if (KT == 16) rc = launch_tile<16, unsigned int>(ctx, s, K, G, (int)chunk);
            else if (KT == 32) rc = launch_tile<32, unsigned int>(ctx, s, K, G, (int)chunk);
            else rc = launch_tile<64, unsigned int>(ctx, s, K, G, (int)chunk);
        }
        if (rc) return rc;
    } else {
        // generic path: row-major scaled centroids in global memory, one wave per point
        G = 1;
        if ((rc = ensure(ctx, ctx->ct, (size_t)p * K * 8))) return rc;
        if ((rc = ensure(ctx, ctx->part_acc, (size_t)n * 8))) return rc;
        if ((rc = ensure(ctx, ctx->part_k, (size_t)n * 4))) return rc;
        hipLaunchKernelGGL(k_prep_rowmajor, dim3(std::min<size_t>(((size_t)p * K + 255) / 256, 2048)), dim3(256), 0,
                           ctx->stream, d_centers, p, K, gamma, (double*)ctx->ct.p);
        HIP_TRY(timing_begin(ctx));
        const int blocks = std::max(1, ctx->num_cus) * 8;
        if (s->ir_bits == 16)
            hipLaunchKernelGGL((k_assign_generic<unsigned short>), dim3(blocks), dim3(256), 0, ctx->stream,
                               (const long long*)s->jc, (const unsigned short*)s->ir, (const double*)s->x,
                               (const double*)ctx->ct.p, K, n, (double*)ctx->part_acc.p, (int*)ctx->part_k.p);
        else
            hipLaunchKernelGGL((k_assign_generic<unsigned int>), dim3(blocks), dim3(256), 0, ctx->stream,
                               (const long long*)s->jc, (const unsigned int*)s->ir, (const double*)s->x,
                               (const double*)ctx->ct.p, K, n, (double*)ctx->part_acc.p, (int*)ctx->part_k.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(timing_end(ctx));
    }
    ctx->assign_KT = KT;
    ctx->assign_G = G;
    return combine_partials(ctx, n, G, K, d_assign, d_mind, d_stats, d_nk_u64);
}

extern "C" int spkm_assign_sparse_centers_dev(spkm_ctx* ctx, const spkm_shard* s, uint64_t K64,
                                              const double* d_centers, const uint8_t* d_mask, double gamma,
                                              int32_t* d_assign, double* d_mind, double* d_stats,
                                              uint64_t* d_nk_u64)
{
    if (!ctx || !s || !d_centers || !d_mask || !d_assign || !d_mind) return SPKM_ERR_NULL_ARG;
    if (K64 == 0 || K64 > 65536) return SPKM_ERR_UNSUPPORTED;
    ctx->sort_owner = nullptr; // this call overwrites (some of) the buffers a kept counting sort lives in
    HIP_TRY(hipSetDevice(ctx->device));
    const_cast<spkm_shard*>(s)->sp_clean = false; // (this call writes d_assign: the block summaries' claim on that buffer ends)
    if (int rcc = ensure_csc(ctx, s)) return rcc;
    const int K = (int)K64, p = (int)s->p;
    const long long n = (long long)s->n;
    int rc;
    if ((rc = ensure(ctx, ctx->nk, (size_t)K * 8))) return rc;
    if ((rc = ensure(ctx, ctx->stats, 4 * 8))) return rc;
    HIP_TRY(hipMemsetAsync(ctx->nk.p, 0, (size_t)K * 8, ctx->stream));
    ctx->ev_valid = false;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(ctx->stats.p, 0, 4 * 8, ctx->stream));
        if (d_stats) HIP_TRY(hipMemsetAsync(d_stats, 0, 3 * 8, ctx->stream));
        if (d_nk_u64) HIP_TRY(hipMemsetAsync(d_nk_u64, 0, (size_t)K * 8, ctx->stream));
        return SPKM_OK;
    }
    if ((rc = ensure(ctx, ctx->ct, (size_t)p * K * 8))) return rc;
    if ((rc = ensure(ctx, ctx->tmp_mind, (size_t)p * K))) return rc;     // row-major mask
    if ((rc = ensure(ctx, ctx->tmp_assign, (size_t)K * 8))) return rc;   // gamma_c per centre
    if ((rc = ensure(ctx, ctx->part_acc, (size_t)n * 8))) return rc;
    if ((rc = ensure(ctx, ctx->part_k, (size_t)n * 4))) return rc;
    hipLaunchKernelGGL(k_prep_sparse_centers, dim3(K), dim3(256), 0, ctx->stream, d_centers, d_mask, p, K, gamma,
                       (double*)ctx->ct.p, (unsigned char*)ctx->tmp_mind.p, (double*)ctx->tmp_assign.p);
    const int blocks = std::max(1, ctx->num_cus) * 8;
    const int scale = gamma > 0.0 ? 1 : 0;
    if (s->ir_bits == 16)
        hipLaunchKernelGGL((k_assign_sparse_centers<unsigned short>), dim3(blocks), dim3(256), 0, ctx->stream,
                           (const long long*)s->jc, (const unsigned short*)s->ir, (const double*)s->x,
                           (const double*)ctx->ct.p, (const unsigned char*)ctx->tmp_mind.p,
                           (const double*)ctx->tmp_assign.p, scale, K, n, (double*)ctx->part_acc.p,
                           (int*)ctx->part_k.p);
    else
        hipLaunchKernelGGL((k_assign_sparse_centers<unsigned int>), dim3(blocks), dim3(256), 0, ctx->stream,
                           (const long long*)s->jc, (const unsigned int*)s->ir, (const double*)s->x,
                           (const double*)ctx->ct.p, (const unsigned char*)ctx->tmp_mind.p,
                           (const double*)ctx->tmp_assign.p, scale, K, n, (double*)ctx->part_acc.p,
                           (int*)ctx->part_k.p);
    HIP_TRY(hipGetLastError());
    return combine_partials(ctx, n, 1, K, d_assign, d_mind, d_stats, d_nk_u64);
}

// ------------------------------------------------------------------------------------------
// accumulation + finalise
// ------------------------------------------------------------------------------------------
// counting-sort placement; reads the assignment with 16-B loads when the caller's pointer allows it
static void launch_scatter(spkm_ctx* ctx, int sb, size_t sc_lds, const int* d_assign, long long n, int K, const unsigned* gate,
                           const int* need, const unsigned* n_dev = nullptr, const int* ids = nullptr)
{
    if (sc_lds > 48 * 1024) { // (K in the thousands: beyond the default dynamic-LDS allowance)
        (void)allow_lds(ctx, (const void*)k_scatter_by_cluster<true>, sc_lds);
        (void)allow_lds(ctx, (const void*)k_scatter_by_cluster<false>, sc_lds);
    }
    if (((uintptr_t)d_assign & 15) == 0)
        hipLaunchKernelGGL(k_scatter_by_cluster<true>, dim3(sb), dim3(256), sc_lds, ctx->stream, d_assign, n, K,
                           (unsigned long long*)ctx->cursor.p, (int*)ctx->perm.p, gate, need, n_dev, ids);
    else
        hipLaunchKernelGGL(k_scatter_by_cluster<false>, dim3(sb), dim3(256), sc_lds, ctx->stream, d_assign, n, K,
                           (unsigned long long*)ctx->cursor.p, (int*)ctx->perm.p, gate, need, n_dev, ids);
}

static constexpr int SEG_POINTS = 2048;
// confirmation pass: longer segments amortise the per-segment slab reset / flush (13.4 -> 12.4 ms at N = 1e8 from
// 2048 to 8192 points) as long as every workgroup still gets >= 16 of them
static int seg_points(long long n, int blocks)
{
    const long long want = n / ((long long)std::max(1, blocks) * 16);
    return (int)std::max<long long>(SEG_POINTS, std::min<long long>(8192, want));
}

extern "C" int spkm_accumulate_dev(spkm_ctx* ctx, const spkm_shard* s, uint64_t K64, const int32_t* d_assign,
                                   double* d_reduce)
{
    if (!ctx || !s || !d_assign || !d_reduce) return SPKM_ERR_NULL_ARG;
    if (K64 == 0 || K64 > 65536) return SPKM_ERR_UNSUPPORTED;
    ctx->sort_owner = nullptr; // this call overwrites (some of) the buffers a kept counting sort lives in
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rcc = ensure_csc(ctx, s)) return rcc;
    const int K = (int)K64, p = (int)s->p;
    const long long n = (long long)s->n;
    const size_t pk = (size_t)p * K;
    double* sums = d_reduce;
    double* counts = d_reduce + pk;
    double* nk_f = d_reduce + 2 * pk;
    double* obj2 = nk_f + K;
    HIP_TRY(hipMemsetAsync(d_reduce, 0, (2 * pk + K + 1) * 8, ctx->stream));
    if (!ctx->nk.p || !ctx->stats.p) return SPKM_ERR_BAD_VALUE; // spkm_assign_dev must come first
    int rc;
    const size_t slab = (size_t)p * 12;
    if (n > 0 && s->nnz > 0) {
        if (slab <= ctx->lds_max && slab <= 64 * 1024) {
            const int max_items = (int)(n / SEG_POINTS) + K + 1;
            if ((rc = ensure(ctx, ctx->perm, (size_t)n * 4))) return rc;
            if ((rc = ensure(ctx, ctx->offs, (size_t)(K + 1) * 8))) return rc;
            if ((rc = ensure(ctx, ctx->cursor, (size_t)K * 8))) return rc;
            if ((rc = ensure(ctx, ctx->items, (size_t)max_items * 16))) return rc;
            if ((rc = ensure(ctx, ctx->nitems, 64))) return rc;
            hipLaunchKernelGGL(k_plan_segments, dim3(1), dim3(256), 0, ctx->stream,
                               (const unsigned long long*)ctx->nk.p, K, SEG_POINTS, (long long*)ctx->offs.p,
                               (unsigned long long*)ctx->cursor.p, (int4*)ctx->items.p, (int*)ctx->nitems.p, (const unsigned*)nullptr);
            const int sb = (int)std::min<long long>(1024, (n + 1023) / 1024);
            const size_t sc_lds = (size_t)((K + 1) & ~1) * 4 + (size_t)K * 12;
            launch_scatter(ctx, sb, sc_lds, (const int*)d_assign, n, K, (const unsigned*)nullptr, (const int*)nullptr);
            const int ab = std::min(max_items, std::max(1, ctx->num_cus) * 8);
            if (s->ir_bits == 16)
                hipLaunchKernelGGL((k_accumulate_sorted<unsigned short>), dim3(ab), dim3(256), slab, ctx->stream,
                                   (const long long*)s->jc, (const unsigned short*)s->ir, (const double*)s->x,
                                   (const int*)ctx->perm.p, (const long long*)ctx->offs.p, (const int4*)ctx->items.p,
                                   (const int*)ctx->nitems.p, p, s->fixed_s, sums, counts);
            else
                hipLaunchKernelGGL((k_accumulate_sorted<unsigned int>), dim3(ab), dim3(256), slab, ctx->stream,
                                   (const long long*)s->jc, (const unsigned int*)s->ir, (const double*)s->x,
                                   (const int*)ctx->perm.p, (const long long*)ctx->offs.p, (const int4*)ctx->items.p,
                                   (const int*)ctx->nitems.p, p, s->fixed_s, sums, counts);
        } else {
            const int blocks = std::max(1, ctx->num_cus) * 8;
            if (s->ir_bits == 16)
                hipLaunchKernelGGL((k_accumulate_atomic<unsigned short>), dim3(blocks), dim3(256), 0, ctx->stream,
                                   (const long long*)s->jc, (const unsigned short*)s->ir, (const double*)s->x,
                                   (const int*)d_assign, p, n, sums, counts);
            else
                hipLaunchKernelGGL((k_accumulate_atomic<unsigned int>), dim3(blocks), dim3(256), 0, ctx->stream,
                                   (const long long*)s->jc, (const unsigned int*)s->ir, (const double*)s->x,
                                   (const int*)d_assign, p, n, sums, counts);
        }
    }
    hipLaunchKernelGGL(k_nk_to_f64, dim3((K + 255) / 256), dim3(256), 0, ctx->stream,
                       (const unsigned long long*)ctx->nk.p, K, nk_f);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(obj2, ctx->stats.p, 8, hipMemcpyDeviceToDevice, ctx->stream));
    return SPKM_OK;
}

// ------------------------------------------------------------------------------------------
// fused iteration front half: assignment + accumulation (everything before the all-reduce)
// ------------------------------------------------------------------------------------------
// The 4-lanes-per-point screen keeps a point's entries in registers (up to 64); longer columns use the
// first-generation 16-lanes-per-point kernel.
static bool screen_use_quad(const spkm_ctx* ctx, const spkm_shard* s)
{
    return s->fixed_s <= 64;
}

static bool screen_eligible(const spkm_ctx* ctx, const spkm_shard* s, int K)
{
    if (ctx->sw.no_screen) return false;
    if (s->fixed_s <= 0 || s->slack < 48 || s->nnz == 0) return false; // the screen reads up to 33 entries past a column
    // K <= 16 fits one exact tile that streams X once; the 4-lanes-per-point screen (one narrow tile) + exact
    // confirmation is still ~13 % faster per iteration there (K = 10, N = 2e7: 4.6 vs 5.2 ms).  K = 1 has nothing to screen.
    if (K < 2 || (K <= 16 && !screen_use_quad(ctx, s))) return false;
    if ((s->p + 1) * (uint64_t)SCREEN_KT * 4 + 16 > ctx->lds_max) return false;
    const int nb = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const int tiles = (K + SCREEN_KT - 1) / SCREEN_KT;
    if (tiles > nb) return false;
    // the 4-lanes-per-point kernel gives every tile at least one workgroup per XCD
    if (screen_use_quad(ctx, s) && tiles > ((nb % 8 == 0) ? nb / 8 : nb)) return false;
    // phase 2 needs the centroid column + slab + at least 8 staged points per wave
    const size_t per_pt = (size_t)(s->fixed_s | 1) * 8;
    if (s->p * 20 + 1024 + 16 * 8 * per_pt > ctx->lds_max) return false;
    return true;
}

template <typename IR>
static int run_screen(spkm_ctx* ctx, const spkm_shard* s, int K, const double* d_centers, double gamma,
                      int32_t* d_assign, double* d_mind, double* d_reduce, int prune_a, bool want_hint,
                      double* d_stats, uint64_t* d_nk_u64)
{
    const int p = (int)s->p;
    const long long n = (long long)s->n;
    const bool quad = screen_use_quad(ctx, s);
    const int G = (K + SCREEN_KT - 1) / SCREEN_KT;
    const size_t pk = (size_t)p * K;
    double* sums = d_reduce;
    double* counts = d_reduce + pk;
    double* nk_f = d_reduce + 2 * pk;
    double* obj2 = nk_f + K;
    int rc;
    spkm_shard* sm = const_cast<spkm_shard*>(s);
    if (!quad && !sm->xnr) HIP_TRY(hipMalloc((void**)&sm->xnr, (size_t)n * 4));
    if (!quad && (rc = ensure_csc(ctx, s))) return rc; // (the 16-lanes-per-point screen streams the CSC arrays themselves)
    if (!quad && !sm->xf) {
        // f32 copy of the values in storage order
        HIP_TRY(hipMalloc((void**)&sm->xf, (size_t)(s->nnz + 48) * 4));
        HIP_TRY(hipMemsetAsync(sm->xf, 0, (size_t)(s->nnz + 48) * 4, ctx->stream));
    }
    if (!quad && (!sm->norms_done || !sm->xf_done)) {
        hipLaunchKernelGGL(k_point_norms, dim3((unsigned)std::min<long long>((n + 15) / 16, 16384)), dim3(256), 0,
                           ctx->stream, (const long long*)s->jc, (const double*)s->x, n, s->fixed_s, sm->xnr, sm->xf);
        sm->norms_done = true;
        sm->xf_done = true;
    }
    if (quad && !sm->xfs) {
        if (!(sm->x == nullptr && sm->rec != nullptr) && (rc = ensure_csc(ctx, s))) return rc; // (the records serve as well)
        if ((rc = build_screen_copy<IR>(ctx, sm))) return rc;
    }
    // Last tile of the 4-lanes-per-point kernel.  <= 4 centroids: no tile of their own -- the workgroups of the
    // previous tile carry them as one extra centroid per lane (pl 5; needs (p+1) x 16 B more LDS); <= 16: a
    // narrow tile with 1 or 2 centroid pairs per lane instead of 4.  Gs = tiles that have workgroups / result slots.
    const int k_last = K - (G - 1) * SCREEN_KT;
    int pl_last = !quad ? 4 : (k_last <= 8 ? 1 : (k_last <= 16 ? 2 : 4));
    if (quad && G >= 2 && k_last <= 4 &&
        (size_t)(p + 1) * (SCREEN_KT * 4 + 16) + 16 <= ctx->lds_max)
        pl_last = 5;
    const int Gs = pl_last == 5 ? G - 1 : G;
    const int q_rounds = (s->fixed_s + 3) / 4;
    if (quad) rc = build_blockmap_quad(ctx, Gs, pl_last, q_rounds);
    else rc = build_blockmap(ctx, G);
    if (rc) return rc;
    const size_t tile_floats = (size_t)G * (p + 1) * SCREEN_KT;
    if ((rc = ensure(ctx, ctx->t32, tile_floats * 4))) return rc;
    if ((rc = ensure(ctx, ctx->cmax, 64))) return rc;
    if ((rc = ensure(ctx, ctx->scr_m1, (size_t)Gs * n * 4))) return rc;
    if ((rc = ensure(ctx, ctx->scr_m2, (size_t)Gs * n * 4))) return rc;
    if ((rc = ensure(ctx, ctx->scr_k, (size_t)Gs * n * 4))) return rc;
    if ((rc = ensure(ctx, ctx->list, (size_t)n * 4))) return rc;
    {
        const bool fresh = ctx->nlist.p == nullptr;
        if ((rc = ensure(ctx, ctx->nlist, 256))) return rc;
        if (fresh) HIP_TRY(hipMemsetAsync(ctx->nlist.p, 0, 256, ctx->stream)); // [8..9]: running total of skipped steps
    }
    if ((rc = ensure(ctx, ctx->ct, pk * 8))) return rc;
    if ((rc = ensure(ctx, ctx->nk, (size_t)K * 8))) return rc;
    if ((rc = ensure(ctx, ctx->stats, 4 * 8))) return rc;
    // everything this call needs zeroed, in one launch (flushed in front of the first kernel): the counters, the
    // largest-drift cell, the touched flags of the cluster shortcut and the caller's reduce buffer
    spkm_zero_jobs zj;
    zj.n = 0;
    auto zero_later = [&](void* q, size_t bytes) { zj.p[zj.n] = (unsigned*)q; zj.words[zj.n] = bytes / 4; zj.n++; };
    auto zero_flush = [&]() {
        if (!zj.n) return;
        unsigned long long mx = 0;
        for (int q = 0; q < zj.n; q++) mx = std::max(mx, zj.words[q]);
        hipLaunchKernelGGL(k_zero_many, dim3((unsigned)std::min<unsigned long long>((mx + 1023) / 1024, 256)), dim3(256), 0,
                           ctx->stream, zj);
        zj.n = 0;
    };
    zero_later(ctx->cmax.p, 16);
    zero_later(ctx->nlist.p, 32);
    zero_later((char*)ctx->nlist.p + 40, 128 - 40);    // (not the running total at [8..9])
    zero_later(d_reduce, (2 * pk + K + 1) * 8);
    // bounds carried from this shard's previous screen call (screen.hip, k_center_drift): steps whose points
    // provably keep their centroids are skipped.  SPKM_NO_BOUNDS=1: A/B switch (bounds are still maintained).
    const long long npad = (n + 63) / 64 * 64;
    int bstat_n = 0; // workgroups of k_bounds_steps whose statistics wait in ctx->bstat
    bool drift_ran = false; // k_center_drift compared this call's centroids with the previous call's (same[] is current)
    bool skipping = false, hinted = false, pt_mode = false, bounds_ok = false, kept = false, ev_path = false, ev_possible = false, pair_ev = false;
    bool sp_maintained = false; // this call's bounds test kept the block summaries
    bool trusted = false;       // the caller's assignment buffer holds the library's copy (lazy contract): changes only are stored
    // bounds_ok: hb describes this shard's previous screen call (same K, gamma)
    if (quad) {
        if (!sm->hb || sm->hb_npad != npad) {
            if (sm->hb) (void)hipFree(sm->hb);
            sm->hb = nullptr;
            sm->hb_valid = false;
            HIP_TRY(hipMalloc((void**)&sm->hb, ((size_t)3 * npad + HB_TAIL) * 4));
            sm->hb_npad = npad;
        }
        if (sm->hb_centers_len < pk) {
            if (sm->hb_centers) (void)hipFree(sm->hb_centers);
            sm->hb_centers = nullptr;
            sm->hb_valid = false;
            HIP_TRY(hipMalloc((void**)&sm->hb_centers, pk * 8));
            sm->hb_centers_len = pk;
        }
        bounds_ok = sm->hb_valid && sm->hb_K == K && sm->hb_gamma == gamma;
        if (!sm->hb_cum) HIP_TRY(hipMalloc((void**)&sm->hb_cum, 16));
        // data in arbitrary order: the previous call (over every point) found most 16-point steps mixing clusters -- the
        // library's order of the points becomes "by cluster" now (regroup_shard; SPKM_NO_REGROUP=1: A/B switch).  Lazy shards
        // only: with the library's order its own, the caller's buffers are reached through a map, which pays while they are
        // written for the points that move and not for all of them.
        if (sm->regroup_wanted && !sm->regroup_done && bounds_ok && sm->lazy && d_mind == nullptr && sm->rec != nullptr && sm->xfs != nullptr &&
            !ctx->sw.no_regroup) {
            if ((rc = regroup_shard<IR>(ctx, sm, K))) return rc;
            sm->regroup_done = true;
        }
        sm->regroup_wanted = false;
        if (!bounds_ok) { // every lower bound is written afresh by this call: the accumulated drift starts over
            HIP_TRY(hipMemsetAsync(sm->hb_cum, 0, 16, ctx->stream));
            sm->cum_par = 0;
        }
        // per-cluster cache / flags of the unchanged-cluster shortcut
        if (!sm->cl_cache || sm->cl_pk != pk || sm->cl_K != K) {
            if (sm->cl_cache) (void)hipFree(sm->cl_cache);
            if (sm->cl_flags) (void)hipFree(sm->cl_flags);
            sm->cl_cache = nullptr; sm->cl_flags = nullptr; sm->cl_valid = false;
            HIP_TRY(hipMalloc((void**)&sm->cl_cache, (2 * pk + 3 * (size_t)K) * 8));
            HIP_TRY(hipMalloc((void**)&sm->cl_flags, (size_t)5 * K * 4));
            sm->cl_pk = pk; sm->cl_K = K;
        }
        if (sm->cl_flags) zero_later(sm->cl_flags + K, (size_t)K * 4); // touched[] (k_combine_screen / k_assign_list mark, k_cluster_need reads)
        // the context's cluster sizes (and, unless the last call was incremental, its sort buffers) still describe this
        // shard's previous screen call
        kept = bounds_ok && ctx->sort_owner == (const void*)sm && ctx->sort_K == K && ctx->sort_n == n;
        // Incremental call (spkm_shard_set_lazy_stats; SPKM_NO_INCREMENTAL=1: A/B switch): no exact pass -- the per-cluster
        // sums are moved by the points that change cluster (events), upper bounds come from the screen's certificate.
        // Needs the caller's permission (lazy, no distances asked for), the library's previous assignment and sums
        // (kept, cl_valid), and pays while not too many points move: the previous counted call saw at most a third of them
        // change (an event pair reads the point twice, through a gather: 0.2 ms per million movers at s = 51 against
        // 10.4 ms for a full pass over 1e8 points); no count yet (a run's second call): taken as few -- at worst every
        // point moves and the events cost what the full pass would have.  Whatever is chosen, the sums are the members' sums.
        // (what an incremental call needs is allocated by the first lazy call, whatever path that one takes: a run's
        //  first call is the one that builds things; 3 x 8 B per point here, and the sort buffers at their event sizes below)
        ev_possible = sm->lazy && !ctx->sw.no_incremental && (size_t)p * 12 <= 64 * 1024;
        // (pair events, below: the bar for "few" is higher.  Only while a pair's run is long enough to pay for its slab --
        //  flushed to BOTH clusters, 4 p atomics per work item -- and for the four extra launches of the second sort level:
        //  at least 256 movers per pair expected, from the previous call's count (n / 3 while there is none).  N = 1e8,
        //  K = 100: 3300 per pair in the iterations that matter; config 3, 6e4 points: never -- 0.28 against 0.16 ms there)
        const unsigned long long est_movers = sm->pol.movers_known ? sm->pol.last_movers : (unsigned long long)n / 3ull;
        // (... and while the second sort level's plan fits this device's LDS: K (K + 1) counters of dynamic LDS beside
        //  k_plan_segments_wide's 8 KB of static arrays -- 74 KB at K = 128, more than a 64-KB part offers from K = 120 on)
        const bool pair_capable = K <= 128 && !ctx->sw.no_pair_events &&
                                  (size_t)K * (size_t)(K + 1) * 4 + 8192 <= ctx->lds_max &&
                                  (est_movers >= 256ull * (unsigned long long)K * (unsigned long long)(K + 1) || ctx->sw.force_pair_events);
        ev_path = ev_possible && d_mind == nullptr && kept && sm->cl_valid && sm->pol.few_movers((double)n, pair_capable) &&
                  !sm->pol.refresh_due((double)n);
        if (ev_possible && sm->ev_cap < (size_t)2 * n) {
            if (sm->ev_pt) (void)hipFree(sm->ev_pt);
            if (sm->ev_k) (void)hipFree(sm->ev_k);
            sm->ev_pt = sm->ev_k = nullptr;
            sm->ev_cap = 0;
            if (hipMalloc((void**)&sm->ev_pt, (size_t)2 * n * 4 + 64) != hipSuccess ||
                hipMalloc((void**)&sm->ev_k, (size_t)2 * n * 4 + 64) != hipSuccess) {
                (void)hipGetLastError();
                if (sm->ev_pt) (void)hipFree(sm->ev_pt);
                sm->ev_pt = sm->ev_k = nullptr;
                ev_path = ev_possible = false; // (no room: the full pass)
            } else
                sm->ev_cap = (size_t)2 * n;
        }
        // PAIR events (K <= 128): one event per mover, sorted by (new, old) pair -- the accumulation reads every mover's
        // record once (k_accumulate_events<.., PAIR>; two events per mover read it twice).  SPKM_NO_PAIR_EVENTS=1: A/B switch
        pair_ev = ev_path && pair_capable;
        // (the plan kernel's dynamic-LDS allowance is raised HERE, before a single event is recorded in the pair format: a
        //  device that refuses it gets two events per mover instead of a failed call)
        if (pair_ev && allow_lds(ctx, (const void*)k_plan_segments_wide, (size_t)K * (size_t)(K + 1) * 4) != hipSuccess) {
            (void)hipGetLastError();
            pair_ev = false;
        }
        if (pair_ev && sm->ev_o_cap < (size_t)n + 4096) {
            if (sm->ev_o) (void)hipFree(sm->ev_o);
            sm->ev_o = nullptr;
            sm->ev_o_cap = 0;
            if (hipMalloc((void**)&sm->ev_o, ((size_t)n + 4096) * 4 + 64) != hipSuccess) {
                (void)hipGetLastError();
                sm->ev_o = nullptr;
                pair_ev = false; // (no room: two events per mover)
            } else
                sm->ev_o_cap = (size_t)n + 4096;
        }
        if (ev_path) {
            if ((rc = ensure(ctx, ctx->nk_ev, (size_t)2 * K * 8))) return rc;
            zero_later(ctx->nk_ev.p, (size_t)2 * K * 8);
        }
        const bool skip_enabled = bounds_ok && !ctx->sw.no_bounds;
        // point-granular list (screen.hip, k_bounds_steps): once the previous call's test passed >= 60 % of the points
        // (counters read back one call late); SPKM_NO_POINT_LIST=1: always 16-point steps (A/B switch)
        pt_mode = skip_enabled && sm->pol.pt_next && !ctx->sw.no_point_list;
        // the two-phase forms' compiled splits (policy.h): the step-major copy lists a point's entries by |x| descending and
        // stops earlier than the point-list kernels, whose entries may come from the records in storage order
        // (the UNCONDITIONAL two-phase form takes the later of the ordered copy's splits, a quarter of the rounds: it finishes
        //  every step on its partial sums, and with the single round of the early split -- 4 entries -- their scatter sends
        //  5 % of a moderately separated shard to the exact list; the hinted form checks before it stops)
        if (prune_a > 0)
            prune_a = (!pt_mode && quad_split_late(q_rounds, false) > 0) ? quad_split_late(q_rounds, false) : quad_split(q_rounds, pt_mode);
        // hinted two-phase form: needs the carried bounds (the hints are ub + drift) and a split that saves rounds
        hinted = want_hint && bounds_ok && prune_a == 0 && quad_split(q_rounds, pt_mode) < q_rounds;
        if (hinted) {
            const bool late = sm->pol.take_hinted_split(q_rounds, ctx->sw.no_late_split) && quad_split_late(q_rounds, pt_mode) > quad_split(q_rounds, pt_mode);
            prune_a = late ? quad_split_late(q_rounds, pt_mode) : quad_split(q_rounds, pt_mode);
            ctx->last_hint_late = late;
            if (sm->hintu_len < npad) {
                if (sm->hintu) (void)hipFree(sm->hintu);
                sm->hintu = nullptr;
                HIP_TRY(hipMalloc((void**)&sm->hintu, (size_t)npad * 4));
                sm->hintu_len = npad;
            }
        }
        if (skip_enabled || hinted) {
            zero_later(sm->hb + 3 * npad + K, 4);
            zero_flush();
            drift_ran = true;
            hipLaunchKernelGGL(k_center_drift, dim3(K), dim3(256), 0, ctx->stream, (const double*)sm->hb_centers,
                               d_centers, K, p, gamma, sm->hb + 3 * npad, sm->cl_flags + 2 * K,
                               ctx->sw.no_support_drift ? 0 : s->fixed_s, 2.0f * (float)s->fixed_s / (float)p,
                               sm->hb + 3 * npad + HB_HTERM);
            // settle the steps (points) the bounds certify, list the others for the screen; write the hints
            // (erode: EVERY lazy call without distances -- an incremental one, and a sums-only full pass, which writes no
            //  upper bound either: a point that passes the test gets no fresh bound from anybody in such a call, so its
            //  bound has to take its centroid's drift here.  A call whose distance pass does run overwrites it again.)
            if ((rc = ensure(ctx, ctx->todo, pt_mode ? (size_t)(npad + 64) * 4 : (size_t)(npad / 16 + 1) * 4))) return rc;
            // (small shards: shorter spans, so that the launch still has >= 8 workgroups per CU)
            // (its statistics leave per workgroup, bstat, and are added up by the call's last kernel: same-address atomics of
            //  a few thousand workgroups took longer than the test itself on small shards)
            long long span = pt_mode ? BOUNDS_SPAN_PT : BOUNDS_SPAN;
            const long long bgrid = 4LL * std::max(1, ctx->num_cus); // (8, 16, 32 per CU measured within noise of 4)
            if ((rc = ensure(ctx, ctx->bstat, (size_t)bgrid * 8))) return rc;
            while (span > 1024 && (npad + span - 1) / span < 4 * bgrid) span /= 2;
            // block summaries: lazy calls only (the only writers of bounds are then this kernel, k_combine_screen and
            // k_assign_list, the latter two for listed points); the lazy contract (spkm.h) lets a settled block's part of the
            // caller's assignment buffer go unvisited as long as it is the buffer of the previous call
            const bool erode = sm->lazy && d_mind == nullptr;
            const bool sp_on = erode && skip_enabled && K <= 128 && sm->pol.blocks_next && !ctx->sw.no_block_skip;
            const long long nblk = npad / 1024 + 1;
            if (sp_on && sm->sp_blocks != nblk) {
                if (sm->sp) (void)hipFree(sm->sp);
                sm->sp = nullptr;
                sm->sp_clean = false;
                HIP_TRY(hipMalloc((void**)&sm->sp, (size_t)nblk * 24));
                sm->sp_blocks = nblk;
            }
            unsigned* sp_mask = sp_on ? reinterpret_cast<unsigned*>(sm->sp) : nullptr;               // 16 B per block
            float* sp_slack = sp_on ? reinterpret_cast<float*>(sm->sp + (size_t)nblk * 16) : nullptr;
            int* sp_valid = sp_on ? reinterpret_cast<int*>(sm->sp + (size_t)nblk * 20) : nullptr;
            const int sp_reset = (sp_on && sm->sp_clean && sm->sp_assign == (const void*)d_assign) ? 0 : 1;
            // (lazy statistics, the buffer of the previous call: it holds the library's copy -- spkm.h -- and is neither read
            //  nor restored by the bounds test, and written by the certification only where a point moves)
            //  -- used for a REGROUPED shard, where every access to the caller's buffer is a scattered one through the map; in
            //  the caller's own order the test keeps repairing a buffer that differs, as it always did)
            trusted = sm->map != nullptr && sm->lazy && d_mind == nullptr && sm->sp_assign == (const void*)d_assign;
            if (sp_on && !sp_reset && ctx->sw.check_assign) {
                // SPKM_CHECK_ASSIGN=1 (debug aid for hosts other than ours): blocks are about to go unvisited on the strength of
                // the lazy contract (spkm.h: the same buffer, not written to between calls) -- compare the caller's buffer
                // with the library's copy of the previous call's assignment first and refuse the call if they differ
                unsigned* cnt = (unsigned*)ctx->nlist.p + 20;
                HIP_TRY(hipMemsetAsync(cnt, 0, 4, ctx->stream));
                hipLaunchKernelGGL(k_count_diff_i32, dim3((unsigned)std::min<long long>(4096, (n + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const int*)d_assign, (const int*)(sm->hb + 2 * npad), n, cnt, (const int*)sm->map);
                unsigned diff = 0;
                HIP_TRY(hipMemcpyAsync(&diff, cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                if (diff) {
                    snprintf(ctx->errmsg, sizeof(ctx->errmsg), "SPKM_CHECK_ASSIGN: d_assign differs from the library's copy of the previous "
                             "call's assignment in %u places (lazy statistics: the buffer is the library's to keep between calls)", diff);
                    sm->sp_clean = false;
                    return SPKM_ERR_BAD_VALUE;
                }
            }
            hipLaunchKernelGGL(sm->map != nullptr ? k_bounds_steps<true> : k_bounds_steps<false>,
                               dim3((unsigned)std::min<long long>((npad + span - 1) / span, bgrid)), dim3(256), 0, ctx->stream, sm->hb, npad, n, K, trusted ? (int*)nullptr : (int*)d_assign, (int*)ctx->todo.p,
                               (unsigned*)ctx->nlist.p, hinted ? sm->hintu : (float*)nullptr, skip_enabled ? 1 : 0,
                               pt_mode ? 1 : 0, (const double*)(sm->hb_cum + sm->cum_par), sm->hb_cum + (sm->cum_par ^ 1),
                               (int)span, (unsigned*)ctx->bstat.p, erode ? 1 : 0, sp_slack, sp_mask, sp_valid, sp_reset,
                               (const int*)(sm->cl_flags + 2 * K), (const int*)sm->map);
            sp_maintained = sp_on;
            bstat_n = (int)std::min<long long>((npad + span - 1) / span, bgrid);
            if (skip_enabled) sm->cum_par ^= 1; // the drift has been added
        }
        skipping = skip_enabled;
        sm->sp_clean = sp_maintained; // (any call that writes bounds without them -- a distance pass, a first call -- starts them over)
        sm->sp_assign = (const void*)d_assign;
        sm->hb_valid = false; // until this call has gone through
    } else
        sm->hb_valid = false;
    zero_flush();
    // (one launch: the f32 tiles, the row-major f64 centres of the exact list, the library's copy for the next call's drift)
    hipLaunchKernelGGL(k_prep_tiles_f32, dim3((unsigned)std::min<size_t>((tile_floats + 255) / 256, 2048)), dim3(256),
                       0, ctx->stream, d_centers, p, K, G, gamma, (float*)ctx->t32.p,
                       (unsigned long long*)ctx->cmax.p, pl_last, quad ? 1 : 0, (double*)ctx->ct.p,
                       quad ? sm->hb_centers : (double*)nullptr);
    ctx->last_sums_only = false;
    // 1. screen
    const int sweep = 16 * 16;
    long long chunk = n / ((long long)(quad ? ctx->bmapq_blocks / 4 : ctx->bmap_streams) * 8);
    chunk = std::max<long long>(sweep, std::min<long long>(chunk, 16 * sweep));
    chunk = (chunk / sweep) * sweep;
    if (quad) { long long c2 = sweep; while (c2 * 2 <= chunk) c2 *= 2; chunk = c2; } // (a power of two: screen_quad.hip's step arithmetic)
    {
        const size_t lds = (size_t)(p + 1) * (SCREEN_KT * 4 + (pl_last == 5 ? 16 : 0)) + 16;
        const void* kern = quad ? screen_quad_kernel<IR>((s->fixed_s + 3) / 4, prune_a > 0 ? prune_a : (s->fixed_s + 3) / 4, pt_mode) : (const void*)k_screen_tile<IR>;
        HIP_TRY(allow_lds(ctx, kern, lds));
        HIP_TRY(timing_begin(ctx));
        const IR* a_ir = quad ? (const IR*)s->irs : (const IR*)s->ir;
        const float* a_xf = quad ? (const float*)s->xfs : (const float*)s->xf;
        const float* a_t = (const float*)ctx->t32.p;
        int a_p = p, a_n = (int)n, a_s = s->fixed_s, a_K = K, a_chunk = (int)chunk;
        const spkm_blockmap* a_bm = (const spkm_blockmap*)(quad ? ctx->bmapq.p : ctx->bmap.p);
        float* a_m1 = (float*)ctx->scr_m1.p;
        float* a_m2 = (float*)ctx->scr_m2.p;
        int* a_k = (int*)ctx->scr_k.p;
        int a_extra = G - 1; // buffer / centroid block of the carried remainder (pl 5)
        // two-phase forms: the split is compiled into the kernel (policy.h)
        const bool two = quad && prune_a > 0 && prune_a < q_rounds; // (prune_a: quad_split or quad_split_late of (q_rounds, pt_mode))
        const int a_rounds = two ? prune_a : q_rounds;
        ctx->last_rounds_all = quad ? a_rounds : 0;
        ctx->last_rounds = quad ? q_rounds : 0;
        const float* a_hint = (hinted && a_rounds < q_rounds) ? sm->hintu : nullptr; // nullptr: every step is finished for the leaders only
        float a_hc = 1.5f; // the other centroids' partial sums must exceed 1.5 x the hinted distance squared
        ctx->last_hinted = a_hint != nullptr;
        unsigned* a_cnt = (unsigned*)ctx->nlist.p;
        const int* a_todo = skipping ? (const int*)ctx->todo.p : nullptr;
        int a_tp = pt_mode ? 1 : 0;
        // point lists: the listed points' entries come from the record layout of the exact pass when this shard has one
        // (built in an earlier call: point lists only appear once most points pass the bounds)
        const char* a_rec = (pt_mode && sm->rec) ? sm->rec : (const char*)nullptr;
        int a_recR = sm->rec_R;
        const int* a_recmap = sm->map;
        void* args[] = {&a_ir, &a_xf, &a_t, &a_p, &a_n, &a_s, &a_K, &a_bm, &a_chunk, &a_m1, &a_m2, &a_k, &a_extra,
                        &a_hint, &a_hc, &a_cnt, &a_todo, &a_tp, &a_rec, &a_recR, &a_recmap};
        HIP_TRY(hipLaunchKernel(kern, dim3(quad ? ctx->bmapq_blocks : ctx->bmap_blocks), dim3(1024), args, lds, ctx->stream));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(timing_end(ctx));
    ctx->last_skipping = skipping;
    ctx->last_pt_mode = pt_mode;
    // what the launch did, for the running totals of executed rounds (k_call_tail; spkm_screen_work_totals)
    const unsigned long long work_steps = (unsigned long long)((n + 15) / 16);
    const int work_tiles = quad ? Gs : 0;
    const int work_flags = ((quad && ctx->last_rounds_all < ctx->last_rounds && !ctx->last_hinted) ? 1 : 0) | (skipping ? 2 : 0) | (pt_mode ? 4 : 0);
    // 4. counting sort by cluster.  When the context still holds the sort of THIS shard's previous screen call (same
    // K, n, segment length; nothing else has written those buffers since) and no assignment changed -- nlist[5],
    // counted on the device by the combine and list kernels against the library's copy of the previous assignment --
    // the histogram, plan and scatter kernels return at once and the previous permutation is used again.
    const int seg = seg_points(n, ctx->num_cus);
    const int max_items = (int)(n / seg) + K + 1;
    // (ev_possible: at the sizes the sort of the EVENTS needs -- 2 per point, 2 K keys, 256-event segments -- from the
    //  start, so that no incremental call has to replace them)
    const int max_items_ev_all = ev_possible ? (int)((2 * n) / 256) + 2 * K + 1 : 0;
    if ((rc = ensure(ctx, ctx->perm, (size_t)(ev_possible ? 2 : 1) * n * 4))) return rc;
    if ((rc = ensure(ctx, ctx->offs, (size_t)((ev_possible ? 2 : 1) * K + 1) * 8))) return rc;
    if ((rc = ensure(ctx, ctx->cursor, (size_t)(ev_possible ? 2 : 1) * K * 8))) return rc;
    if ((rc = ensure(ctx, ctx->items, (size_t)std::max(max_items, max_items_ev_all) * 16))) return rc;
    if ((rc = ensure(ctx, ctx->nitems, 64))) return rc;
    // (the exact pass's geometry is needed here already: the plan below depends on which kernel runs)
    const int threads = 1024;
    const int nw = threads / 64;
    const size_t per_pt = (size_t)(s->fixed_s | 1) * 8;
    const size_t fixed_lds = (size_t)p * 20 + 16;
    // Record layout of the exact entries (build_records): built once per shard on the first screen call, when the device
    // has room for it.  With the points of a cluster scattered over the shard (data in arbitrary order) it takes a third
    // off the exact pass; in cluster-contiguous order it is neutral.  SPKM_NO_REC=1: the two separate arrays.
    if ((rc = build_records<IR>(ctx, sm))) return rc;
    const bool use_rec = sm->rec != nullptr;
    if (!use_rec && (rc = ensure_csc(ctx, s))) return rc;
    // software-pipelined record kernel (k_exact_accumulate_rec): batches of exactly 16 points per wave, columns of up
    // to 64 entries
    const bool pipe = use_rec && s->fixed_s <= 64 &&
                      fixed_lds + (size_t)nw * 16 * per_pt + 1024 <= ctx->lds_max;
    // Unchanged-cluster shortcut (screen.hip, k_cluster_need): clusters whose centroid is bitwise the previous call's and
    // that no point left or entered are not streamed again -- their sums, counts, distances, bounds and statistics are
    // what the previous call produced.  Needs the per-item statistics of the pipelined kernel, the library's copy of
    // the previous assignment (bounds_ok) and a complete cache; not when the caller wants the distances written.
    // SPKM_NO_CLUSTER_SKIP=1: A/B switch.
    const bool cl_on = quad && pipe && sm->cl_cache != nullptr;
    const bool cl_skip = cl_on && bounds_ok && drift_ran && sm->cl_valid && sm->cl_stats_valid && d_mind == nullptr && !ctx->sw.no_cluster_skip && !ev_path;
    // Sums-only full pass (SPKM_NO_SUMS_ONLY=1: A/B switch): a LAZY call that cannot take the event path -- a run's first
    // call, or too many movers -- still has to add up every member, but nobody asked for a distance: the pass leaves the
    // centroid reads, the squared terms and the per-point sums out (k_exact_accumulate_rec<..., DIST = false>); upper bounds
    // come from the screen's certificate as in an incremental call, objective and largest distance are NaN.
    const bool sums_only = cl_on && sm->lazy && d_mind == nullptr && !ev_path && !cl_skip && !ctx->sw.no_sums_only;
    // the certificate writes the upper bounds (k_combine_screen, k_assign_list) -- also for a regrouped shard, whose exact
    // pass walks the records in the caller's order and does not know the library's index of a point
    const bool lazy_ub = ev_path || sums_only || sm->map != nullptr;
    ctx->last_sums_only = sums_only;
    // Form chosen on the device (SPKM_NO_DUAL=1: A/B switch): an incremental call issued WITHOUT a mover count -- a run's
    // second call: the counters come back one call late, and from a random start nearly every point moves -- queues the
    // full sums-only pass as well; k_pick_form, behind k_assign_list, opens one of the two from the number of events
    // (policy.h, few_movers: events while at most a third of the points move).  Round 3 took the events blindly there:
    // 12.1 ms of gathers where the pass takes 8.5 (N = 1e8), 25.0 against 21 ms for a config-5 iteration.
    const bool dual = ev_path && sm->pol.form_on_device() && cl_on && !ctx->sw.no_dual && !ctx->sw.no_sums_only;
    // (pair events: one per mover, so half the count stands for the same third of the points)
    const unsigned ev_cap = dual ? (unsigned)std::min<unsigned long long>(spkm_policy::event_cap((unsigned long long)n, pair_ev), 0xfffffff0ull) : 0xffffffffu;
    ctx->last_dual = dual;
    int* cl_need = cl_on ? sm->cl_flags : nullptr;
    int* cl_touched = cl_on ? sm->cl_flags + K : nullptr;
    int* cl_same = cl_on ? sm->cl_flags + 2 * K : nullptr;
    int* cl_ibeg = cl_on ? sm->cl_flags + 3 * K : nullptr;
    int* cl_icnt = cl_on ? sm->cl_flags + 4 * K : nullptr;
    // (ctx->sort_owner still set: none of the sort buffers was replaced by the ensure() calls above)
    const bool reuse = kept && ctx->sort_owner == (const void*)sm && ctx->sort_perm_valid && ctx->sort_seg == seg &&
                       !ctx->sort_partial;
    const unsigned* gate = reuse ? (const unsigned*)ctx->nlist.p + 5 : (const unsigned*)nullptr;
    // cluster sizes: updated by the points that moved (k_combine_screen / k_assign_list see every change against the
    // library's copy of the previous assignment) instead of a histogram over all points
    const bool nk_incr = kept;
    // 2. certification, 3. exact evaluation of the uncertified points.  Both kernels also keep the library's own copy of
    // the assignment (hb + 2 npad; the caller's buffer may change between calls) up to date IN PLACE -- only they can
    // change an assignment -- and, against the previous call's value, mark the clusters a point left or entered and move
    // the cluster sizes (a separate pass comparing the two arrays used to do that: 0.16 ms per call at N = 1e8)
    int* a_lib = quad ? (int*)(sm->hb + 2 * npad) : (int*)nullptr;
    if ((rc = ensure(ctx, ctx->wgstat, (size_t)4 * 4096 * 4))) return rc; // k_combine_screen's per-workgroup statistics
    const int cb = (int)std::min<long long>(4096, (n + 255) / 256); // (8192+: the cold pass gains 6 %, the short lists of a converged run lose 70 %)
    hipLaunchKernelGGL(k_combine_screen, dim3(cb), dim3(256), ((nk_incr ? (size_t)K : 0) + (ev_path ? (size_t)2 * K : 0)) * 4, ctx->stream, (const float*)ctx->scr_m1.p,
                       (const float*)ctx->scr_m2.p, (const int*)ctx->scr_k.p, n, Gs, (const float*)s->xnr,
                       s->fixed_s, (const unsigned long long*)ctx->cmax.p, (int*)d_assign,
                       (int*)ctx->list.p, (unsigned int*)ctx->nlist.p, quad ? sm->hb : (float*)nullptr, npad,
                       skipping ? 1 : 0, (const int*)ctx->todo.p, pt_mode ? 1 : 0,
                       quad ? (const double*)(sm->hb_cum + sm->cum_par) : (const double*)nullptr,
                       bounds_ok ? 1 : 0, cl_skip ? cl_touched : (int*)nullptr, K,
                       nk_incr ? (unsigned long long*)ctx->nk.p : (unsigned long long*)nullptr,
                       lazy_ub ? 1 : 0, ev_path ? sm->ev_pt : (int*)nullptr, ev_path ? sm->ev_k : (int*)nullptr,
                       ev_path ? (unsigned long long*)ctx->nk_ev.p : (unsigned long long*)nullptr, ev_cap,
                       (unsigned*)ctx->wgstat.p, pair_ev ? sm->ev_o : (int*)nullptr, (const int*)sm->map,
                       (trusted && bounds_ok) ? 1 : 0);
    hipLaunchKernelGGL((k_assign_list<IR>), dim3(std::max(1, ctx->num_cus) * 8), dim3(256), 0, ctx->stream,
                       (const long long*)s->jc, (const IR*)s->ir, (const double*)s->x, (const double*)ctx->ct.p, K,
                       s->fixed_s, (const int*)ctx->list.p, (const unsigned int*)ctx->nlist.p, (int*)d_assign,
                       a_lib, bounds_ok ? 1 : 0, (unsigned*)ctx->nlist.p + 5, cl_skip ? cl_touched : (int*)nullptr,
                       nk_incr ? (unsigned long long*)ctx->nk.p : (unsigned long long*)nullptr,
                       lazy_ub ? sm->hb : (float*)nullptr, ev_path ? sm->ev_pt : (int*)nullptr,
                       ev_path ? sm->ev_k : (int*)nullptr, (unsigned*)ctx->nlist.p,
                       s->x == nullptr ? (const char*)sm->rec : (const char*)nullptr, sm->rec_R,
                       ev_path ? (unsigned long long*)ctx->nk_ev.p : (unsigned long long*)nullptr, ev_cap,
                       (const unsigned*)ctx->wgstat.p, cb, pair_ev ? sm->ev_o : (int*)nullptr, (const int*)sm->map);
    ctx->sort_owner = nullptr; // until this call's sort (or its confirmation) has been queued
    ctx->last_lib_valid = bounds_ok;
    ctx->last_incremental = ev_path;
    ctx->last_direct_events = false;
    ctx->last_pair_events = pair_ev;
    if (ev_path) sm->pol.sums_by_events(); else sm->pol.sums_by_full_pass();
    if (ev_path) {
        // ---- incremental call: the per-cluster sums move by the points that changed cluster; no exact pass ----
        // events (point, key) are sorted by key over 2 K keys (K + k: leaves cluster k; k: enters it) with the same
        // histogram / plan / placement kernels as the points of a full pass, their number read on the device
        const unsigned* ev_n = (const unsigned*)ctx->nlist.p + 16;
        const int K2 = 2 * K;
        // (few events -- a settled run moves a few thousand points per call --: short segments, so that they spread over
        //  more than a handful of workgroups)
        const int seg_ev = (sm->pol.movers_known && sm->pol.last_movers < 100000) ? 256 : SEG_POINTS;
        const int max_items_ev = (int)((2 * n) / seg_ev) + K2 + 1;
        if ((rc = ensure(ctx, ctx->perm, (size_t)2 * n * 4))) return rc;
        if ((rc = ensure(ctx, ctx->offs, (size_t)(K2 + 1) * 8))) return rc;
        if ((rc = ensure(ctx, ctx->cursor, (size_t)K2 * 8))) return rc;
        if ((rc = ensure(ctx, ctx->items, (size_t)max_items_ev * 16))) return rc;
        // (sized by what usually moves, not by the worst case: every kernel strides over the device-side count)
        const long long ev_est = std::max<long long>(4096, (long long)std::min<unsigned long long>(sm->pol.movers_known ? 4 * sm->pol.last_movers + 4096 : (unsigned long long)n, (unsigned long long)2 * n));
        const int hb_ = (int)std::min<long long>(1024, (ev_est + 1023) / 1024);
        // dual: k_pick_form opens the events (gate_ev) or the full pass (gate_full, further down); the events' plan counts
        // its items in nitems[1], the full pass's in nitems[0] -- whichever does not run leaves an empty work list
        const unsigned* gate_ev = dual ? (const unsigned*)ctx->nlist.p + 18 : (const unsigned*)nullptr;
        const unsigned* gate_full = dual ? (const unsigned*)ctx->nlist.p + 19 : (const unsigned*)nullptr;
        int* nitems_ev = (int*)ctx->nitems.p + (dual ? 1 : 0);
        if (dual)
            hipLaunchKernelGGL(k_pick_form, dim3(1), dim3(1), 0, ctx->stream, (unsigned*)ctx->nlist.p, ev_cap, (int*)ctx->nitems.p);
        double* cache_s = sm->cl_cache;
        double* cache_c = cache_s + pk;
        // few movers (the previous call's count is back and small -- a settled run): the events are applied one by one
        // where they were appended, no counting sort (k_events_direct).  Should many points move after all, the kernel
        // still applies them all, only slower than the sorted form would have.  SPKM_NO_DIRECT_EVENTS=1: A/B switch
        const bool direct = !dual && sm->pol.events_direct() && !ctx->sw.no_direct_events;
        ctx->last_direct_events = direct;
        if (direct) {
            if (ctx->tlog_both) HIP_TRY(timing_begin(ctx));
            hipLaunchKernelGGL((k_events_direct<IR>), dim3(1024), dim3(256), 0, ctx->stream, (const char*)sm->rec, sm->rec_R,
                               (const IR*)s->ir, (const double*)s->x, (const int*)sm->ev_pt, (const int*)sm->ev_k, ev_n, p,
                               s->fixed_s, K, cache_s, cache_c, pair_ev ? (const int*)sm->ev_o : (const int*)nullptr);
        } else if (pair_ev) {
            // ---- pair events: two-level counting sort by (new, old), then one slab per run of one pair (update.hip) ----
            const int Kp = K * (K + 1);
            constexpr int CH = 8192; // events per chunk of a new-cluster bucket (the second level's work items)
            const int max_items1 = (int)(n / CH) + K + 1;
            const int max_items2 = (int)(n / seg_ev) + Kp + 1;
            int* perm1 = (int*)ctx->perm.p;          // points, by new cluster
            int* perm2 = (int*)ctx->perm.p + n;      // points, by (new, old) pair
            if ((rc = ensure(ctx, ctx->perm_o, (size_t)n * 4 + 64))) return rc; // old clusters, by new cluster
            if ((rc = ensure(ctx, ctx->offs2, (size_t)(Kp + 1) * 8))) return rc;
            if ((rc = ensure(ctx, ctx->cursor2, (size_t)Kp * 8))) return rc;
            if ((rc = ensure(ctx, ctx->hist2, (size_t)Kp * 8))) return rc;
            if ((rc = ensure(ctx, ctx->items2, (size_t)max_items2 * 16))) return rc;
            if ((rc = ensure(ctx, ctx->items, (size_t)max_items1 * 16))) return rc;
            int* nitems1 = (int*)ctx->nitems.p + 2;
            hipLaunchKernelGGL(k_plan_segments, dim3(1), dim3(256), 0, ctx->stream, (const unsigned long long*)ctx->nk_ev.p, K,
                               CH, (long long*)ctx->offs.p, (unsigned long long*)ctx->cursor.p, (int4*)ctx->items.p,
                               nitems1, gate_ev, (const int*)nullptr, (int*)nullptr, (int*)nullptr,
                               (unsigned long long*)ctx->hist2.p, Kp); // (clears the second level's histogram on the way)
            {
                const size_t sc1 = (size_t)((K + 1) & ~1) * 4 + (size_t)K * 12;
                if (sc1 > 48 * 1024) {
                    (void)allow_lds(ctx, (const void*)k_scatter_by_cluster<true>, sc1);
                    (void)allow_lds(ctx, (const void*)k_scatter_by_cluster<false>, sc1);
                }
                hipLaunchKernelGGL(k_scatter_by_cluster<true>, dim3(hb_), dim3(256), sc1, ctx->stream, (const int*)sm->ev_k, 0LL, K,
                                   (unsigned long long*)ctx->cursor.p, perm1, gate_ev, (const int*)nullptr, ev_n,
                                   (const int*)sm->ev_pt, (const int*)sm->ev_o, (int*)ctx->perm_o.p);
            }
            const int gb = std::min(max_items1, std::max(1, ctx->num_cus) * 8);
            const size_t l2 = (size_t)((K + 2) & ~1) * 4 + (size_t)(K + 1) * 8;
            hipLaunchKernelGGL(k_pair_hist, dim3(gb), dim3(256), l2, ctx->stream, (const int*)ctx->perm_o.p,
                               (const long long*)ctx->offs.p, (const int4*)ctx->items.p, (const int*)nitems1, K,
                               (unsigned long long*)ctx->hist2.p, gate_ev);
            hipLaunchKernelGGL(k_plan_segments_wide, dim3(1), dim3(1024), (size_t)Kp * 4, ctx->stream, (const unsigned long long*)ctx->hist2.p, Kp,
                               seg_ev, (long long*)ctx->offs2.p, (unsigned long long*)ctx->cursor2.p, (int4*)ctx->items2.p,
                               nitems_ev, gate_ev);
            hipLaunchKernelGGL(k_pair_scatter, dim3(gb), dim3(256), l2, ctx->stream, (const int*)perm1, (const int*)ctx->perm_o.p,
                               (const long long*)ctx->offs.p, (const int4*)ctx->items.p, (const int*)nitems1, K,
                               (unsigned long long*)ctx->cursor2.p, perm2, gate_ev);
            const size_t slab = (size_t)p * 12;
            const int ab_ev = (int)std::min<long long>(max_items2, std::max<long long>(std::max(1, ctx->num_cus) * 8, 1));
            if (ctx->tlog_both) HIP_TRY(timing_begin(ctx));
            hipLaunchKernelGGL((k_accumulate_events<IR, true>), dim3(ab_ev), dim3(256), slab, ctx->stream, (const char*)sm->rec,
                               sm->rec_R, (const IR*)s->ir, (const double*)s->x, (const int*)perm2,
                               (const long long*)ctx->offs2.p, (const int4*)ctx->items2.p, (const int*)nitems_ev, p,
                               s->fixed_s, K, cache_s, cache_c);
        } else {
        // (the histogram over the 2 K keys was collected by k_combine_screen / k_assign_list as they appended the events)
        hipLaunchKernelGGL(k_plan_segments, dim3(1), dim3(256), 0, ctx->stream, (const unsigned long long*)ctx->nk_ev.p, K2,
                           seg_ev, (long long*)ctx->offs.p, (unsigned long long*)ctx->cursor.p, (int4*)ctx->items.p,
                           nitems_ev, gate_ev, (const int*)nullptr, (int*)nullptr, (int*)nullptr);
        const size_t sc_lds_ev = (size_t)((K2 + 1) & ~1) * 4 + (size_t)K2 * 12;
        launch_scatter(ctx, hb_, sc_lds_ev, (const int*)sm->ev_k, 0, K2, gate_ev, (const int*)nullptr, ev_n,
                       (const int*)sm->ev_pt);
        const size_t slab = (size_t)p * 12;
        const int ab_ev = (int)std::min<long long>(max_items_ev, std::max<long long>(std::max(1, ctx->num_cus) * 8, 1));
        if (ctx->tlog_both) HIP_TRY(timing_begin(ctx));
        hipLaunchKernelGGL((k_accumulate_events<IR>), dim3(ab_ev), dim3(256), slab, ctx->stream, (const char*)sm->rec,
                           sm->rec_R, (const IR*)s->ir, (const double*)s->x, (const int*)ctx->perm.p,
                           (const long long*)ctx->offs.p, (const int4*)ctx->items.p, (const int*)nitems_ev, p,
                           s->fixed_s, K, cache_s, cache_c);
        }
        if (dual) {
            // ---- ... and the full sums-only pass, for the case that too many points moved: the same kernels, in the same
            // order, as a call that knows it from the start (below); every one of them returns at once unless
            // k_pick_form opened gate_full.  Its sums go to the reduce buffer (zeroed at the top of the call), from there
            // into the cache (every cluster is `fresh`), and the tail hands the cache over as it does after the events.
            if ((rc = ensure(ctx, ctx->blk_obj, (size_t)max_items * 8))) return rc;
            if ((rc = ensure(ctx, ctx->blk_max, (size_t)max_items * 8))) return rc;
            if ((rc = ensure(ctx, ctx->blk_imax, (size_t)max_items * 8))) return rc;
            hipLaunchKernelGGL(k_cluster_need, dim3(1), dim3(256), 0, ctx->stream, cl_touched, (const int*)cl_same, 1, K,
                               (const unsigned long long*)ctx->nk.p, cl_need, (unsigned*)ctx->nlist.p, gate_full);
            hipLaunchKernelGGL(k_plan_segments, dim3(1), dim3(256), 0, ctx->stream, (const unsigned long long*)ctx->nk.p, K,
                               seg, (long long*)ctx->offs.p, (unsigned long long*)ctx->cursor.p, (int4*)ctx->items.p,
                               (int*)ctx->nitems.p, gate_full, (const int*)cl_need, cl_ibeg, cl_icnt);
            const int sb2 = (int)std::max<long long>(std::min<long long>(1024, (n + 1023) / 1024), std::min<long long>(8192, n / 4096));
            launch_scatter(ctx, sb2, (size_t)((K + 1) & ~1) * 4 + (size_t)K * 12, (const int*)d_assign, n, K, gate_full, (const int*)nullptr);
            const void* k3 = (const void*)k_exact_accumulate_rec<IR, 4, true, false>;
            const size_t lds3 = fixed_lds + (size_t)nw * 16 * per_pt;
            HIP_TRY(allow_lds(ctx, k3, lds3));
            const char* a_rec = sm->rec;
            int a_R = sm->rec_R, a_p = p, a_s = s->fixed_s;
            const int* a_perm = (const int*)ctx->perm.p;
            const long long* a_offs = (const long long*)ctx->offs.p;
            const int4* a_items = (const int4*)ctx->items.p;
            const int* a_nitems = (const int*)ctx->nitems.p;
            const double* a_C = d_centers;
            double a_gamma = gamma;
            double* a_mind = nullptr;
            float* a_ub = sm->hb;
            double *a_sums = sums, *a_counts = counts, *a_bo = (double*)ctx->blk_obj.p, *a_bm = (double*)ctx->blk_max.p;
            long long* a_bi = (long long*)ctx->blk_imax.p;
            void* args[] = {&a_rec, &a_R, &a_perm, &a_offs, &a_items, &a_nitems, &a_C, &a_gamma, &a_p, &a_s,
                            &a_mind, &a_ub, &a_sums, &a_counts, &a_bo, &a_bm, &a_bi};
            const int ab2 = std::min(max_items, std::max(1, ctx->num_cus));
            HIP_TRY(hipLaunchKernel(k3, dim3(ab2), dim3(threads), args, lds3, ctx->stream));
            hipLaunchKernelGGL(k_cluster_restore, dim3((unsigned)std::min<size_t>((pk + 255) / 256, 2048)), dim3(256), 0, ctx->stream,
                               (const int*)cl_touched, K, p, sums, counts, cache_s, cache_c, gate_full);
        }
        if (ctx->tlog_both) HIP_TRY(timing_end(ctx));
        HIP_TRY(hipGetLastError());
        // the call's sums and counts ARE the cache (rows that no member stores any more: exactly 0)
        hipLaunchKernelGGL(k_call_tail, dim3((unsigned)std::max<size_t>((K + 255) / 256, std::min<size_t>((pk + 255) / 256, 1024))), dim3(256),
                           0, ctx->stream, (const unsigned long long*)ctx->nk.p, K, nk_f, (const double*)ctx->stats.p, obj2, d_stats,
                           (unsigned long long*)d_nk_u64, (const unsigned*)ctx->bstat.p, bstat_n, (unsigned*)ctx->nlist.p, 1,
                           cache_s, (const double*)cache_c, pk, sums, counts,
                           sm->nlist_pending ? (unsigned*)nullptr : sm->h_nlist_dev, sm->nlist_seq + 1u,
                           work_steps, work_tiles, q_rounds, ctx->last_rounds_all, work_flags);
        HIP_TRY(hipGetLastError());
        sm->hb_K = K;
        sm->hb_gamma = gamma;
        sm->hb_valid = true;
        sm->cl_stats_valid = false; // obj2 / largest distance per cluster were not evaluated
        ctx->sort_owner = sm;       // (the cluster sizes in ctx->nk stay this shard's; its sort buffers do not)
        ctx->sort_K = K;
        ctx->sort_n = n;
        ctx->sort_perm_valid = false;
        ctx->sort_partial = false;
        ctx->last_path = 1;
        return SPKM_OK;
    }
    if (!nk_incr) {
        hipLaunchKernelGGL(k_zero_u64_gated, dim3((K + 255) / 256), dim3(256), 0, ctx->stream,
                           (unsigned long long*)ctx->nk.p, K, gate);
        hipLaunchKernelGGL(k_hist, dim3((unsigned)std::min<long long>(1024, (n + 1023) / 1024)), dim3(256), (size_t)K * 4,
                           ctx->stream, (const int*)d_assign, n, K, (unsigned long long*)ctx->nk.p, gate);
    }
    if (cl_on)
        hipLaunchKernelGGL(k_cluster_need, dim3(1), dim3(256), 0, ctx->stream, cl_touched, (const int*)cl_same,
                           cl_skip ? 0 : 1, K, (const unsigned long long*)ctx->nk.p, cl_need, (unsigned*)ctx->nlist.p);
    // (with the shortcut on the plan is never gated: which clusters need work changes even when no assignment does)
    hipLaunchKernelGGL(k_plan_segments, dim3(1), dim3(256), 0, ctx->stream, (const unsigned long long*)ctx->nk.p, K,
                       seg, (long long*)ctx->offs.p, (unsigned long long*)ctx->cursor.p, (int4*)ctx->items.p,
                       (int*)ctx->nitems.p, cl_on ? (const unsigned*)nullptr : gate, (const int*)cl_need, cl_ibeg, cl_icnt);
    // (two passes over 4 B per point are latency bound: 8192 workgroups at N = 1e8 -- 0.23 -> 0.12 ms against 1024)
    int sb = (int)std::max<long long>(std::min<long long>(1024, (n + 1023) / 1024), std::min<long long>(8192, n / 4096));
    const size_t sc_lds = (size_t)((K + 1) & ~1) * 4 + (size_t)K * 12;
    // (with the shortcut on the scatter is never gated either -- a cluster may need its part of the permutation again
    //  without any assignment having changed -- and places only the points of clusters that will be streamed)
    launch_scatter(ctx, sb, sc_lds, (const int*)d_assign, n, K, cl_on ? (const unsigned*)nullptr : gate, cl_skip ? (const int*)cl_need : (const int*)nullptr);
    ctx->sort_partial = cl_skip;
    if (quad) {
        ctx->sort_owner = sm;
        ctx->sort_K = K;
        ctx->sort_n = n;
        ctx->sort_seg = seg;
        ctx->sort_perm_valid = true;
    }
    // 5. exact distance to the assigned centroid + per-cluster accumulation
    // 1 KB headroom: the kernel also has 384 B of static LDS (per-wave partial statistics)
    int pts = (int)std::min<size_t>(64, (ctx->lds_max - fixed_lds - 1024) / nw / per_pt);
    pts = std::max(8, pts & ~7);
    const size_t lds2 = fixed_lds + (size_t)nw * pts * per_pt;
    const int per_cu = 1;
    // 16 points' loads in flight per wave; 4 waves per SIMD (2 with 512-thread workgroups)
    const void* k2 = use_rec ? (const void*)k_exact_accumulate<IR, 16, 4, false, true> : (const void*)k_exact_accumulate<IR, 16, 4, false, false>;
    HIP_TRY(allow_lds(ctx, (const void*)k2, lds2));
    const int ab = std::min(max_items, std::max(1, ctx->num_cus) * per_cu);
    // statistics: per workgroup (k_exact_accumulate) or per work item (k_exact_accumulate_rec)
    if ((rc = ensure(ctx, ctx->blk_obj, (size_t)std::max(ab, max_items) * 8))) return rc;
    if ((rc = ensure(ctx, ctx->blk_max, (size_t)std::max(ab, max_items) * 8))) return rc;
    if ((rc = ensure(ctx, ctx->blk_imax, (size_t)std::max(ab, max_items) * 8))) return rc;
    if (ctx->tlog_both) HIP_TRY(timing_begin(ctx));
    if (pipe) {
        const void* k3 = sums_only ? (const void*)k_exact_accumulate_rec<IR, 4, true, false> : (const void*)k_exact_accumulate_rec<IR, 4>;
        const size_t lds3 = fixed_lds + (size_t)nw * 16 * per_pt;
        HIP_TRY(allow_lds(ctx, k3, lds3));
        const char* a_rec = sm->rec;
        int a_R = sm->rec_R, a_p = p, a_s = s->fixed_s;
        const int* a_perm = (const int*)ctx->perm.p;
        const long long* a_offs = (const long long*)ctx->offs.p;
        const int4* a_items = (const int4*)ctx->items.p;
        const int* a_nitems = (const int*)ctx->nitems.p;
        const double* a_C = d_centers;
        double a_gamma = gamma;
        double* a_mind = d_mind;
        float* a_ub = (quad && sm->map == nullptr) ? sm->hb : (float*)nullptr; // (a regrouped shard: the certificate wrote them)
        double *a_sums = sums, *a_counts = counts, *a_bo = (double*)ctx->blk_obj.p, *a_bm = (double*)ctx->blk_max.p;
        long long* a_bi = (long long*)ctx->blk_imax.p;
        void* args[] = {&a_rec, &a_R, &a_perm, &a_offs, &a_items, &a_nitems, &a_C, &a_gamma, &a_p, &a_s,
                        &a_mind, &a_ub, &a_sums, &a_counts, &a_bo, &a_bm, &a_bi};
        HIP_TRY(hipLaunchKernel(k3, dim3(ab), dim3(threads), args, lds3, ctx->stream));
    } else {
        const char* a_rec = sm->rec;
        int a_R = sm->rec_R, a_p = p, a_s = s->fixed_s, a_pts = pts;
        const IR* a_ir = (const IR*)s->ir;
        const double* a_x = (const double*)s->x;
        const int* a_perm = (const int*)ctx->perm.p;
        const long long* a_offs = (const long long*)ctx->offs.p;
        const int4* a_items = (const int4*)ctx->items.p;
        const int* a_nitems = (const int*)ctx->nitems.p;
        const double* a_C = d_centers;
        double a_gamma = gamma;
        double* a_mind = d_mind;
        float* a_ub = (quad && sm->map == nullptr) ? sm->hb : (float*)nullptr; // (a regrouped shard: the certificate wrote them)
        double *a_sums = sums, *a_counts = counts, *a_bo = (double*)ctx->blk_obj.p, *a_bm = (double*)ctx->blk_max.p;
        long long* a_bi = (long long*)ctx->blk_imax.p;
        void* args[] = {&a_rec, &a_R, &a_ir, &a_x, &a_perm, &a_offs, &a_items, &a_nitems, &a_C, &a_gamma, &a_p, &a_s, &a_pts,
                        &a_mind, &a_ub, &a_sums, &a_counts, &a_bo, &a_bm, &a_bi};
        HIP_TRY(hipLaunchKernel(k2, dim3(ab), dim3(threads), args, lds2, ctx->stream));
    }
    if (ctx->tlog_both) HIP_TRY(timing_end(ctx));
    if (cl_on) {
        double* cache_s = sm->cl_cache;
        double* cache_c = cache_s + pk;
        double* cl_obj = cache_c + pk;
        double* cl_max = cl_obj + K;
        long long* cl_imax = reinterpret_cast<long long*>(cl_max + K);
        hipLaunchKernelGGL(k_cluster_restore, dim3((unsigned)std::min<size_t>((pk + 255) / 256, 2048)), dim3(256), 0, ctx->stream,
                           (const int*)cl_touched /* = fresh, after k_cluster_need */, K, p, sums, counts, cache_s, cache_c);
        if (!sums_only)
            hipLaunchKernelGGL(k_cluster_stats, dim3(1), dim3(256), 0, ctx->stream, (const int*)cl_need, K, (const int*)cl_ibeg,
                               (const int*)cl_icnt, (const double*)ctx->blk_obj.p, (const double*)ctx->blk_max.p,
                               (const long long*)ctx->blk_imax.p, cl_obj, cl_max, cl_imax, (double*)ctx->stats.p);
        sm->cl_valid = true;
        sm->cl_stats_valid = !sums_only;
    } else {
        sm->cl_valid = false;
        sm->cl_stats_valid = false;
        if (pipe) { // per-item statistics without the per-cluster stage: the items are simply reduced as blocks were
            // (nitems lives on the device; unused slots are not read: reduce over the items the plan emitted)
            hipLaunchKernelGGL(k_reduce_stats_n, dim3(1), dim3(64), 0, ctx->stream, (const double*)ctx->blk_obj.p,
                               (const double*)ctx->blk_max.p, (const long long*)ctx->blk_imax.p, (const int*)ctx->nitems.p,
                               (double*)ctx->stats.p);
        } else
            hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(64), 0, ctx->stream, (const double*)ctx->blk_obj.p,
                               (const double*)ctx->blk_max.p, (const long long*)ctx->blk_imax.p, ab, (double*)ctx->stats.p);
    }
    hipLaunchKernelGGL(k_call_tail, dim3((K + 255) / 256), dim3(256), 0, ctx->stream,
                       (const unsigned long long*)ctx->nk.p, K, nk_f, (const double*)ctx->stats.p, obj2, d_stats,
                       (unsigned long long*)d_nk_u64, (const unsigned*)ctx->bstat.p, bstat_n, (unsigned*)ctx->nlist.p,
                       sums_only ? 1 : 0, (double*)nullptr, (const double*)nullptr, (size_t)0, (double*)nullptr, (double*)nullptr,
                       sm->nlist_pending ? (unsigned*)nullptr : sm->h_nlist_dev, sm->nlist_seq + 1u,
                       work_steps, work_tiles, q_rounds, ctx->last_rounds_all, work_flags);
    HIP_TRY(hipGetLastError());
    if (quad) { // the bounds now describe this call: its centroids are what the next call's drift is measured from
        sm->hb_K = K;
        sm->hb_gamma = gamma;
        sm->hb_valid = true;
    }
    ctx->last_path = 1;
    return SPKM_OK;
}

extern "C" int spkm_assign_accumulate_dev(spkm_ctx* ctx, const spkm_shard* s, uint64_t K64, const double* d_centers,
                                          double gamma, int32_t* d_assign, double* d_mind, double* d_stats,
                                          uint64_t* d_nk_u64, double* d_reduce)
{
    if (!ctx || !s || !d_centers || !d_assign || !d_reduce) return SPKM_ERR_NULL_ARG; // d_mind may be NULL (spkm.h)
    if (K64 == 0 || K64 > 65536) return SPKM_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    spkm_shard* sm = const_cast<spkm_shard*>(s);
    // The screen pays K-fold exact work for every point it cannot certify.  Its counters are copied back
    // asynchronously and looked at one call later (no host sync on the hot path):
    //  * more than 5 % of the points on the exact list: the next 8 calls use the all-exact kernels;
    //  * two-phase screen (partial sums for all centroids, only each tile's leader finished -- screen.hip):
    //    switched on when a plain screen found < 0.2 % of the points with a runner-up within 2.25x of the winner
    //    (converged iterations on separated data), switched off for 16 calls when it listed > 0.5 %.
    if (!sm->h_nlist) {
        HIP_TRY(hipHostMalloc((void**)&sm->h_nlist, 128, hipHostMallocMapped | hipHostMallocCoherent));
        memset(sm->h_nlist, 0, 128);
        HIP_TRY(hipHostGetDevicePointer((void**)&sm->h_nlist_dev, sm->h_nlist, 0));
    }
    if (sm->nlist_pending && __atomic_load_n(sm->h_nlist + SPKM_REPORT_WORDS, __ATOMIC_ACQUIRE) == sm->nlist_seq) {
        sm->nlist_pending = false;
        ctx->last_listed = sm->h_nlist[0];
        spkm_policy_counters c;
        c.listed = sm->h_nlist[0]; c.ambig = sm->h_nlist[1]; c.early = sm->h_nlist[2]; c.skipped = sm->h_nlist[3];
        c.kept = sm->h_nlist[12]; c.movers = sm->h_nlist[14];
        c.full_opened = sm->h_nlist[19] != 0u;
        // data in arbitrary order: the call looked at every point in the library's order and fewer than one in eight of its
        // 16-point steps held one cluster -- the next call regroups the shard first (run_screen).  (Cluster-contiguous data
        // whose first cells cut across its clusters still has a third or more of its steps in one cell; regrouping it by
        // those cells was measured: 37 ms spent, nothing gained in the cold iterations, and settled blocks that hold several
        // clusters pass their summaries less often -- 0.53 against 0.32 ms per converged iteration at N = 1e8.)
        // (not while clusters overlap -- nine points in ten with a runner-up within 2.25x, what the policy calls crowded: their steps are mixed whatever the order,
        //  and stay on the screen whatever their neighbours do)
        if (sm->pend_full && sm->lazy && !sm->regroup_done && s->n >= 4096 &&
            (double)sm->h_nlist[21] < 0.125 * (double)((s->n + 15) / 16) && (double)sm->h_nlist[1] < 0.9 * (double)s->n)
            sm->regroup_wanted = true;
        sm->pol.observe(c, (double)s->n, (int)((K64 + SCREEN_KT - 1) / SCREEN_KT), (s->fixed_s + 3) / 4);
    }
    const spkm_policy::choice ch = sm->pol.next(ctx->sw.no_prune, ctx->sw.no_hint, screen_use_quad(ctx, s));
    const bool cooling = ch.exact;
    if (s->n > 0 && !cooling && screen_eligible(ctx, s, (int)K64)) {
        ctx->ev_valid = false;
        // Hinted two-phase screen: when the unconditional two-phase form is not chosen and hints are not paused, the
        // screen compares the competition's partial sums with per-point upper bounds taken from the carried bounds
        // (run_screen / k_bounds_steps); needs this shard's previous call to have been a screen call.
        const int prune_a = ch.prune_a;
        const bool want_hint = ch.want_hint;
        rc = (s->ir_bits == 16) ? run_screen<unsigned short>(ctx, s, (int)K64, d_centers, gamma, d_assign, d_mind, d_reduce, prune_a, want_hint, d_stats, d_nk_u64)
                                : run_screen<unsigned int>(ctx, s, (int)K64, d_centers, gamma, d_assign, d_mind, d_reduce, prune_a, want_hint, d_stats, d_nk_u64);
        if (rc) return rc;
        ctx->last_mode = ctx->last_hinted ? 2 : (ctx->last_rounds_all < ctx->last_rounds ? 1 : 0);
        if (!sm->nlist_pending) { // (run_screen's k_call_tail was told to report under the number nlist_seq + 1)
            sm->nlist_seq++;
            sm->nlist_pending = true;
            sm->pol.launched(ctx->last_rounds_all, ctx->last_rounds, ctx->last_hinted, ctx->last_hint_late, ctx->last_skipping,
                             ctx->last_lib_valid, ctx->last_incremental, ctx->last_dual);
            sm->pend_full = !ctx->last_skipping && screen_use_quad(ctx, s);
        }
        return SPKM_OK; // (statistics and cluster sizes were handed over by run_screen's last kernel)
    }
    ctx->last_path = 0;
    ctx->last_dual = false;
    sm->sp_clean = false;
    sm->hb_valid = false; // the carried bounds describe the previous SCREEN call only
    if (!d_mind) { // the exact kernels produce the distances on their way to the argmin: park them in scratch
        if ((rc = ensure(ctx, ctx->mscr, (size_t)std::max<uint64_t>(s->n, 1) * 8))) return rc;
        d_mind = (double*)ctx->mscr.p;
    }
    rc = spkm_assign_dev(ctx, s, K64, d_centers, gamma, d_assign, d_mind, d_stats, d_nk_u64);
    if (rc) return rc;
    return spkm_accumulate_dev(ctx, s, K64, d_assign, d_reduce);
}

template <typename IR>
static int run_distances(spkm_ctx* ctx, const spkm_shard* s, int K, const double* d_centers, double gamma,
                         const int32_t* d_assign, double* d_mind, double* d_stats)
{
    const int p = (int)s->p;
    const long long n = (long long)s->n;
    int rc;
    if ((rc = ensure(ctx, ctx->stats, 4 * 8))) return rc;
    // Streaming path: the pipelined exact pass without its sums (the same loads, the same storage-order additions) over a
    // counting sort of d_assign -- the one kept from this shard's last fused call when it still describes d_assign
    // (checked against the library's own copy of that assignment), else one made here (three small kernels).  Needs
    // the record layout; per-item statistics give obj2 / the largest distance / its first index for d_stats.
    const long long npad = (n + 63) / 64 * 64;
    const int threads = 1024, nw = threads / 64;
    const size_t per_pt = (size_t)(s->fixed_s | 1) * 8;
    const size_t fixed_lds = (size_t)p * 20 + 16;
    const bool stream_ok = s->rec && s->fixed_s > 0 && s->fixed_s <= 64 && K <= 16384 &&
                           fixed_lds + (size_t)nw * 16 * per_pt + 1024 <= ctx->lds_max;
    if (stream_ok) {
        bool have_sort = ctx->sort_owner == (const void*)s && ctx->sort_perm_valid && ctx->sort_K == K && ctx->sort_n == n &&
                         s->hb && s->hb_valid && s->hb_npad == npad;
        if (have_sort) {
            if ((rc = ensure(ctx, ctx->nlist, 256))) return rc;
            unsigned* cnt = (unsigned*)ctx->nlist.p + 20;
            HIP_TRY(hipMemsetAsync(cnt, 0, 4, ctx->stream));
            hipLaunchKernelGGL(k_count_diff_i32, dim3((unsigned)std::min<long long>(4096, (n + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const int*)d_assign, (const int*)(s->hb + 2 * npad), n, cnt, (const int*)s->map);
            unsigned diff = 1;
            HIP_TRY(hipMemcpyAsync(&diff, cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream)); // an end-of-run call, not the hot path
            have_sort = diff == 0;
        }
        const int seg = have_sort ? ctx->sort_seg : seg_points(n, ctx->num_cus);
        const int max_items = (int)(n / seg) + K + 1;
        const unsigned long long* nk_src = (const unsigned long long*)ctx->nk.p;
        if (!have_sort) {
            // a counting sort of the caller's assignment: histogram, plan, placement
            ctx->sort_owner = nullptr;
            if ((rc = ensure(ctx, ctx->dn_nk, (size_t)K * 8))) return rc;
            if ((rc = ensure(ctx, ctx->perm, (size_t)n * 4))) return rc;
            if ((rc = ensure(ctx, ctx->offs, (size_t)(K + 1) * 8))) return rc;
            if ((rc = ensure(ctx, ctx->cursor, (size_t)K * 8))) return rc;
            if ((rc = ensure(ctx, ctx->items, (size_t)max_items * 16))) return rc;
            if ((rc = ensure(ctx, ctx->nitems, 64))) return rc;
            HIP_TRY(hipMemsetAsync(ctx->dn_nk.p, 0, (size_t)K * 8, ctx->stream));
            hipLaunchKernelGGL(k_hist, dim3((unsigned)std::min<long long>(1024, (n + 1023) / 1024)), dim3(256), (size_t)K * 4,
                               ctx->stream, (const int*)d_assign, n, K, (unsigned long long*)ctx->dn_nk.p, (const unsigned*)nullptr);
            nk_src = (const unsigned long long*)ctx->dn_nk.p;
        }
        // (a kept plan may cover only the clusters the last call had to process: plan all of them again -- the
        //  permutation and the offsets stand; the scatter cursors it rewrites are used only when placing below)
        hipLaunchKernelGGL(k_plan_segments, dim3(1), dim3(256), 0, ctx->stream, nk_src, K, seg, (long long*)ctx->offs.p,
                           (unsigned long long*)ctx->cursor.p, (int4*)ctx->items.p, (int*)ctx->nitems.p,
                           (const unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, (int*)nullptr);
        if (!have_sort || ctx->sort_partial) { // ... and so may the kept permutation: place every point (again)
            const int sb = (int)std::max<long long>(std::min<long long>(1024, (n + 1023) / 1024), std::min<long long>(8192, n / 4096));
            const size_t sc_lds = (size_t)((K + 1) & ~1) * 4 + (size_t)K * 12;
            launch_scatter(ctx, sb, sc_lds, (const int*)d_assign, n, K, (const unsigned*)nullptr, (const int*)nullptr);
            ctx->sort_partial = false;
        }
        const void* k3 = (const void*)k_exact_accumulate_rec<IR, 4, false>; // (distance-only variant: no sums / counts)
        const size_t lds3 = fixed_lds + (size_t)nw * 16 * per_pt;
        HIP_TRY(allow_lds(ctx, k3, lds3));
        const int ab = std::min(max_items, std::max(1, ctx->num_cus));
        // scratch for the per-item statistics: the plan above emits at most n / seg + K + 1 items for THIS segment length
        if ((rc = ensure(ctx, ctx->blk_dff, (size_t)std::max(max_items, FIN_BLOCKS_MAX) * 24))) return rc;
        const char* a_rec = s->rec;
        int a_R = s->rec_R, a_p = p, a_s = s->fixed_s;
        const int* a_perm = (const int*)ctx->perm.p;
        const long long* a_offs = (const long long*)ctx->offs.p;
        const int4* a_items = (const int4*)ctx->items.p;
        const int* a_nitems = (const int*)ctx->nitems.p;
        const double* a_C = d_centers;
        double a_gamma = gamma;
        double* a_mind = d_mind;
        float* a_ub = nullptr;
        double *a_sums = nullptr, *a_counts = nullptr, *a_bo = (double*)ctx->blk_dff.p, *a_bm = a_bo + max_items;
        long long* a_bi = (long long*)(a_bm + max_items);
        void* args[] = {&a_rec, &a_R, &a_perm, &a_offs, &a_items, &a_nitems, &a_C, &a_gamma, &a_p, &a_s,
                        &a_mind, &a_ub, &a_sums, &a_counts, &a_bo, &a_bm, &a_bi};
        HIP_TRY(hipLaunchKernel(k3, dim3(ab), dim3(threads), args, lds3, ctx->stream));
        if (d_stats) {
            hipLaunchKernelGGL(k_reduce_stats_n, dim3(1), dim3(64), 0, ctx->stream, (const double*)a_bo, (const double*)a_bm,
                               (const long long*)a_bi, (const int*)ctx->nitems.p, (double*)ctx->stats.p);
            HIP_TRY(hipMemcpyAsync(d_stats, ctx->stats.p, 3 * 8, hipMemcpyDeviceToDevice, ctx->stream));
        }
        HIP_TRY(hipGetLastError());
        return SPKM_OK;
    }
    if ((rc = ensure_csc(ctx, s))) return rc;
    if ((rc = ensure(ctx, ctx->ct, (size_t)p * K * 8))) return rc;
    hipLaunchKernelGGL(k_prep_rowmajor, dim3((unsigned)std::min<size_t>(((size_t)p * K + 255) / 256, 2048)), dim3(256), 0,
                       ctx->stream, d_centers, p, K, gamma, (double*)ctx->ct.p);
    hipLaunchKernelGGL((k_point_distances<IR>), dim3(std::max(1, ctx->num_cus) * 8), dim3(256), 0, ctx->stream,
                       (const long long*)s->jc, (const IR*)s->ir, (const double*)s->x, (const double*)ctx->ct.p, K, n,
                       s->fixed_s, (const int*)d_assign, d_mind);
    if (d_stats) {
        // obj2, the largest distance and its first index from the distances just written (fixed reduction order)
        const int cb = (int)std::min<long long>(COMBINE_BLOCKS, (n + 255) / 256);
        if ((rc = ensure(ctx, ctx->blk_obj, (size_t)cb * 8))) return rc;
        if ((rc = ensure(ctx, ctx->blk_max, (size_t)cb * 8))) return rc;
        if ((rc = ensure(ctx, ctx->blk_imax, (size_t)cb * 8))) return rc;
        hipLaunchKernelGGL(k_mind_stats, dim3(cb), dim3(256), 0, ctx->stream, (const double*)d_mind, n, (double*)ctx->blk_obj.p,
                           (double*)ctx->blk_max.p, (long long*)ctx->blk_imax.p);
        hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(64), 0, ctx->stream, (const double*)ctx->blk_obj.p,
                           (const double*)ctx->blk_max.p, (const long long*)ctx->blk_imax.p, cb, (double*)ctx->stats.p);
        HIP_TRY(hipMemcpyAsync(d_stats, ctx->stats.p, 3 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIP_TRY(hipGetLastError());
    return SPKM_OK;
}

extern "C" int spkm_distances_stats_dev(spkm_ctx* ctx, const spkm_shard* s, uint64_t K64, const double* d_centers, double gamma,
                                        const int32_t* d_assign, double* d_mind, double* d_stats)
{
    if (!ctx || !s || !d_centers || !d_assign || !d_mind) return SPKM_ERR_NULL_ARG;
    if (K64 == 0 || K64 > 65536) return SPKM_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->device));
    if (s->n == 0) {
        if (d_stats) HIP_TRY(hipMemsetAsync(d_stats, 0, 3 * 8, ctx->stream));
        return SPKM_OK;
    }
    return s->ir_bits == 16 ? run_distances<unsigned short>(ctx, s, (int)K64, d_centers, gamma, d_assign, d_mind, d_stats)
                            : run_distances<unsigned int>(ctx, s, (int)K64, d_centers, gamma, d_assign, d_mind, d_stats);
}

extern "C" int spkm_distances_dev(spkm_ctx* ctx, const spkm_shard* s, uint64_t K64, const double* d_centers, double gamma,
                                  const int32_t* d_assign, double* d_mind)
{
    return spkm_distances_stats_dev(ctx, s, K64, d_centers, gamma, d_assign, d_mind, nullptr);
}

extern "C" int spkm_exact_pass_points(spkm_ctx* ctx, int64_t info[2])
{
    if (!ctx || !info) return SPKM_ERR_NULL_ARG;
    info[0] = info[1] = 0;
    if (ctx->nlist.p) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        unsigned v[34] = {0};
        HIP_TRY(hipMemcpy(v, ctx->nlist.p, sizeof(v), hipMemcpyDeviceToHost));
        info[0] = (int64_t)(((unsigned long long)v[33] << 32) | v[32]);
        info[1] = v[13];
    }
    return SPKM_OK;
}

extern "C" int spkm_screen_work_totals(spkm_ctx* ctx, int64_t info[2])
{
    if (!ctx || !info) return SPKM_ERR_NULL_ARG;
    info[0] = info[1] = 0;
    if (ctx->nlist.p) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        unsigned long long v[2] = {0ull, 0ull};
        HIP_TRY(hipMemcpy(v, (const unsigned*)ctx->nlist.p + 34, sizeof(v), hipMemcpyDeviceToHost));
        info[0] = (int64_t)v[0];
        info[1] = (int64_t)v[1];
    }
    return SPKM_OK;
}

extern "C" int spkm_last_screen_rounds(spkm_ctx* ctx, int64_t info[2])
{
    if (!ctx || !info) return SPKM_ERR_NULL_ARG;
    info[0] = ctx->last_path == 1 ? ctx->last_rounds_all : 0;
    info[1] = ctx->last_path == 1 ? ctx->last_rounds : 0;
    return SPKM_OK;
}

// Form and counters of the last screen call.  Blocks on the stream (diagnostics, not the hot path).
extern "C" int spkm_last_screen_mode(spkm_ctx* ctx, int64_t info[8])
{
    if (!ctx || !info) return SPKM_ERR_NULL_ARG;
    info[0] = -1;
    info[1] = info[2] = info[3] = info[4] = info[5] = info[6] = info[7] = 0;
    if (ctx->last_path == 1 && ctx->nlist.p) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        unsigned v[10] = {0};
        HIP_TRY(hipMemcpy(v, ctx->nlist.p, 40, hipMemcpyDeviceToHost));
        info[0] = ctx->last_mode;
        for (int j = 0; j < 4; j++) info[1 + j] = v[j];
        info[5] = (int64_t)(((unsigned long long)v[9] << 32) | v[8]);
        // how the call got its sums: 0 full pass with every distance, 2 incremental (events),
        // 3 full pass without distances (sums only)
        info[6] = ctx->last_incremental ? (ctx->last_direct_events ? 4 : 2) : (ctx->last_sums_only ? 3 : 0);
        if (ctx->last_dual) { // both forms were queued: which one the device opened (k_pick_form)
            unsigned f[2] = {0, 0};
            HIP_TRY(hipMemcpy(f, (const unsigned*)ctx->nlist.p + 18, 8, hipMemcpyDeviceToHost));
            info[6] = f[1] ? 3 : 2;
        }
        info[7] = ctx->last_pt_mode ? 2 : 0; // 2: point-granular list
    }
    return SPKM_OK;
}

extern "C" int spkm_last_events_form(spkm_ctx* ctx, int64_t info[2])
{
    if (!ctx || !info) return SPKM_ERR_NULL_ARG;
    const bool inc = ctx->last_path == 1 && ctx->last_incremental;
    info[0] = inc ? (ctx->last_direct_events ? 2 : 1) : 0;
    info[1] = (inc && ctx->last_pair_events) ? 1 : 0;
    return SPKM_OK;
}

// [0] = path of the last spkm_assign_accumulate_dev (0 exact tiles, 1 f32 screen + exact confirmation),
// [1] = number of points the screen could not certify (evaluated exactly over all K).  Blocks on the stream.
extern "C" int spkm_debug_shard_bounds(spkm_ctx* ctx, const spkm_shard* s, float* ub, double* lb, int32_t* lib_assign)
{
    if (!ctx || !s) return SPKM_ERR_NULL_ARG;
    if (!s->hb || !s->hb_valid || !s->hb_cum) return SPKM_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const size_t n = (size_t)s->n, npad = (size_t)s->hb_npad;
    // (a regrouped shard keeps its bounds in its own order: handed out in the caller's, point map[i] <- entry i)
    std::vector<int> map;
    if (s->map) {
        map.resize(n);
        HIP_TRY(hipMemcpy(map.data(), s->map, n * 4, hipMemcpyDeviceToHost));
    }
    auto at = [&](size_t i) { return s->map ? (size_t)map[i] : i; };
    std::vector<float> tmp(n);
    if (ub) {
        HIP_TRY(hipMemcpy(tmp.data(), s->hb, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) ub[at(i)] = tmp[i];
    }
    if (lib_assign) {
        std::vector<int32_t> ta(n);
        HIP_TRY(hipMemcpy(ta.data(), s->hb + 2 * npad, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) lib_assign[at(i)] = ta[i];
    }
    if (lb) {
        double cum = 0.0;
        HIP_TRY(hipMemcpy(tmp.data(), s->hb + npad, n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&cum, s->hb_cum + s->cum_par, 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) lb[at(i)] = (double)tmp[i] - cum;
    }
    return SPKM_OK;
}

extern "C" int spkm_last_path_info(spkm_ctx* ctx, int64_t info[2])
{
    if (!ctx || !info) return SPKM_ERR_NULL_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    info[0] = ctx->last_path;
    info[1] = 0;
    if (ctx->last_path == 1 && ctx->nlist.p) {
        unsigned v = 0;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(&v, ctx->nlist.p, 4, hipMemcpyDeviceToHost));
        info[1] = v;
    }
    return SPKM_OK;
}

static constexpr int FIN_BLOCKS = 256;

extern "C" int spkm_finalize_dev(spkm_ctx* ctx, uint64_t p, uint64_t K, const double* d_reduce, double gamma,
                                 double* d_centers, double* d_out)
{
    return spkm_finalize_impl(ctx, p, K, d_reduce, gamma, d_centers, d_out, false);
}

int spkm_finalize_impl(spkm_ctx* ctx, uint64_t p, uint64_t K, const double* d_reduce, double gamma, double* d_centers,
                       double* d_out, bool to_host)
{
    if (!ctx || !d_reduce || !d_centers || !d_out) return SPKM_ERR_NULL_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t pk = (size_t)p * K;
    int rc;
    if ((rc = ensure(ctx, ctx->blk_dff, FIN_BLOCKS * 8))) return rc;
    if (!ctx->fin_ticket.p) {
        if ((rc = ensure(ctx, ctx->fin_ticket, 64))) return rc;
        HIP_TRY(hipMemsetAsync(ctx->fin_ticket.p, 0, 64, ctx->stream)); // (the kernel's last workgroup resets it after every call)
    }
    const int fb = (int)std::min<size_t>(FIN_BLOCKS, (pk + 255) / 256);
    if (to_host && ctx->h_res_len < 3 + K) { // (pinned host memory the device maps: grown to the largest K seen)
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->h_res) (void)hipHostFree(ctx->h_res);
        ctx->h_res = nullptr;
        ctx->h_res_len = 0;
        const size_t len = 3 + std::max<size_t>(K, 125);
        HIP_TRY(hipHostMalloc((void**)&ctx->h_res, len * 8, hipHostMallocMapped | hipHostMallocCoherent));
        memset(ctx->h_res, 0, len * 8);
        HIP_TRY(hipHostGetDevicePointer((void**)&ctx->h_res_dev, ctx->h_res, 0));
        ctx->h_res_len = len;
        ctx->res_seq = 0ull;
    }
    hipLaunchKernelGGL(k_finalize_centers, dim3(fb), dim3(256), 0, ctx->stream, d_reduce, d_reduce + pk,
                       d_reduce + 2 * pk, (int)p, (int)K, gamma, d_centers, (double*)ctx->blk_dff.p,
                       (unsigned*)ctx->fin_ticket.p, d_out, (const double*)(d_reduce + 2 * pk + K),
                       to_host ? ctx->h_res_dev : (double*)nullptr, to_host ? ++ctx->res_seq : 0ull);
    HIP_TRY(hipGetLastError());
    return SPKM_OK;
}

// ------------------------------------------------------------------------------------------
// k-means++ seeding helpers (private/Arthur_initialization.m:38-69)
// ------------------------------------------------------------------------------------------
extern "C" int spkm_kpp_update_dev(spkm_ctx* ctx, uint64_t n64, const double* d_dist_new, double* d_run, int first_round,
                                   double* d_cum, double* total)
{
    if (!ctx || (n64 && (!d_dist_new || !d_run || !d_cum))) return SPKM_ERR_NULL_ARG;
    if (n64 > 0x7ff00000ull) return SPKM_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->device));
    if (total) *total = 0.0;
    if (n64 == 0) return SPKM_OK;
    const long long n = (long long)n64;
    const int nb = (int)((n + KPP_BLOCK - 1) / KPP_BLOCK);
    int rc;
    if ((rc = ensure(ctx, ctx->tmp_mind, (size_t)(nb + 1) * 8))) return rc;
    double* part = (double*)ctx->tmp_mind.p;
    hipLaunchKernelGGL(k_kpp_min_partial, dim3(nb), dim3(256), 0, ctx->stream, d_dist_new, d_run, n, first_round ? 1 : 0, part);
    hipLaunchKernelGGL(k_kpp_scan_partials, dim3(1), dim3(1), 0, ctx->stream, part, nb);
    hipLaunchKernelGGL(k_kpp_block_scan, dim3(nb), dim3(256), 0, ctx->stream, (const double*)d_run, n, (const double*)part, d_cum);
    HIP_TRY(hipGetLastError());
    if (total) {
        // (the LAST prefix sum, not the sum of the block partials: the two differ in the last bit, and the draw searches d_cum)
        HIP_TRY(hipMemcpyAsync(total, d_cum + (n - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return SPKM_OK;
}

extern "C" int spkm_kpp_draw_dev(spkm_ctx* ctx, uint64_t n64, const double* d_cum, double target, int64_t* index)
{
    if (!ctx || !d_cum || !index) return SPKM_ERR_NULL_ARG;
    if (n64 == 0 || n64 > 0x7ff00000ull) return SPKM_ERR_BAD_VALUE;
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->tmp_assign, 64))) return rc;
    hipLaunchKernelGGL(k_kpp_search, dim3(1), dim3(1), 0, ctx->stream, d_cum, (long long)n64, target, (long long*)ctx->tmp_assign.p);
    HIP_TRY(hipGetLastError());
    long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, ctx->tmp_assign.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *index = (int64_t)v;
    return SPKM_OK;
}

// ------------------------------------------------------------------------------------------
// dense (unsampled) data: two-pass outputs
// ------------------------------------------------------------------------------------------
extern "C" int spkm_dense_assign_dev(spkm_ctx* ctx, uint64_t p64, uint64_t n64, const double* d_X, uint64_t K64,
                                     const double* d_centers, int32_t* d_assign, double* d_dist)
{
    if (!ctx || !d_X || !d_centers || !d_assign || !d_dist) return SPKM_ERR_NULL_ARG;
    if (K64 == 0 || K64 > 65536 || p64 == 0 || p64 > (1u << 24) || n64 > 0x7fffffffull) return SPKM_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n64 == 0) return SPKM_OK;
    const int p = (int)p64, K = (int)K64;
    const long long n = (long long)n64;
    int rc;
    if ((rc = ensure(ctx, ctx->dn_x, (size_t)n * 8))) return rc;
    if ((rc = ensure(ctx, ctx->dn_c, (size_t)K * 8))) return rc;
    const int wb = (int)std::min<long long>(4096, (n + 3) / 4);
    hipLaunchKernelGGL(k_rows_normsq, dim3(wb), dim3(256), 0, ctx->stream, d_X, n, p, (double*)ctx->dn_x.p);
    hipLaunchKernelGGL(k_rows_normsq, dim3((K + 3) / 4), dim3(256), 0, ctx->stream, d_centers, (long long)K, p,
                       (double*)ctx->dn_c.p);
    const size_t lds = (size_t)(DA_PTS + DA_KP) * DA_LD * 8;
    HIP_TRY(hipFuncSetAttribute((const void*)k_dense_assign, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_dense_assign, dim3((unsigned)((n + DA_PTS - 1) / DA_PTS)), dim3(256), lds, ctx->stream, d_X, n, p,
                       d_centers, K, (const double*)ctx->dn_x.p, (const double*)ctx->dn_c.p, (int*)d_assign, d_dist);
    HIP_TRY(hipGetLastError());
    return SPKM_OK;
}

extern "C" int spkm_dense_accumulate_dev(spkm_ctx* ctx, uint64_t p64, uint64_t n64, const double* d_X, uint64_t K64,
                                         const int32_t* d_assign, double* d_sums, double* d_counts)
{
    if (!ctx || !d_X || !d_assign || !d_sums || !d_counts) return SPKM_ERR_NULL_ARG;
    if (K64 == 0 || K64 > 65536 || p64 == 0 || p64 > (1u << 24) || n64 > 0x7fffffffull) return SPKM_ERR_UNSUPPORTED;
    ctx->sort_owner = nullptr; // this call overwrites (some of) the buffers a kept counting sort lives in
    HIP_TRY(hipSetDevice(ctx->device));
    if (n64 == 0) return SPKM_OK;
    const int p = (int)p64, K = (int)K64;
    const long long n = (long long)n64;
    int rc;
    const int seg = 256;
    const int max_items = (int)(n / seg) + K + 1;
    if ((rc = ensure(ctx, ctx->dn_nk, (size_t)K * 8))) return rc;
    if ((rc = ensure(ctx, ctx->perm, (size_t)n * 4))) return rc;
    if ((rc = ensure(ctx, ctx->offs, (size_t)(K + 1) * 8))) return rc;
    if ((rc = ensure(ctx, ctx->cursor, (size_t)K * 8))) return rc;
    if ((rc = ensure(ctx, ctx->items, (size_t)max_items * 16))) return rc;
    if ((rc = ensure(ctx, ctx->nitems, 64))) return rc;
    HIP_TRY(hipMemsetAsync(ctx->dn_nk.p, 0, (size_t)K * 8, ctx->stream));
    hipLaunchKernelGGL(k_hist, dim3((unsigned)std::min<long long>(1024, (n + 1023) / 1024)), dim3(256), (size_t)K * 4,
                       ctx->stream, (const int*)d_assign, n, K, (unsigned long long*)ctx->dn_nk.p, (const unsigned*)nullptr);
    hipLaunchKernelGGL(k_plan_segments, dim3(1), dim3(256), 0, ctx->stream, (const unsigned long long*)ctx->dn_nk.p, K, seg,
                       (long long*)ctx->offs.p, (unsigned long long*)ctx->cursor.p, (int4*)ctx->items.p,
                       (int*)ctx->nitems.p, (const unsigned*)nullptr);
    const int sb = (int)std::min<long long>(1024, (n + 1023) / 1024);
    const size_t sc_lds = (size_t)((K + 1) & ~1) * 4 + (size_t)K * 12;
    launch_scatter(ctx, sb, sc_lds, (const int*)d_assign, n, K, (const unsigned*)nullptr, (const int*)nullptr);
    const int ab = std::min(max_items, std::max(1, ctx->num_cus) * 8);
    hipLaunchKernelGGL(k_dense_accumulate, dim3(ab), dim3(256), 0, ctx->stream, d_X, p, (const int*)ctx->perm.p,
                       (const long long*)ctx->offs.p, (const int4*)ctx->items.p, (const int*)ctx->nitems.p, d_sums);
    hipLaunchKernelGGL(k_nk_add_f64, dim3((K + 255) / 256), dim3(256), 0, ctx->stream,
                       (const unsigned long long*)ctx->dn_nk.p, K, d_counts);
    HIP_TRY(hipGetLastError());
    return SPKM_OK;
}
