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
    if (n) {
        // the 256 bytes of slack behind the last record (spkm.h) are checked against the allocation the pointer lies in: no
        // size crosses this boundary, but the runtime knows it.  (A pointer it does not know -- another runtime's -- is let through.)
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)const_cast<void*>(d_rec)) == hipSuccess && base != nullptr) {
            const size_t off = (size_t)((const char*)d_rec - (const char*)base);
            const size_t need = (size_t)n * (size_t)spkm_record_bytes(s_entries, ir_bits) + 256;
            if (off > size || size - off < need) {
                snprintf(ctx->errmsg, sizeof(ctx->errmsg), "spkm_shard_create_rec_dev: the allocation ends %zu bytes after d_rec, n records + 256 bytes "
                         "of slack need %zu (spkm.h)", off > size ? (size_t)0 : size - off, need);
                return SPKM_ERR_BAD_VALUE;
            }
        } else
            (void)hipGetLastError();
    }
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
    // (scratch and the new arrays belong to this guard until the very end: every early return -- ensure(), HIP_TRY -- frees them)
    struct Owned {
        int *keys = nullptr, *perm = nullptr, *newmap = nullptr;
        float* hb_new = nullptr;
        ~Owned()
        {
            if (keys) (void)hipFree(keys);
            if (perm) (void)hipFree(perm);
            if (newmap) (void)hipFree(newmap);
            if (hb_new) (void)hipFree(hb_new);
        }
    } own;
    int*& keys = own.keys;
    int*& perm = own.perm;
    int*& newmap = own.newmap;
    float*& hb_new = own.hb_new;
    auto fail = [&](hipError_t e, const char* what) {
        (void)hipGetLastError();
        if (e == hipErrorOutOfMemory) return (int)SPKM_OK; // (no room: the shard simply stays as it is)
        snprintf(ctx->errmsg, sizeof(ctx->errmsg), "regroup_shard: %s: %s", what, hipGetErrorString(e));
        return (int)e;                                     // (positive status = HIP error, as HIP_TRY returns it)
    };
    hipError_t e;
    if ((e = hipMalloc((void**)&keys, (size_t)n * 4 + 64)) != hipSuccess) return fail(e, "hipMalloc(keys)");
    if ((e = hipMalloc((void**)&perm, (size_t)n * 4 + 64)) != hipSuccess) return fail(e, "hipMalloc(perm)");
    if ((e = hipMalloc((void**)&newmap, (size_t)n * 4 + 64)) != hipSuccess) return fail(e, "hipMalloc(map)");
    if ((e = hipMalloc((void**)&hb_new, ((size_t)3 * npad + HB_TAIL) * 4)) != hipSuccess) return fail(e, "hipMalloc(bounds)");
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
    (void)hipFree(sm->hb);
    if (sm->map) (void)hipFree(sm->map);
    sm->hb = hb_new;
    sm->map = newmap;
    hb_new = nullptr; // (the shard's from here on; keys / perm go with the guard)
    newmap = nullptr;
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
    const_cast<spkm_shard*>(s)->assign_synced = false;
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
    const_cast<spkm_shard*>(s)->assign_synced = false;
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

#include "api_lloyd_fused.inc" // the fused call: screen_use_quad, screen_eligible, run_screen, spkm_assign_accumulate_dev

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
