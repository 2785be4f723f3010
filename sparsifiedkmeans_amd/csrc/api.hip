// libspkm.so -- C ABI implementation (include/spkm.h), part 1 of 2: contexts, shards, the mex-equivalent host-buffer
// operators, FWHT / mix / sparsifier, the RCCL exchange.  The Lloyd engine proper -- everything that launches the assignment,
// accumulation and screen kernels -- is api_lloyd.hip (its own translation unit: the two compile in parallel with the four
// of screen_quad.hip).  Host side is plain C++17; no torch types.
#include "fwht.hip"
#include "sparse_ops.hip"
#include "sample.hip"

#include "api_internal.h"
#include <chrono>
#include <sched.h>
#include <time.h>

static spkm_switches read_switches()
{
    auto on = [](const char* name) { const char* v = getenv(name); return v != nullptr && *v != 0 && !(v[0] == '0' && v[1] == 0); }; // (unset, empty or "0": off)
    spkm_switches w;
    w.no_screen = on("SPKM_NO_SCREEN");
    w.no_prune = on("SPKM_NO_PRUNE");
    w.no_hint = on("SPKM_NO_HINT");
    w.no_bounds = on("SPKM_NO_BOUNDS");
    w.no_rec = on("SPKM_NO_REC");
    w.no_point_list = on("SPKM_NO_POINT_LIST");
    w.no_cluster_skip = on("SPKM_NO_CLUSTER_SKIP");
    w.no_late_split = on("SPKM_NO_LATE_SPLIT");
    w.no_incremental = on("SPKM_NO_INCREMENTAL");
    w.no_support_drift = on("SPKM_NO_SUPPORT_DRIFT");
    w.no_teams = on("SPKM_NO_TEAMS");
    w.no_sums_only = on("SPKM_NO_SUMS_ONLY");
    w.no_dual = on("SPKM_NO_DUAL");
    w.no_block_skip = on("SPKM_NO_BLOCK_SKIP");
    w.no_direct_events = on("SPKM_NO_DIRECT_EVENTS");
    w.no_pair_events = on("SPKM_NO_PAIR_EVENTS");
    w.force_pair_events = on("SPKM_FORCE_PAIR_EVENTS");
    w.check_assign = on("SPKM_CHECK_ASSIGN");
    w.no_regroup = on("SPKM_NO_REGROUP");
    if (const char* v = getenv("SPKM_X_HINT_CHUNK")) w.x_hint_chunk = atoi(v);   // points per chunk in the two-phase screen launches (default 256)
    if (const char* v = getenv("SPKM_X_PLAIN_CHUNK")) w.x_plain_chunk = atoi(v); // ... in the plain launch (default: n / (8 x teams), at most 4096)
    return w;
}

// hipFuncAttributeMaxDynamicSharedMemorySize, raised at most once per (device, kernel) and size: the call is not
// free and the screen path needs it for two kernels per call.  The attribute belongs to the function, not to a
// context, so the record is process-wide (several contexts may share a device).
hipError_t allow_lds(spkm_ctx* ctx, const void* kern, size_t bytes)
{
    static std::mutex mu;
    static std::vector<std::tuple<int, const void*, size_t>> allowed;
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : allowed)
        if (std::get<0>(e) == ctx->device && std::get<1>(e) == kern) {
            if (std::get<2>(e) >= bytes) return hipSuccess;
            hipError_t r = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (r == hipSuccess) std::get<2>(e) = bytes;
            return r;
        }
    hipError_t r = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (r == hipSuccess) allowed.emplace_back(ctx->device, kern, bytes);
    return r;
}

int ensure(spkm_ctx* ctx, devbuf& b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return SPKM_OK;
    // (a buffer that is replaced may have held the kept counting sort; one that is allocated for the first time cannot --
    //  a context's first call allocates several after its sort is queued, and used to lose the sort for the second call)
    if (b.p) ctx->sort_owner = nullptr;
    if (b.p) HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const size_t want = std::max<size_t>(bytes, 256);
    HIP_TRY(hipMalloc(&b.p, want));
    b.cap = want;
    return SPKM_OK;
}
void release(devbuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

extern "C" const char* spkm_strerror(int status)
{
    switch (status) {
    case SPKM_OK: return "ok";
    case SPKM_ERR_NULL_ARG: return "required pointer argument is NULL";
    case SPKM_ERR_CENTER_ROWS: return "Center vector must be or pxk, but this vector did not have p rows";
    case SPKM_ERR_BETA_K: return "Have not yet implemented case for using 'beta' with p x k (k!=1) centers";
    case SPKM_ERR_LEN_LE_1: return "Vector length must be greater than 1.";
    case SPKM_ERR_NOT_POW2: return "Vector length must be power of 2.";
    case SPKM_ERR_BAD_CSC: return "sparse matrix is not valid CSC (jc not monotone, row index out of range, or rows not ascending)";
    case SPKM_ERR_UNSUPPORTED: return "shape not supported by this build";
    case SPKM_ERR_NO_DEVICE: return "no usable HIP device (gfx950) available";
    case SPKM_ERR_BAD_VALUE: return "scalar argument out of range";
    case SPKM_ERR_COMM: return "RCCL is not available or an RCCL call failed (see spkm_ctx_last_error)";
    default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown spkm status";
}

extern "C" int spkm_version(void) { return SPKM_VERSION; }

extern "C" int spkm_ctx_create(int device, void* stream, spkm_ctx** out)
{
    if (!out) return SPKM_ERR_NULL_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return SPKM_ERR_NO_DEVICE;
    spkm_ctx* ctx = new spkm_ctx();
    ctx->device = device;
    ctx->sw = read_switches();
    ctx->stream = (hipStream_t)stream;
    hipError_t e = hipSetDevice(device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete ctx; return (int)e; }
    ctx->num_cus = prop.multiProcessorCount;
    ctx->mem_bytes = prop.totalGlobalMem;
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || lds <= 0)
        lds = (int)prop.sharedMemPerBlock;
    ctx->lds_max = (size_t)lds;
    e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e != hipSuccess) { delete ctx; return (int)e; }
    *out = ctx;
    return SPKM_OK;
}

extern "C" void spkm_ctx_destroy(spkm_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)spkm_comm_destroy(ctx);
    devbuf* all[] = {&ctx->tiles, &ctx->part_acc, &ctx->part_k, &ctx->blk_obj, &ctx->blk_max, &ctx->blk_imax,
                     &ctx->nk, &ctx->stats, &ctx->perm, &ctx->offs, &ctx->cursor, &ctx->items, &ctx->nitems,
                     &ctx->bmap, &ctx->blk_dff, &ctx->ct, &ctx->tmp_assign, &ctx->tmp_mind, &ctx->mscr, &ctx->dbg, &ctx->t32, &ctx->scr_m1, &ctx->scr_m2,
                     &ctx->scr_k, &ctx->cmax, &ctx->list, &ctx->nlist, &ctx->dn_x, &ctx->dn_c, &ctx->dn_nk, &ctx->bmapq, &ctx->todo, &ctx->bstat, &ctx->nk_ev, &ctx->fin_ticket, &ctx->wgstat, &ctx->offs2, &ctx->cursor2, &ctx->hist2,
                     &ctx->items2, &ctx->perm_o};
    for (devbuf* b : all) release(*b);
    if (ctx->h_res) (void)hipHostFree(ctx->h_res);
    for (auto& pr : ctx->tlog) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

extern "C" int spkm_ctx_reload_switches(spkm_ctx* ctx)
{
    if (!ctx) return SPKM_ERR_NULL_ARG;
    ctx->sw = read_switches();
    return SPKM_OK;
}

extern "C" int spkm_ctx_sync(spkm_ctx* ctx)
{
    if (!ctx) return SPKM_ERR_NULL_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SPKM_OK;
}

extern "C" int spkm_device_info(spkm_ctx* ctx, int64_t info[4])
{
    if (!ctx || !info) return SPKM_ERR_NULL_ARG;
    info[0] = ctx->num_cus;
    info[1] = (int64_t)ctx->lds_max;
    info[2] = (int64_t)ctx->mem_bytes;
    info[3] = SPKM_WAVE;
    return SPKM_OK;
}

// ------------------------------------------------------------------------------------------
// shards
// ------------------------------------------------------------------------------------------
static int validate_csc(uint64_t p, uint64_t n, const uint64_t* jc, const uint64_t* ir)
{
    if (jc[0] != 0) return SPKM_ERR_BAD_CSC;
    for (uint64_t i = 0; i < n; i++) {
        if (jc[i + 1] < jc[i]) return SPKM_ERR_BAD_CSC;
        for (uint64_t j = jc[i]; j < jc[i + 1]; j++) {
            if (ir[j] >= p) return SPKM_ERR_BAD_CSC;
            if (j > jc[i] && ir[j] <= ir[j - 1]) return SPKM_ERR_BAD_CSC;
        }
    }
    return SPKM_OK;
}

extern "C" int spkm_shard_create_host(spkm_ctx* ctx, uint64_t p, uint64_t n, const uint64_t* jc,
                                      const uint64_t* ir, const double* x, spkm_shard** out)
{
    if (!ctx || !out || !jc || (!ir && jc[n]) || (!x && jc[n])) return SPKM_ERR_NULL_ARG;
    *out = nullptr;
    if (p == 0 || p > 0x7fffffffull || n > 0x7ff00000ull) return SPKM_ERR_UNSUPPORTED; // kernels index points with int (+ a step of slack)
    int rc = validate_csc(p, n, jc, ir);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t nnz = jc[n];
    spkm_shard* s = new spkm_shard();
    s->ctx = ctx; s->p = p; s->n = n; s->nnz = nnz; s->owned = true; s->owned_csc = true;
    s->ir_bits = (p <= 65536) ? 16 : 32;
    s->slack = 48;
    if (n > 0 && nnz > 0 && nnz % n == 0) {
        const uint64_t st = nnz / n;
        bool fixed = st <= 0x7fffffffull;
        for (uint64_t i = 0; fixed && i <= n; i++) fixed = (jc[i] == i * st);
        if (fixed) s->fixed_s = (int)st;
    }
    const size_t irb = (size_t)s->ir_bits / 8;
    // +48 entries of slack so that the batch loads of the fixed-stride kernels never leave the allocation
    hipError_t e = hipMalloc((void**)&s->jc, (n + 1) * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc(&s->ir, (nnz + 48) * irb);
    if (e == hipSuccess) e = hipMalloc((void**)&s->x, (nnz + 48) * sizeof(double));
    if (e == hipSuccess) e = hipMemsetAsync(s->ir, 0, (nnz + 48) * irb, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->x, 0, (nnz + 48) * sizeof(double), ctx->stream);
    if (e != hipSuccess) { spkm_shard_destroy(s); return (int)e; }
    std::vector<long long> jcn(n + 1);
    for (uint64_t i = 0; i <= n; i++) jcn[i] = (long long)jc[i];
    e = hipMemcpyAsync(s->jc, jcn.data(), (n + 1) * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    std::vector<unsigned short> ir16;
    std::vector<unsigned int> ir32;
    if (e == hipSuccess && nnz) {
        if (s->ir_bits == 16) {
            ir16.resize(nnz);
            for (uint64_t j = 0; j < nnz; j++) ir16[j] = (unsigned short)ir[j];
            e = hipMemcpyAsync(s->ir, ir16.data(), nnz * 2, hipMemcpyHostToDevice, ctx->stream);
        } else {
            ir32.resize(nnz);
            for (uint64_t j = 0; j < nnz; j++) ir32[j] = (unsigned int)ir[j];
            e = hipMemcpyAsync(s->ir, ir32.data(), nnz * 4, hipMemcpyHostToDevice, ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(s->x, x, nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); // staging vectors die here
    if (e != hipSuccess) { spkm_shard_destroy(s); return (int)e; }
    *out = s;
    return SPKM_OK;
}

// flag stays 1 iff jc[i] == i*st for every i in [0, n]
__global__ void k_check_fixed(const long long* __restrict__ jc, long long n, long long st, int* __restrict__ flag)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += (long long)gridDim.x * blockDim.x)
        if (jc[i] != i * st) *flag = 0;
}

extern "C" int spkm_shard_create_dev(spkm_ctx* ctx, uint64_t p, uint64_t n, uint64_t nnz, const int64_t* d_jc,
                                     const void* d_ir, int ir_bits, const double* d_x, uint64_t capacity,
                                     spkm_shard** out)
{
    if (!ctx || !out || !d_jc || (nnz && (!d_ir || !d_x))) return SPKM_ERR_NULL_ARG;
    *out = nullptr;
    if (p == 0 || p > 0x7fffffffull || n > 0x7ff00000ull) return SPKM_ERR_UNSUPPORTED; // kernels index points with int (+ a step of slack)
    if (ir_bits != 16 && ir_bits != 32) return SPKM_ERR_BAD_VALUE;
    if (ir_bits == 16 && p > 65536) return SPKM_ERR_BAD_VALUE;
    if (capacity < nnz) return SPKM_ERR_BAD_VALUE;
    HIP_TRY(hipSetDevice(ctx->device));
    spkm_shard* s = new spkm_shard();
    s->ctx = ctx; s->p = p; s->n = n; s->nnz = nnz; s->ir_bits = ir_bits;
    s->jc = (long long*)d_jc; s->ir = (void*)d_ir; s->x = (double*)d_x; s->owned = false; s->owned_csc = false;
    s->slack = capacity - nnz;
    if (n > 0 && nnz > 0 && nnz % n == 0 && nnz / n <= 0x7fffffffull) {
        int rc = ensure(ctx, ctx->nitems, 64);
        if (rc) { delete s; return rc; }
        int one = 1, flag = 0;
        hipError_t e = hipMemcpyAsync(ctx->nitems.p, &one, 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_check_fixed, dim3((unsigned)std::min<uint64_t>((n + 256) / 256, 4096)), dim3(256), 0,
                               ctx->stream, (const long long*)d_jc, (long long)n, (long long)(nnz / n),
                               (int*)ctx->nitems.p);
            e = hipMemcpyAsync(&flag, ctx->nitems.p, 4, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { delete s; return (int)e; }
        if (flag) s->fixed_s = (int)(nnz / n);
    }
    *out = s;
    return SPKM_OK;
}

extern "C" void spkm_shard_destroy(spkm_shard* s)
{
    if (!s) return;
    if (s->ctx && s->ctx->sort_owner == s) s->ctx->sort_owner = nullptr;
    if (s->ctx) (void)hipSetDevice(s->ctx->device);
    if (s->ctx) (void)hipStreamSynchronize(s->ctx->stream); // (a queued k_call_tail may still report into h_nlist)
    if (s->xnr) (void)hipFree(s->xnr);
    if (s->xf) (void)hipFree(s->xf);
    if (s->xfs) (void)hipFree(s->xfs);
    if (s->rec && s->rec_owned) (void)hipFree(s->rec);
    if (s->hb_cum) (void)hipFree(s->hb_cum);
    if (s->sp) (void)hipFree(s->sp);
    if (s->cl_cache) (void)hipFree(s->cl_cache);
    if (s->cl_flags) (void)hipFree(s->cl_flags);
    if (s->irs) (void)hipFree(s->irs);
    if (s->hb) (void)hipFree(s->hb);
    if (s->hintu) (void)hipFree(s->hintu);
    if (s->ev_pt) (void)hipFree(s->ev_pt);
    if (s->ev_k) (void)hipFree(s->ev_k);
    if (s->ev_o) (void)hipFree(s->ev_o);
    if (s->map) (void)hipFree(s->map);
    if (s->hb_centers) (void)hipFree(s->hb_centers);
    if (s->h_nlist) (void)hipHostFree(s->h_nlist);
    if (s->owned && s->jc) (void)hipFree(s->jc);
    if (s->owned_csc) {
        if (s->ir) (void)hipFree(s->ir);
        if (s->x) (void)hipFree(s->x);
    }
    delete s;
}

extern "C" int spkm_shard_reset_policy(spkm_shard* s)
{
    if (!s) return SPKM_ERR_NULL_ARG;
    s->pol.reset();
    s->cl_valid = false;
    s->cl_stats_valid = false;
    s->nlist_pending = false; // counters of the last call before the reset say nothing about what comes next
    s->hb_valid = false;
    s->sp_clean = false;
    s->assign_synced = false;
    s->regroup_wanted = false;
    s->regroup_done = false; // (the order a previous run left stays; a new run may ask once more)
    s->pend_full = false;
    return SPKM_OK;
}

extern "C" int spkm_shard_set_lazy_stats(spkm_shard* s, int on)
{
    if (!s) return SPKM_ERR_NULL_ARG;
    s->lazy = on != 0;
    s->assign_synced = false;
    s->sp_clean = false; // (a host that (re)declares its contract starts with every block visited: its buffer may be a new one at an old address)
    return SPKM_OK;
}

extern "C" int spkm_shard_order_info(const spkm_shard* s, int64_t info[2])
{
    if (!s || !info) return SPKM_ERR_NULL_ARG;
    info[0] = s->map != nullptr ? 1 : 0;
    info[1] = s->regroup_done ? 1 : 0;
    return SPKM_OK;
}

extern "C" int spkm_shard_info(const spkm_shard* s, uint64_t* p, uint64_t* n, uint64_t* nnz, int* ir_bits)
{
    if (!s) return SPKM_ERR_NULL_ARG;
    if (p) *p = s->p;
    if (n) *n = s->n;
    if (nnz) *nnz = s->nnz;
    if (ir_bits) *ir_bits = s->ir_bits;
    return SPKM_OK;
}

// One column of a resident shard back on the host: its stored entries (row ids ascending, values), whichever layout
// holds them now (CSC arrays, or the record layout after spkm_shard_release_csc).  'sample' starts, k-means++ centres and
// X(:,iMax) of EmptyAction='singleton' (kmeans_sparsified.m:387,437; Arthur_initialization.m:36,68) need a handful per run.
extern "C" int spkm_shard_get_column_host(spkm_ctx* ctx, const spkm_shard* s, uint64_t col, uint64_t cap, uint64_t* ir_out,
                                          double* x_out, uint64_t* count)
{
    if (!ctx || !s || !count || (cap && (!ir_out || !x_out))) return SPKM_ERR_NULL_ARG;
    if (col >= s->n) return SPKM_ERR_BAD_VALUE;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    long long j0, j1;
    if (s->fixed_s > 0) { j0 = (long long)col * s->fixed_s; j1 = j0 + s->fixed_s; }
    else {
        long long jj[2];
        HIP_TRY(hipMemcpy(jj, s->jc + col, 16, hipMemcpyDeviceToHost));
        j0 = jj[0]; j1 = jj[1];
    }
    const uint64_t cnt = (uint64_t)(j1 - j0);
    *count = cnt;
    if (cnt > cap) return SPKM_ERR_BAD_VALUE; // (count tells the caller how much room the column needs)
    if (cnt == 0) return SPKM_OK;
    const size_t irb = (size_t)s->ir_bits / 8;
    std::vector<unsigned char> raw(cnt * irb);
    if (s->x != nullptr) {
        HIP_TRY(hipMemcpy(x_out, s->x + j0, cnt * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(raw.data(), (const char*)s->ir + (size_t)j0 * irb, cnt * irb, hipMemcpyDeviceToHost));
    } else {
        if (!s->rec) return SPKM_ERR_BAD_VALUE;
        const char* b = s->rec + (size_t)col * (size_t)s->rec_R;
        HIP_TRY(hipMemcpy(x_out, b, cnt * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(raw.data(), b + (size_t)s->fixed_s * 8, cnt * irb, hipMemcpyDeviceToHost));
    }
    for (uint64_t j = 0; j < cnt; j++)
        ir_out[j] = irb == 2 ? (uint64_t)reinterpret_cast<const unsigned short*>(raw.data())[j]
                             : (uint64_t)reinterpret_cast<const unsigned int*>(raw.data())[j];
    return SPKM_OK;
}

extern "C" uint64_t spkm_reduce_len(uint64_t p, uint64_t K) { return 2 * p * K + K + 1; }


extern "C" int spkm_last_assign_kernel_ms(spkm_ctx* ctx, double* ms)
{
    if (!ctx || !ms) return SPKM_ERR_NULL_ARG;
    if (!ctx->ev_valid) return SPKM_ERR_BAD_VALUE;
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, ctx->ev0, ctx->ev1));
    *ms = (double)f;
    return SPKM_OK;
}

// Developer aid: per-workgroup wall-clock stamps of the tiled kernel (start, end; 100 MHz ticks).
// enable allocates the buffer (filled by every later spkm_assign_dev); read copies it out.
extern "C" int spkm_debug_block_times(spkm_ctx* ctx, int enable, int64_t* out, int cap, int* nblocks)
{
    if (!ctx) return SPKM_ERR_NULL_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (enable) {
        const int nb = ctx->num_cus > 0 ? ctx->num_cus : 256;
        int rc = ensure(ctx, ctx->dbg, (size_t)nb * 16);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(ctx->dbg.p, 0, (size_t)nb * 16, ctx->stream));
        return SPKM_OK;
    }
    if (!ctx->dbg.p || !out || !nblocks) return SPKM_ERR_BAD_VALUE;
    const int nb = ctx->num_cus > 0 ? ctx->num_cus : 256;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(out, ctx->dbg.p, (size_t)std::min(nb, cap) * 16, hipMemcpyDeviceToHost));
    *nblocks = nb;
    return SPKM_OK;
}

extern "C" int spkm_timing_log(spkm_ctx* ctx, int enable)
{
    if (!ctx) return SPKM_ERR_NULL_ARG;
    ctx->tlog_on = enable != 0;
    ctx->tlog_both = enable == 2;
    ctx->tlog_used = 0;
    return SPKM_OK;
}

extern "C" int spkm_timing_read(spkm_ctx* ctx, double* ms, int cap, int* count)
{
    if (!ctx || !count || (cap > 0 && !ms)) return SPKM_ERR_NULL_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const int nrec = (int)ctx->tlog_used;
    for (int i = 0; i < nrec && i < cap; i++) {
        float f = 0.f;
        HIP_TRY(hipEventElapsedTime(&f, ctx->tlog[i].first, ctx->tlog[i].second));
        ms[i] = (double)f;
    }
    *count = nrec;
    return SPKM_OK;
}

// ------------------------------------------------------------------------------------------
// Part 3: RCCL inside the library, and the whole iteration in one call
// ------------------------------------------------------------------------------------------
// librccl is bound at run time: a single-GPU process never loads it, and a PyTorch process must use the copy PyTorch
// already loaded (one RCCL, one HIP runtime per process).  Only the five entry points used are resolved; their
// signatures are RCCL's public C API (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce,
// ncclGetErrorString; ncclDouble = 8, ncclSum = 0, NCCL_UNIQUE_ID_BYTES = 128).
namespace {
struct rccl_id { char internal[SPKM_COMM_ID_BYTES]; };
struct rccl_api {
    int (*GetUniqueId)(rccl_id*) = nullptr;
    int (*CommInitRank)(void**, int, rccl_id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    char why[200] = {0};
};
rccl_api& rccl()
{
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        // 1. already in the process (PyTorch's bundled librccl.so, or whatever the host linked): bind through the global
        //    scope.  RTLD_DEFAULT is a null pointer on glibc, so "found there" is a flag of its own, not a handle.
        void* h = RTLD_DEFAULT;
        bool have = dlsym(RTLD_DEFAULT, "ncclAllReduce") != nullptr;
        if (!have) {
            const char* cands[] = {getenv("SPKM_RCCL_PATH"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
            // 2. loaded under one of these names but not exported globally: NOLOAD for EVERY candidate before any real
            //    load (a second, different RCCL beside the one the host already uses is what must not happen)
            for (const char* c : cands) {
                if (have || !c || !*c) continue;
                if (void* q = dlopen(c, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL)) { h = q; have = true; }
            }
            // 3. not in the process at all: load it
            for (const char* c : cands) {
                if (have || !c || !*c) continue;
                if (void* q = dlopen(c, RTLD_NOW | RTLD_GLOBAL)) { h = q; have = true; }
            }
        }
        if (!have) { snprintf(api.why, sizeof(api.why), "librccl not found (set SPKM_RCCL_PATH): %s", dlerror()); return; }
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
        if (!api.ok) snprintf(api.why, sizeof(api.why), "librccl lacks an expected entry point");
    });
    return api;
}
int rccl_fail(spkm_ctx* ctx, const char* what, int r)
{
    if (ctx) {
        rccl_api& a = rccl();
        snprintf(ctx->errmsg, sizeof(ctx->errmsg), "%s: %s", what,
                 r < 0 ? a.why : (a.GetErrorString ? a.GetErrorString(r) : "RCCL error"));
    }
    return SPKM_ERR_COMM;
}
} // namespace

extern "C" const char* spkm_ctx_last_error(spkm_ctx* ctx) { return ctx ? ctx->errmsg : ""; }

extern "C" int spkm_comm_unique_id(uint8_t id[SPKM_COMM_ID_BYTES])
{
    if (!id) return SPKM_ERR_NULL_ARG;
    rccl_api& a = rccl();
    if (!a.ok) return SPKM_ERR_COMM;
    rccl_id u;
    memset(&u, 0, sizeof(u));
    const int r = a.GetUniqueId(&u);
    if (r != 0) return SPKM_ERR_COMM;
    memcpy(id, u.internal, SPKM_COMM_ID_BYTES);
    return SPKM_OK;
}

extern "C" int spkm_comm_init(spkm_ctx* ctx, int nranks, int rank, const uint8_t id[SPKM_COMM_ID_BYTES])
{
    if (!ctx || !id) return SPKM_ERR_NULL_ARG;
    if (nranks < 1 || rank < 0 || rank >= nranks || ctx->comm) return SPKM_ERR_BAD_VALUE;
    rccl_api& a = rccl();
    if (!a.ok) return rccl_fail(ctx, "spkm_comm_init", -1);
    HIP_TRY(hipSetDevice(ctx->device));
    rccl_id u;
    memcpy(u.internal, id, SPKM_COMM_ID_BYTES);
    void* comm = nullptr;
    const int r = a.CommInitRank(&comm, nranks, u, rank);
    if (r != 0 || !comm) return rccl_fail(ctx, "ncclCommInitRank", r ? r : 1);
    ctx->comm = comm;
    ctx->comm_nranks = nranks;
    ctx->comm_rank = rank;
    return SPKM_OK;
}

extern "C" int spkm_comm_destroy(spkm_ctx* ctx)
{
    if (!ctx) return SPKM_ERR_NULL_ARG;
    if (!ctx->comm) return SPKM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    const int r = rccl().CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_nranks = 0;
    ctx->comm_rank = 0;
    return r == 0 ? SPKM_OK : rccl_fail(ctx, "ncclCommDestroy", r);
}

extern "C" int spkm_comm_info(spkm_ctx* ctx, int* nranks, int* rank)
{
    if (!ctx) return SPKM_ERR_NULL_ARG;
    if (nranks) *nranks = ctx->comm ? ctx->comm_nranks : 0;
    if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
    return SPKM_OK;
}

extern "C" int spkm_allreduce_f64_dev(spkm_ctx* ctx, double* d_buf, uint64_t count)
{
    if (!ctx || (count && !d_buf)) return SPKM_ERR_NULL_ARG;
    if (!ctx->comm || count == 0) return SPKM_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const int r = rccl().AllReduce(d_buf, d_buf, (size_t)count, /*ncclDouble*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    return r == 0 ? SPKM_OK : rccl_fail(ctx, "ncclAllReduce", r);
}

extern "C" int spkm_lloyd_iter(spkm_ctx* ctx, const spkm_shard* s, uint64_t K, double* d_centers, double gamma,
                               int unbiased, int32_t* d_assign, double* d_mind, double* d_stats, uint64_t* d_nk_u64,
                               double* d_reduce, double* d_out)
{
    if (!ctx || !s || !d_centers || !d_assign || !d_reduce || !d_out) return SPKM_ERR_NULL_ARG; // d_mind may be NULL
    // findClusters = findClusterAssignments(X, centers, [], SparsityLevel) or (X, centers) (kmeans_sparsified.m:369-373)
    int rc = spkm_assign_accumulate_dev(ctx, s, K, d_centers, unbiased ? gamma : 0.0, d_assign, d_mind, d_stats,
                                        d_nk_u64, d_reduce);
    if (rc) return rc;
    // the ONE exchange of an iteration: 2 p K + K + 1 doubles (1.64 MB at p = 1024, K = 100), latency bound on xGMI
    if ((rc = spkm_allreduce_f64_dev(ctx, d_reduce, spkm_reduce_len(s->p, K)))) return rc;
    return spkm_finalize_dev(ctx, s->p, K, d_reduce, gamma, d_centers, d_out); // :448 scales by SparsityLevel either way
}

// The same iteration, and the host gets what it decides on -- dff^2, obj^2 and the cluster sizes (kmeans_sparsified.m:432,
// 470-487) -- without a device-to-host copy and without synchronising the stream: the finalisation's last workgroup stores
// them into pinned host memory the device maps, followed by the call's sequence number; this function returns when the
// number has arrived.  (A settled iteration of a 1.25e7-point shard is 0.15 ms of kernels; the copy kernel, its launch gap
// and the synchronisation behind it were another 0.03.)
extern "C" int spkm_lloyd_iter_host(spkm_ctx* ctx, const spkm_shard* s, uint64_t K, double* d_centers, double gamma,
                                    int unbiased, int32_t* d_assign, double* d_mind, double* d_stats, uint64_t* d_nk_u64,
                                    double* d_reduce, double* d_out, double* host_out)
{
    if (!ctx || !s || !d_centers || !d_assign || !d_reduce || !d_out || !host_out) return SPKM_ERR_NULL_ARG;
    int rc = spkm_assign_accumulate_dev(ctx, s, K, d_centers, unbiased ? gamma : 0.0, d_assign, d_mind, d_stats,
                                        d_nk_u64, d_reduce);
    if (rc) return rc;
    if ((rc = spkm_allreduce_f64_dev(ctx, d_reduce, spkm_reduce_len(s->p, K)))) return rc;
    if ((rc = spkm_finalize_impl(ctx, s->p, K, d_reduce, gamma, d_centers, d_out, true))) return rc;
    const unsigned long long want = ctx->res_seq;
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(ctx->h_res);
    auto by_copy = [&]() -> int { // the ordinary way: two small copies behind the stream
        HIP_TRY(hipMemcpyAsync(host_out, d_out, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(host_out + 2, d_reduce + 2 * (size_t)s->p * K, (size_t)K * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return SPKM_OK;
    };
    if (ctx->res_map_failed) return by_copy(); // (a platform whose host mapping did not deliver once is not asked again)
    auto t0 = std::chrono::steady_clock::now();
    const auto t_start = t0;
    bool drained = false;
    for (unsigned polls = 1;; polls++) {
        if (__atomic_load_n(q, __ATOMIC_ACQUIRE) == want) break;
        // spin for the first ~50 us (a settled iteration is 0.15 ms of kernels: the answer is usually this close), then let
        // other threads have the core, then sleep in 20-us naps: a cold iteration of a large shard is tens of milliseconds
        if (polls < 2048u) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#elif defined(__aarch64__)
            asm volatile("yield");
#endif
            continue;
        }
        const auto waited = std::chrono::steady_clock::now() - t_start;
        if (waited < std::chrono::microseconds(500)) sched_yield();
        else {
            const struct timespec nap = {0, 20000};
            nanosleep(&nap, nullptr);
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(250)) {
            // a quarter of a second without the number: make sure the stream is still alive (a failed launch would never
            // report), and if it has run dry with the number still missing take the results the ordinary way -- from now on
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) {
                if (drained) {
                    if (__atomic_load_n(q, __ATOMIC_ACQUIRE) == want) break;
                    ctx->res_map_failed = true;
                    return by_copy();
                }
                drained = true;
            } else if (e != hipErrorNotReady) {
                HIP_TRY(e);
            }
            t0 = std::chrono::steady_clock::now();
        }
    }
    memcpy(host_out, ctx->h_res + 1, (size_t)(2 + K) * 8);
    return SPKM_OK;
}

// ------------------------------------------------------------------------------------------
// FWHT / mix
// ------------------------------------------------------------------------------------------
static int check_pow2(uint64_t m)
{
    if (m <= 1) return SPKM_ERR_LEN_LE_1; // hadamard.c:100-102
    if (m & (m - 1)) return SPKM_ERR_NOT_POW2; // hadamard.c:104-110
    return SPKM_OK;
}

static int fwht_launch(spkm_ctx* ctx, uint64_t p_in, uint64_t m, uint64_t n, const double* d_x,
                       const double* d_sign, double premul, double postdiv, double* d_y,
                       const void* gather_ir = nullptr, int gather_bits = 0, int gather_s = 0,
                       double gather_level = 1.0, long long gather_stride = 0)
{
    int rc = check_pow2(m);
    if (rc) return rc;
    if (p_in > m) return SPKM_ERR_BAD_VALUE;
    if (n == 0) return SPKM_OK;
    int logm = 0;
    while ((1ull << logm) < m) logm++;
    const int blocks_cap = std::max(1, ctx->num_cus) * 16;
    if (gather_ir && !(m >= 16 && m <= 16384 && (m + m / 8) * 8 <= ctx->lds_max)) return SPKM_ERR_UNSUPPORTED;
    if (m < 16) {
        hipLaunchKernelGGL(k_fwht_small, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 30)), dim3(256), 0,
                           ctx->stream, d_x, d_y, (int)m, (long long)n, (int)p_in, d_sign, premul, postdiv);
    } else if (m <= 16384 && (m + m / 8) * 8 <= ctx->lds_max) {
        const int T = (int)(m / 16);
        const int threads = std::max(T, 256);
        const int cpb = threads / T;
        const size_t lds = (size_t)cpb * (m + m / 8) * 8;
        HIP_TRY(hipFuncSetAttribute((const void*)k_fwht_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const uint64_t want = (n + cpb - 1) / cpb;
        hipLaunchKernelGGL(k_fwht_lds, dim3((unsigned)std::min<uint64_t>(want, (uint64_t)blocks_cap)), dim3(threads),
                           lds, ctx->stream, d_x, d_y, (int)m, logm, (long long)n, (int)p_in, d_sign, premul,
                           postdiv, cpb, gather_ir, gather_bits, gather_s, gather_level, gather_stride);
    } else {
        hipLaunchKernelGGL(k_fwht_load, dim3(blocks_cap), dim3(256), 0, ctx->stream, d_x, d_y, (long long)m,
                           (long long)n, (long long)p_in, d_sign, premul);
        for (uint64_t bit = 1; bit < m; bit <<= 1)
            hipLaunchKernelGGL(k_fwht_stage, dim3(blocks_cap), dim3(256), 0, ctx->stream, d_y, (long long)m,
                               (long long)n, (long long)bit, (bit << 1 == m) ? postdiv : 0.0);
    }
    HIP_TRY(hipGetLastError());
    return SPKM_OK;
}

extern "C" int spkm_fwht_dev(spkm_ctx* ctx, uint64_t m, uint64_t n, const double* d_x, double* d_y)
{
    if (!ctx || (n && (!d_x || !d_y))) return SPKM_ERR_NULL_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    return fwht_launch(ctx, m, m, n, d_x, nullptr, 1.0, 0.0, d_y);
}

extern "C" int spkm_mix_dev(spkm_ctx* ctx, uint64_t p, uint64_t p2, uint64_t n, const double* d_x,
                            const double* d_sign, double premul, double postdiv, double* d_y)
{
    if (!ctx || (n && (!d_x || !d_y))) return SPKM_ERR_NULL_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    return fwht_launch(ctx, p, p2, n, d_x, d_sign, premul, postdiv, d_y);
}

extern "C" int spkm_mix_sample_dev(spkm_ctx* ctx, uint64_t p, uint64_t p2, uint64_t n, const double* d_x,
                                   const double* d_sign, double premul, double postdiv, uint64_t s, uint64_t seed,
                                   uint64_t col0, void* d_ir_out, int ir_bits, double* d_x_out)
{
    if (!ctx || (n && (!d_x || !d_ir_out || !d_x_out))) return SPKM_ERR_NULL_ARG;
    if (s == 0 || s > p2 || (ir_bits != 16 && ir_bits != 32) || (ir_bits == 16 && p2 > 65536)) return SPKM_ERR_BAD_VALUE;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return SPKM_OK;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)std::max(1, ctx->num_cus) * 16);
    if (ir_bits == 16)
        hipLaunchKernelGGL((k_sample_rows<unsigned short>), dim3(blocks), dim3(256), 0, ctx->stream,
                           (unsigned long long)seed, (long long)col0, (long long)n, (int)p2, (int)s,
                           (unsigned short*)d_ir_out);
    else
        hipLaunchKernelGGL((k_sample_rows<unsigned int>), dim3(blocks), dim3(256), 0, ctx->stream,
                           (unsigned long long)seed, (long long)col0, (long long)n, (int)p2, (int)s,
                           (unsigned int*)d_ir_out);
    HIP_TRY(hipGetLastError());
    // SparsityLevel = small_p / p with p the row count of the mixed matrix (randsample_fixedNumberEntries.m:30-31)
    const double level = (double)s / (double)p2;
    return fwht_launch(ctx, p, p2, n, d_x, d_sign, premul, postdiv, d_x_out, d_ir_out, ir_bits, (int)s, level);
}

extern "C" uint64_t spkm_record_bytes(uint64_t s, int ir_bits)
{
    return (s * (8 + (uint64_t)ir_bits / 8) + 15) / 16 * 16;
}

// The sparsifier writing RECORDS: column c's s values at d_rec + c R, its row ids s * 8 bytes further (R =
// spkm_record_bytes(s, ir_bits)) -- the layout the fused call reads, so that a shard made from them
// (spkm_shard_create_rec_dev) never holds the separate CSC arrays: no second copy of the entries at any time.
extern "C" int spkm_mix_sample_rec_dev(spkm_ctx* ctx, uint64_t p, uint64_t p2, uint64_t n, const double* d_x,
                                       const double* d_sign, double premul, double postdiv, uint64_t s, uint64_t seed,
                                       uint64_t col0, int ir_bits, void* d_rec_out)
{
    if (!ctx || (n && (!d_x || !d_rec_out))) return SPKM_ERR_NULL_ARG;
    if (s == 0 || s > p2 || (ir_bits != 16 && ir_bits != 32) || (ir_bits == 16 && p2 > 65536)) return SPKM_ERR_BAD_VALUE;
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return SPKM_OK;
    const long long R = (long long)spkm_record_bytes(s, ir_bits);
    char* ir0 = (char*)d_rec_out + s * 8;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)std::max(1, ctx->num_cus) * 16);
    if (ir_bits == 16)
        hipLaunchKernelGGL((k_sample_rows<unsigned short>), dim3(blocks), dim3(256), 0, ctx->stream,
                           (unsigned long long)seed, (long long)col0, (long long)n, (int)p2, (int)s, (unsigned short*)ir0, R);
    else
        hipLaunchKernelGGL((k_sample_rows<unsigned int>), dim3(blocks), dim3(256), 0, ctx->stream,
                           (unsigned long long)seed, (long long)col0, (long long)n, (int)p2, (int)s, (unsigned int*)ir0, R);
    HIP_TRY(hipGetLastError());
    const double level = (double)s / (double)p2;
    // (the gather reads column c's ids at ir0 + c R and writes its values at d_rec_out + c R)
    int rc = check_pow2(p2);
    if (rc) return rc;
    if (!(p2 >= 16 && p2 <= 16384 && (p2 + p2 / 8) * 8 <= ctx->lds_max)) return SPKM_ERR_UNSUPPORTED;
    // fwht_launch's gather takes ONE base for ids and ONE for values: the values' base is d_rec_out, the ids' ir0
    return fwht_launch(ctx, p, p2, n, d_x, d_sign, premul, postdiv, (double*)d_rec_out, ir0, ir_bits, (int)s, level, R);
}

extern "C" int spkm_widen_f64_dev(spkm_ctx* ctx, int kind, uint64_t count, const void* d_src, double* d_dst)
{
    if (!ctx || (count && (!d_src || !d_dst))) return SPKM_ERR_NULL_ARG;
    if (kind < 1 || kind > 4) return SPKM_ERR_BAD_VALUE;
    HIP_TRY(hipSetDevice(ctx->device));
    if (count == 0) return SPKM_OK;
    const unsigned blocks = (unsigned)std::min<uint64_t>((count / 2 + 255) / 256 + 1, (uint64_t)std::max(1, ctx->num_cus) * 32);
    hipLaunchKernelGGL(k_widen_f64, dim3(blocks), dim3(256), 0, ctx->stream, d_src, kind, (long long)count, d_dst);
    HIP_TRY(hipGetLastError());
    return SPKM_OK;
}

// ------------------------------------------------------------------------------------------
// Part 1: mex-equivalent host-buffer operators
// ------------------------------------------------------------------------------------------
struct tmpdev {
    void* p = nullptr;
    ~tmpdev() { if (p) (void)hipFree(p); }
};

extern "C" int spkm_hadamard_host(spkm_ctx* ctx, uint64_t m, uint64_t n, const double* x, double* y)
{
    if (!ctx || (n && (!x || !y))) return SPKM_ERR_NULL_ARG;
    int rc = check_pow2(m);
    if (rc) return rc;
    if (n == 0) return SPKM_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    // bounded device slabs so that arbitrarily large host matrices stream through
    const uint64_t slab_cols = std::max<uint64_t>(1, (256ull << 20) / (m * 8));
    tmpdev dx, dy;
    HIP_TRY(hipMalloc(&dx.p, std::min(n, slab_cols) * m * 8));
    HIP_TRY(hipMalloc(&dy.p, std::min(n, slab_cols) * m * 8));
    for (uint64_t c0 = 0; c0 < n; c0 += slab_cols) {
        const uint64_t nc = std::min(slab_cols, n - c0);
        HIP_TRY(hipMemcpyAsync(dx.p, x + c0 * m, nc * m * 8, hipMemcpyHostToDevice, ctx->stream));
        rc = fwht_launch(ctx, m, m, nc, (const double*)dx.p, nullptr, 1.0, 0.0, (double*)dy.p);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(y + c0 * m, dy.p, nc * m * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return SPKM_OK;
}

extern "C" int spkm_hadamard_pthreads_host(spkm_ctx* ctx, uint64_t m, uint64_t n, const double* x, double* y)
{
    return spkm_hadamard_host(ctx, m, n, x, y);
}

extern "C" int spkm_SparseMatrixMinusCluster_host(spkm_ctx* ctx, uint64_t p, uint64_t n, const uint64_t* jc,
                                                  const uint64_t* ir, const double* x, uint64_t c_rows, uint64_t K,
                                                  const double* C, const double* beta, double* dist)
{
    if (!ctx || !jc || !C || (n && K && !dist)) return SPKM_ERR_NULL_ARG;
    if (c_rows != p) return SPKM_ERR_CENTER_ROWS;    // SparseMatrixMinusCluster.c:104-107
    if (beta && K != 1) return SPKM_ERR_BETA_K;      // :119-120
    if (n == 0 || K == 0) return SPKM_OK;
    spkm_shard* s = nullptr;
    int rc = spkm_shard_create_host(ctx, p, n, jc, ir, x, &s);
    if (rc) return rc;
    tmpdev dC, dCt, dD;
    hipError_t e = hipMalloc(&dC.p, p * K * 8);
    if (e == hipSuccess) e = hipMalloc(&dCt.p, p * K * 8);
    if (e == hipSuccess) e = hipMalloc(&dD.p, n * K * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(dC.p, C, p * K * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { spkm_shard_destroy(s); return (int)e; }
    const int blocks = std::max(1, ctx->num_cus) * 8;
    if (beta) {
        if (s->ir_bits == 16)
            hipLaunchKernelGGL((k_dist_beta<unsigned short>), dim3(blocks), dim3(256), 0, ctx->stream,
                               (const long long*)s->jc, (const unsigned short*)s->ir, (const double*)s->x,
                               (const double*)dC.p, *beta, (long long)n, (double*)dD.p);
        else
            hipLaunchKernelGGL((k_dist_beta<unsigned int>), dim3(blocks), dim3(256), 0, ctx->stream,
                               (const long long*)s->jc, (const unsigned int*)s->ir, (const double*)s->x,
                               (const double*)dC.p, *beta, (long long)n, (double*)dD.p);
    } else {
        hipLaunchKernelGGL(k_transpose_centers, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)dC.p, (int)p,
                           (int)K, (double*)dCt.p);
        if (s->ir_bits == 16)
            hipLaunchKernelGGL((k_dist_full<unsigned short>), dim3(blocks), dim3(256), 0, ctx->stream,
                               (const long long*)s->jc, (const unsigned short*)s->ir, (const double*)s->x,
                               (const double*)dCt.p, (int)K, (long long)n, (double*)dD.p);
        else
            hipLaunchKernelGGL((k_dist_full<unsigned int>), dim3(blocks), dim3(256), 0, ctx->stream,
                               (const long long*)s->jc, (const unsigned int*)s->ir, (const double*)s->x,
                               (const double*)dCt.p, (int)K, (long long)n, (double*)dD.p);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(dist, dD.p, n * K * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    spkm_shard_destroy(s);
    return (int)e;
}

extern "C" int spkm_SparseMatrixInnerProduct_host(spkm_ctx* ctx, uint64_t p, uint64_t n, const uint64_t* jc,
                                                  const uint64_t* ir, const double* x, const double* c, double* ip,
                                                  double* nx2)
{
    if (!ctx || !jc || !c || (n && !ip)) return SPKM_ERR_NULL_ARG;
    if (n == 0) return SPKM_OK;
    spkm_shard* s = nullptr;
    int rc = spkm_shard_create_host(ctx, p, n, jc, ir, x, &s);
    if (rc) return rc;
    tmpdev dc, dip, dn;
    hipError_t e = hipMalloc(&dc.p, p * 8);
    if (e == hipSuccess) e = hipMalloc(&dip.p, n * 8);
    if (e == hipSuccess) e = hipMalloc(&dn.p, n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(dc.p, c, p * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { spkm_shard_destroy(s); return (int)e; }
    const int blocks = (int)std::min<uint64_t>((n + 255) / 256, (uint64_t)std::max(1, ctx->num_cus) * 16);
    if (s->ir_bits == 16)
        hipLaunchKernelGGL((k_innerprod<unsigned short>), dim3(blocks), dim3(256), 0, ctx->stream,
                           (const long long*)s->jc, (const unsigned short*)s->ir, (const double*)s->x,
                           (const double*)dc.p, (long long)n, (double*)dip.p, (double*)dn.p);
    else
        hipLaunchKernelGGL((k_innerprod<unsigned int>), dim3(blocks), dim3(256), 0, ctx->stream,
                           (const long long*)s->jc, (const unsigned int*)s->ir, (const double*)s->x,
                           (const double*)dc.p, (long long)n, (double*)dip.p, (double*)dn.p);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ip, dip.p, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && nx2) e = hipMemcpyAsync(nx2, dn.p, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    spkm_shard_destroy(s);
    return (int)e;
}

extern "C" int spkm_SparseMatrixColumnNormSq_host(spkm_ctx* ctx, uint64_t n, const uint64_t* jc, const double* x,
                                                  double* nx2)
{
    if (!ctx || !jc || (n && !nx2)) return SPKM_ERR_NULL_ARG;
    if (n == 0) return SPKM_OK;
    for (uint64_t i = 0; i < n; i++) if (jc[i + 1] < jc[i]) return SPKM_ERR_BAD_CSC;
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t nnz = jc[n];
    if (nnz && !x) return SPKM_ERR_NULL_ARG;
    tmpdev djc, dx, dn;
    HIP_TRY(hipMalloc(&djc.p, (n + 1) * 8));
    HIP_TRY(hipMalloc(&dx.p, std::max<uint64_t>(nnz, 1) * 8));
    HIP_TRY(hipMalloc(&dn.p, n * 8));
    HIP_TRY(hipMemcpyAsync(djc.p, jc, (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream)); // u64 == i64 bits here
    if (nnz) HIP_TRY(hipMemcpyAsync(dx.p, x, nnz * 8, hipMemcpyHostToDevice, ctx->stream));
    const int blocks = (int)std::min<uint64_t>((n + 255) / 256, (uint64_t)std::max(1, ctx->num_cus) * 16);
    hipLaunchKernelGGL(k_colnormsq, dim3(blocks), dim3(256), 0, ctx->stream, (const long long*)djc.p,
                       (const double*)dx.p, (long long)n, (double*)dn.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(nx2, dn.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SPKM_OK;
}

