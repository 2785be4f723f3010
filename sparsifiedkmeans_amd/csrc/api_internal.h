// Internals shared by the translation units of libspkm.so's host side (api.hip: contexts, shards, the mex-equivalent
// operators, FWHT / sparsifier, RCCL; api_lloyd.hip: everything that launches the assignment / accumulation / screen kernels).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/spkm.h"
#include "policy.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <mutex>
#include <tuple>
#include <vector>

#define SPKM_VERSION 100

struct devbuf {
    void* p = nullptr;
    size_t cap = 0;
};

// A/B switches.  None changes an output; each turns one work-saving layer off so that its share
// can be measured and so that the tests can hold every layer against the all-exact kernels.  Read from the environment
// ONCE, when the context is created (spkm_ctx_reload_switches re-reads them: tests and A/B tools that toggle a switch
// inside one process).
struct spkm_switches {
    bool no_screen = false;       // SPKM_NO_SCREEN: all-exact f64 kernels instead of screen + confirmation
    bool no_prune = false;        // SPKM_NO_PRUNE: never a two-phase form
    bool no_hint = false;         // SPKM_NO_HINT: no hinted two-phase form
    bool no_bounds = false;       // SPKM_NO_BOUNDS: carried bounds are maintained but nothing is skipped on them
    bool no_rec = false;          // SPKM_NO_REC: no record layout (the exact pass reads the two separate arrays)
    bool no_point_list = false;   // SPKM_NO_POINT_LIST: the carried bounds always settle whole 16-point steps
    bool no_cluster_skip = false; // SPKM_NO_CLUSTER_SKIP: every cluster is planned, placed and streamed in every call
    int x_hint_chunk = 0;         // SPKM_X_HINT_CHUNK: chunk size of the two-phase screen launches (0: the default, 256 points)
    int x_plain_chunk = 0;        // SPKM_X_PLAIN_CHUNK: ... of the plain launch (0: the default)
    bool no_late_split = false;   // SPKM_NO_LATE_SPLIT: the hinted screen always asks after a quarter of the rounds
    bool no_incremental = false;  // SPKM_NO_INCREMENTAL: per-cluster sums are always re-accumulated over every member
    bool no_support_drift = false; // SPKM_NO_SUPPORT_DRIFT: centroid drift by its full 2-norm, not its s largest entries
    bool no_sums_only = false;    // SPKM_NO_SUMS_ONLY: a lazy call's full pass still evaluates every point's distance
    bool no_block_skip = false;   // SPKM_NO_BLOCK_SKIP: the carried-bounds test reads every point (no per-block summaries)
    bool no_dual = false;         // SPKM_NO_DUAL: a run's second lazy call takes the events whatever moves (round 3) instead of deciding on the device
    bool no_pair_events = false;   // SPKM_NO_PAIR_EVENTS: two events per mover over 2 K keys (each applied on its own: the record is read twice) also for K <= 128
    bool force_pair_events = false; // SPKM_FORCE_PAIR_EVENTS: pair events also when few movers per pair are expected (tests)
    bool no_direct_events = false; // SPKM_NO_DIRECT_EVENTS: a lazy call with few movers still sorts its events (plan, placement, slab kernel)
    bool no_teams = false;        // SPKM_NO_TEAMS: screen workgroups split over the tiles by cost (tiles drift apart) instead of teams
    bool no_regroup = false;      // SPKM_NO_REGROUP: the library's order of the points stays the caller's whatever their steps look like
    bool check_assign = false;    // SPKM_CHECK_ASSIGN: before blocks are skipped, verify the lazy contract on d_assign (debug aid; syncs)
};
struct spkm_ctx {
    int device = 0;
    spkm_switches sw;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    size_t lds_max = 0;
    size_t mem_bytes = 0;
    // grow-only device scratch
    devbuf tiles, part_acc, part_k, blk_obj, blk_max, blk_imax, nk, stats, perm, offs, cursor, items, nitems,
        bmap, blk_dff, ct, tmp_assign, tmp_mind, mscr, dbg, t32, scr_m1, scr_m2, scr_k, cmax, list, nlist, dn_x, dn_c, dn_nk, bmapq, todo, bstat, nk_ev, fin_ticket, wgstat, offs2, cursor2, hist2, items2, perm_o;
    // results of an iteration handed to the host without a copy or a stream synchronisation (spkm_lloyd_iter_host): pinned host
    // memory the device maps -- [sequence number | dff^2 | obj^2 | cluster sizes], written by k_finalize_centers' last workgroup
    double* h_res = nullptr;
    double* h_res_dev = nullptr; // the same memory as the device addresses it
    size_t h_res_len = 0;        // doubles
    unsigned long long res_seq = 0ull;
    bool res_map_failed = false;      // spkm_lloyd_iter_host: the mapped host memory did not deliver once -- results by copy from then on
    // cached launch geometry of the tiled kernel
    int bmap_G = -1, bmap_blocks = 0, bmap_streams = 0;
    int bmapq_key = -1, bmapq_blocks = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    // optional per-launch timing log of the dominant assignment kernel (bench.py)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tlog;
    size_t tlog_used = 0;
    bool tlog_on = false;
    // counting sort kept from the last screen call (perm / offs / items / nitems / nk describe THAT call's assignment):
    // whose shard and shape it was; cleared by everything else that writes those buffers
    const void* sort_owner = nullptr;
    bool sort_partial = false; // the kept permutation covers only the clusters the last exact pass had to stream
    int sort_K = 0, sort_seg = 0;
    long long sort_n = 0;
    bool tlog_both = false; // fused screen path: log the exact accumulation kernel too (pairs alternate)
    int assign_KT = 0, assign_G = 0; // of the last assign call
    int last_path = 0;               // 0 = exact tiled/generic, 1 = f32 screen + exact confirmation
    unsigned last_listed = 0;        // points sent to the exact list by the last screen (read lazily)
    bool last_hint_late = false; // the last hinted call used the late split
    int last_rounds_all = 0, last_rounds = 0; // rounds for all centroids / total rounds of the last 4-lane screen call
    bool sort_perm_valid = false;    // ... and perm / offs / items really hold that call's counting sort (not after an incremental call)
    bool last_lib_valid = false;     // the last screen call could compare with the library's previous assignment (movers counted)
    bool last_incremental = false;   // the last screen call updated the sums by events (no exact pass)
    bool last_sums_only = false;     // the last screen call's full pass left the distances out (lazy statistics)
    bool last_dual = false;          // the last screen call queued both forms; the device chose (counters[19]: the full pass)
    int last_mode = 0;               // 0 plain screen, 1 two-phase, 2 hinted two-phase (last screen call)
    bool last_skipping = false;      // the last screen call ran the carried-bounds test
    bool last_direct_events = false; // ... applied its events one by one (k_events_direct)
    bool last_pair_events = false;   // ... recorded one event per mover (pair events)
    bool last_pt_mode = false;       // ... and listed points instead of 16-point steps
    bool last_hinted = false;        // ... used the hinted two-phase form
    char errmsg[256] = {0};
    // data-parallel exchange: an RCCL communicator bound to this context's device and stream (Part 3 of spkm.h)
    void* comm = nullptr; // ncclComm_t
    int comm_nranks = 0, comm_rank = 0;
};

struct spkm_shard {
    spkm_ctx* ctx = nullptr;
    uint64_t p = 0, n = 0, nnz = 0;
    int ir_bits = 32;
    long long* jc = nullptr;
    void* ir = nullptr;
    double* x = nullptr;
    bool owned = false;     // jc is the library's
    bool owned_csc = false; // ir / x are the library's
    int fixed_s = 0;   // > 0: every column has exactly this many entries
    uint64_t slack = 0; // entries readable past nnz in ir / x
    float* xfs = nullptr;  // screen copy for the 4-lanes-per-point kernel: f32 values, columns partitioned by row parity
    void* irs = nullptr;   // ... and their row ids
    bool norms_done = false, xf_done = false;
    float* xnr = nullptr; // per point: >= sqrt(sum x^2), rounded up (the screen's error bound, screen.hip), built on first use
    float* xf = nullptr;   // f32 copy of x for the screen (with the same slack)
    char* rec = nullptr;   // record layout of the exact entries (k_build_records): x | ir of a point side by side
    int rec_R = 0;
    bool rec_owned = true; // false: the caller's buffer (spkm_shard_create_rec_dev)
    bool rec_tried = false; // one attempt per shard (no retry every call when memory is short)
    // screen bookkeeping of THIS data set (see spkm_assign_accumulate_dev)
    // the screen call's counters for the host policy, written by the call's last kernel (k_call_tail) straight into pinned,
    // device-mapped host memory: 16 counters, then the call's sequence number (system-scope release).  The host looks at
    // them one call later, and only if the number is the one it is waiting for -- no copy, no event, no wait on the hot
    // path (the copy and its event cost a settled iteration 10 of its 230 us).  SPKM_REPORT_WORDS counters (update.hip).
    unsigned* h_nlist = nullptr;
    unsigned* h_nlist_dev = nullptr; // the same memory as the device addresses it
    unsigned nlist_seq = 0;          // number of the report the host is waiting for (nlist_pending)
    bool nlist_pending = false;
    spkm_policy pol;             // which form the next fused call takes (policy.h), fed by the counters read back one call late
    // hinted two-phase screen: the hints (written by k_bounds_steps from the carried bounds)
    float* hintu = nullptr;   // per-point hints of the two-phase screen (k_bounds_steps), npad floats
    long long hintu_len = 0;
    // bounds carried between screen calls (screen.hip, k_center_drift): ub | lb | assignment | drift table, the
    // centroids of the call that produced them, and whether they describe this shard's previous call
    float* hb = nullptr;
    double* hb_centers = nullptr;
    size_t hb_centers_len = 0;
    long long hb_npad = 0;
    int hb_K = 0;
    double hb_gamma = 0.0;
    bool hb_valid = false;
    // unchanged-cluster shortcut of the exact pass (screen.hip, k_cluster_need): per-cluster cache of the LOCAL sums and
    // counts (2 p K doubles), obj2 / max distance / its index (3 K), flags need | touched | same | ibeg | icnt (5 K ints)
    // block summaries of the carried bounds (screen.hip, k_bounds_steps): per 1024 points the clusters present (K <= 128
    // bits), the smallest slack between the bounds, a valid flag -- one allocation of 24 B per block
    char* sp = nullptr;
    long long sp_blocks = 0;
    bool sp_clean = false;            // the last call that wrote bounds maintained the summaries
    const void* sp_assign = nullptr;  // the caller's assignment buffer of that call (a skipped block's part of it is not touched)
    bool assign_synced = false;       // every point of sp_assign holds the library's copy: the previous fused call on this buffer wrote or
                                      // repaired all of it and nobody -- set_lazy_stats, reset_policy, another entry point -- has ended the claim since
    double* hb_cum = nullptr;  // [2]: drift accumulated since the lower bounds were stored (screen.hip, k_bounds_steps), by call parity
    int cum_par = 0;
    double* cl_cache = nullptr;
    int* cl_flags = nullptr;
    size_t cl_pk = 0;
    int cl_K = 0;
    bool cl_valid = false;     // the cache describes this shard's previous screen call completely
    // lazy statistics + incremental sums (spkm_shard_set_lazy_stats): the caller does not need obj2 / the largest distance
    // from every fused call, so a call may leave the exact pass out and move the per-cluster sums by the points that
    // changed cluster only (events: run_screen, k_accumulate_events)
    bool csc_released = false;   // x / ir are gone (spkm_shard_release_csc): the record layout is the only copy of the entries
    bool lazy = false;
    bool cl_stats_valid = false; // cl_cache's obj2 / max / argmax describe the previous call (false after an incremental call)
    int* ev_pt = nullptr;        // events of the current call: point | key (K + old cluster, or new cluster); 2 n each
    int* ev_k = nullptr;
    // the library's own order of the points (regroup_shard): point i of the screen copy / of every per-point array the
    // library keeps is the caller's point map[i] (null: the caller's order).  The records stay in the caller's order.
    int* map = nullptr;
    bool regroup_wanted = false;   // the last call over all points found most 16-point steps mixing clusters
    bool regroup_done = false;     // ... and it has been acted on since the last spkm_shard_reset_policy
    bool pend_full = false;        // the call whose counters are pending screened every point in plain order (its step statistics count)
    int* ev_o = nullptr;         // pair events (K <= 128): the mover's old cluster (-1: none); n + 4096 of them
    size_t ev_o_cap = 0;
    size_t ev_cap = 0;
};

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            if (ctx) snprintf(ctx->errmsg, sizeof(ctx->errmsg), "%s: %s", #expr, hipGetErrorString(_e)); \
            return (int)_e;                                                                             \
        }                                                                                               \
    } while (0)


// (api.hip)
hipError_t allow_lds(spkm_ctx* ctx, const void* kern, size_t bytes);
int ensure(spkm_ctx* ctx, devbuf& b, size_t bytes);
void release(devbuf& b);

// (api_lloyd.hip) spkm_finalize_dev, optionally reporting [dff^2, obj^2, cluster sizes] into ctx->h_res under ctx->res_seq
int spkm_finalize_impl(spkm_ctx* ctx, uint64_t p, uint64_t K, const double* d_reduce, double gamma, double* d_centers,
                       double* d_out, bool to_host);
