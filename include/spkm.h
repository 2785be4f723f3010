/*
 * spkm.h -- C ABI of the MI355X-native sparsified-K-means Lloyd engine (libspkm.so).
 *
 * This is the drop-in boundary for the reference's native layer
 * (stephenbeckr/SparsifiedKMeans v2.1): MATLAB mex files exporting
 *     void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
 * (private/SparseMatrixMinusCluster.c:44-45, private/SparseMatrixInnerProduct.c:40-41,
 *  private/SparseMatrixColumnNormSq.c:35-36, private/hadamard.c:115-116,
 *  private/hadamard_pthreads.c:227-228).  A mex gateway unpacks its mxArrays
 * (mxGetM/N/Pr/Ir/Jc) and calls the matching spkm_* host-buffer entry point below;
 * INTEGRATION.md shows those ~30-line gateways.  Part 2 is the device-resident engine the
 * reference does not have (fused assign / accumulate / finalise on a shard kept in HBM).
 *
 * Conventions
 *  - plain C types only; all matrices column-major (MATLAB layout); sparse matrices are CSC
 *    with 64-bit jc/ir exactly as mxGetJc/mxGetIr return them under -largeArrayDims.
 *  - every function returns an int status: 0 = ok, < 0 = argument error (mirrors the
 *    reference's mexErrMsgTxt conditions; text via spkm_strerror), > 0 = hipError_t.
 *    Nothing throws or long-jumps across the boundary; outputs are caller-owned.
 *  - "_host" entry points take host pointers and do their own transfers; "_dev" entry points
 *    take device pointers (e.g. torch.Tensor.data_ptr()) and enqueue on the context's stream.
 *  - cluster indices are 0-based int32 on this side of the ABI (MATLAB value minus one).
 */
#ifndef SPKM_H
#define SPKM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPKM_OK 0
#define SPKM_ERR_NULL_ARG (-1)       /* required pointer is NULL                                          */
#define SPKM_ERR_CENTER_ROWS (-2)    /* "Center vector must be ... pxk" SparseMatrixMinusCluster.c:104-107 */
#define SPKM_ERR_BETA_K (-3)         /* beta form needs K == 1          SparseMatrixMinusCluster.c:119-120 */
#define SPKM_ERR_LEN_LE_1 (-4)       /* "Vector length must be greater than 1."       hadamard.c:100-102   */
#define SPKM_ERR_NOT_POW2 (-5)       /* "Vector length must be power of 2."           hadamard.c:108-110   */
#define SPKM_ERR_BAD_CSC (-6)        /* jc not monotone / ir out of range / rows not ascending (not checked
                                        by the reference; rejected here because the kernels index with them) */
#define SPKM_ERR_UNSUPPORTED (-7)    /* shape outside what this build supports (see message)               */
#define SPKM_ERR_NO_DEVICE (-8)      /* no HIP device / gfx950 code object could not be loaded             */
#define SPKM_ERR_BAD_VALUE (-9)      /* scalar argument out of range                                       */

typedef struct spkm_ctx spkm_ctx;     /* device id + stream + scratch; one per host thread               */
typedef struct spkm_shard spkm_shard; /* a CSC block of points resident in HBM                            */

const char *spkm_strerror(int status);
int spkm_version(void); /* 10000*major + 100*minor + patch */

/* stream: a hipStream_t (as void*) to enqueue on, or NULL for the default stream.  The
 * context never creates threads.  Not thread-safe: use one context per host thread. */
int spkm_ctx_create(int device, void *stream, spkm_ctx **out);
void spkm_ctx_destroy(spkm_ctx *ctx);
int spkm_ctx_sync(spkm_ctx *ctx); /* block until everything enqueued on the context's stream is done */
/* The library's A/B switches (environment variables SPKM_NO_SCREEN, SPKM_NO_BOUNDS, ...: DESIGN.md section 6.1; none
 * changes an output) are read ONCE, by spkm_ctx_create.  This re-reads them: for tests and A/B tools that toggle a
 * switch inside one process. */
int spkm_ctx_reload_switches(spkm_ctx *ctx);
/* device facts used by the host driver / bench: [0]=CU count, [1]=LDS bytes per workgroup,
 * [2]=device memory bytes, [3]=wavefront size */
int spkm_device_info(spkm_ctx *ctx, int64_t info[4]);

/* ------------------------------------------------------------------------------------------
 * Part 1 -- mex-equivalent operators, HOST buffers (one call == one mexFunction call)
 * ------------------------------------------------------------------------------------------ */

/* dist = SparseMatrixMinusCluster(X, C)        (private/SparseMatrixMinusCluster.c:1-8,104-182)
 * dist = SparseMatrixMinusCluster(X, c, beta)  (:9-11,118-129; K must be 1)
 * X: p x n sparse (jc[n+1], ir[nnz], x[nnz]); C: c_rows x K dense; dist: K x n dense out.
 * c_rows != p -> SPKM_ERR_CENTER_ROWS.  beta == NULL selects the plain form. */
int spkm_SparseMatrixMinusCluster_host(spkm_ctx *ctx, uint64_t p, uint64_t n, const uint64_t *jc,
                                       const uint64_t *ir, const double *x, uint64_t c_rows, uint64_t K,
                                       const double *C, const double *beta, double *dist);

/* [ip, nx2] = SparseMatrixInnerProduct(X, c)    (private/SparseMatrixInnerProduct.c:1-9,87-100)
 * c must have at least p entries (the reference's size check at :71-77 compares with n; the
 * kernel indexes c by row).  ip, nx2: n doubles each (nx2 may be NULL). */
int spkm_SparseMatrixInnerProduct_host(spkm_ctx *ctx, uint64_t p, uint64_t n, const uint64_t *jc,
                                       const uint64_t *ir, const double *x, const double *c, double *ip,
                                       double *nx2);

/* nx2 = SparseMatrixColumnNormSq(X)             (private/SparseMatrixColumnNormSq.c:1-9,71-77) */
int spkm_SparseMatrixColumnNormSq_host(spkm_ctx *ctx, uint64_t n, const uint64_t *jc, const double *x,
                                       double *nx2);

/* y = hadamard(x) / y = hadamard_pthreads(x)    (private/hadamard.c:57-152,
 * private/hadamard_pthreads.c:69-264).  x, y: m x n dense; m must be a power of two > 1.
 * Both reference functions give bit-identical output; both names map to one kernel. */
int spkm_hadamard_host(spkm_ctx *ctx, uint64_t m, uint64_t n, const double *x, double *y);
int spkm_hadamard_pthreads_host(spkm_ctx *ctx, uint64_t m, uint64_t n, const double *x, double *y);

/* ------------------------------------------------------------------------------------------
 * Part 2 -- device-resident Lloyd engine (replaces the MATLAB hot loops of
 * kmeans_sparsified.m:417-486 and private/findClusterAssignments.m:76-82,168-171)
 * ------------------------------------------------------------------------------------------ */

/* Upload a CSC block of n points (columns) of dimension p; indices are validated and narrowed
 * (row ids to 16 bit when p <= 65536).  Rows must ascend within a column (MATLAB invariant).
 * A shard holds at most 0x7ff00000 (~2.1e9) points: SPKM_ERR_UNSUPPORTED beyond (shard the data). */
int spkm_shard_create_host(spkm_ctx *ctx, uint64_t p, uint64_t n, const uint64_t *jc, const uint64_t *ir,
                           const double *x, spkm_shard **out);
/* Adopt device arrays without copying (caller keeps them alive): d_jc int64[n+1],
 * d_ir uint32 (ir_bits=32) or uint16 (ir_bits=16), d_x double; both hold nnz entries inside
 * allocations of `capacity` >= nnz entries.  With capacity >= nnz + 16 and a fixed number of
 * entries per column the fastest exact kernel variant is used (it reads, and ignores, up to 15
 * entries past a column's end); capacity >= nnz + 48 also unlocks the certified f32 screen for K > 16.
 * Rows must ascend within a column; not re-validated here. */
int spkm_shard_create_dev(spkm_ctx *ctx, uint64_t p, uint64_t n, uint64_t nnz, const int64_t *d_jc,
                          const void *d_ir, int ir_bits, const double *d_x, uint64_t capacity,
                          spkm_shard **out);
/* Forget what earlier calls learned about how well the screen certifies on this shard (the choice between the
 * exact kernels, the plain screen and the two-phase screen adapts from call to call).  Hosts call it when the
 * centres are about to jump -- a new replicate, a new start -- so that the first call does not run in a mode
 * tuned for the previous, converged centres.  Outputs never depend on it. */
int spkm_shard_reset_policy(spkm_shard *s);
/* Lazy statistics (on != 0).  kmeans_sparsified.m:471 evaluates obj = sqrt(sum(distances.^2)) in every iteration but
 * uses it only for Display='iter' (:472-475) and, after the loop, for the iteration that turned out to be the last
 * (:489-503); [~,iMax] = max(distances) (:436) only when a cluster is empty.  A host that does not display per iteration
 * says so here, and fused calls (spkm_assign_accumulate_dev / spkm_lloyd_iter with d_mind == NULL) may then leave the
 * exact pass out: the per-cluster sums and counts are moved by the points that CHANGED cluster (each read twice: out of
 * its old cluster's sums, into its new one's) instead of re-accumulated over every member, the library's upper bounds
 * come from the screen's certificate.  Assignments, counts, cluster sizes: as before, bit for bit.  Sums: the members'
 * sums, to rounding (an add and a subtract per move instead of a fresh summation; bar 1e-6 relative).  In such a call
 *     d_reduce[2pK + K] (obj2) and d_stats[0..2] are NaN -- not evaluated --
 * and the host obtains them for the iteration it needs from spkm_distances_stats_dev.  The library decides per call
 * (few points moved in the previous call, its caches describe the previous call, ...).  A lazy call that has to run the
 * full pass -- a run's first call, one in which too many points move -- runs it WITHOUT the distances (sums and counts
 * only; the statistics are NaN there too); with d_mind != NULL a call evaluates everything, as always.
 * SPKM_NO_INCREMENTAL=1 switches the incremental calls off, SPKM_NO_SUMS_ONLY=1 the distance-free full pass.
 * While lazy statistics are on, d_assign is the library's to keep between calls: a host that passes the SAME buffer
 * again must not have written to it -- blocks of 1024 points whose carried bounds settle them as a whole are then not
 * visited at all, their part of d_assign included (SPKM_NO_BLOCK_SKIP=1: every point is looked at, as without lazy
 * statistics).  A different buffer is noticed and filled completely. */
int spkm_shard_set_lazy_stats(spkm_shard *s, int on);
/* Halve the resident footprint of a fixed-stride shard (every column has the same number of entries, at most 64): build
 * now what the fused call would build on its first use -- the record layout (a point's values and row ids side by side)
 * and the screen's f32 copy + norms -- and let go of the CSC value / row-id arrays.  A shard made by
 * spkm_shard_create_host frees them; for an adopted one (spkm_shard_create_dev) the library merely stops referencing
 * d_ir / d_x and the caller may free them.  jc stays.  From then on the fused call, spkm_distances_*_dev and the point
 * lists read the records; an entry point that needs CSC (spkm_assign_dev, spkm_accumulate_dev, the sparse-centres
 * assignment, SPKM_NO_SCREEN=1) re-materialises library-owned arrays from the records first (one streaming pass) --
 * results never change.  N = 1e8, s = 51: 146 GB -> 93 GB resident.  SPKM_ERR_UNSUPPORTED: ragged shard, columns longer
 * than 64, or no room for the records beside the arrays -- nothing was released. */
int spkm_shard_release_csc(spkm_ctx *ctx, spkm_shard *s);
/* The stored entries of column `col` (0-based) back on the host -- row ids ascending in ir_out, values in x_out, *count of
 * them -- whichever layout holds them now (CSC arrays, or the records after spkm_shard_release_csc).  cap = room in
 * ir_out / x_out; *count > cap: SPKM_ERR_BAD_VALUE and *count says how much is needed.  Blocks on the stream. */
int spkm_shard_get_column_host(spkm_ctx *ctx, const spkm_shard *s, uint64_t col, uint64_t cap, uint64_t *ir_out,
                               double *x_out, uint64_t *count);
void spkm_shard_destroy(spkm_shard *s);
int spkm_shard_info(const spkm_shard *s, uint64_t *p, uint64_t *n, uint64_t *nnz, int *ir_bits);
/* Data in arbitrary order.  A 16-point step of the screen is skipped on the carried bounds, or finished early by the hinted
 * form, only if all its 16 points allow it; with the points of a cluster scattered over the shard that is rare.  When a
 * fused call over all points of a LAZY shard (spkm_shard_set_lazy_stats) finds fewer than one in eight of its steps holding one
 * cluster, the next call first REGROUPS the shard: the library's own order of the points -- the order of its screen copy,
 * bounds, hints and block summaries -- becomes "by cluster, the points that are sure of it first" (one gather of the
 * records, one write of the 306-B-per-point screen copy: once per spkm_shard_reset_policy at most).  Nothing the caller sees
 * moves: d_assign, d_mind, spkm_shard_get_column_host and the records keep the caller's order (the library reaches them
 * through an index map), and no result depends on the order (kmeans_sparsified.m:430-431 sums over find(assignments == k)).
 * SPKM_NO_REGROUP=1: A/B switch.  info[0] = 1 if the shard's own order differs from the caller's, info[1] = 1 if it was
 * regrouped since the last spkm_shard_reset_policy. */
int spkm_shard_order_info(const spkm_shard *s, int64_t info[2]);

/* Length (in doubles) of the per-iteration reduce buffer for (p, K):
 *   [ sums p*K | counts p*K | nk K | obj2 1 ]      -- one SUM all-reduce covers all of it. */
uint64_t spkm_reduce_len(uint64_t p, uint64_t K);

/* [assignments, distances] = findClusterAssignments(X, centers, [], gamma), dense-centre branch
 * (private/findClusterAssignments.m:76-82,168-171): d_centers is p x K on the device; gamma <= 0
 * means "gamma empty" (no centers/gamma scaling).  Outputs (device): d_assign int32[n] (0-based),
 * d_mind double[n].  d_stats double[3] (device, may be NULL): { sum(mind^2), max(mind),
 * first index of the max } -- the latter two feed EmptyAction='singleton'
 * (kmeans_sparsified.m:436).  d_nk_u64 (device, K uint64, may be NULL) receives the cluster sizes. */
int spkm_assign_dev(spkm_ctx *ctx, const spkm_shard *s, uint64_t K, const double *d_centers, double gamma,
                    int32_t *d_assign, double *d_mind, double *d_stats, uint64_t *d_nk_u64);

/* The same with SPARSE centres (private/findClusterAssignments.m:63-75; Lloyd iteration 1 under the
 * default denseCenters=false, and k-means++ style starts): d_centers is p x K dense with zeros where a
 * centre has no entry, d_mask (p x K bytes) marks each centre's support.  For centre k the distance
 * runs over supp(x_i) ∩ supp(c_k) with x divided by gamma_c(k) = nnz(c_k)/p and c by gamma
 * (gamma <= 0: no scaling at all, :73).  Outputs as spkm_assign_dev. */
int spkm_assign_sparse_centers_dev(spkm_ctx *ctx, const spkm_shard *s, uint64_t K, const double *d_centers,
                                   const uint8_t *d_mask, double gamma, int32_t *d_assign, double *d_mind,
                                   double *d_stats, uint64_t *d_nk_u64);

/* Per-cluster accumulation (kmeans_sparsified.m:430-431,447-448 with the mask of :352-355):
 * fills d_reduce (device, spkm_reduce_len doubles) = { S = sum X(:,ind), Cnt = sum spones(X)(:,ind),
 * nk, obj2 } for this shard.  Call after spkm_assign_dev on the same context (obj2/nk are taken
 * from that call).  The buffer is overwritten, not accumulated. */
int spkm_accumulate_dev(spkm_ctx *ctx, const spkm_shard *s, uint64_t K, const int32_t *d_assign,
                        double *d_reduce);

/* spkm_assign_dev + spkm_accumulate_dev in one call -- the front half of a Lloyd iteration, up to the
 * all-reduce.  Same outputs, bit for bit.  For K > 16 on a fixed-stride shard the library may run its
 * fast path: a certified f32 screen over all centroids, exact reference arithmetic for every point the
 * screen cannot certify, and the exact distance of every point to its assigned centroid fused into the
 * accumulation pass (csrc/screen.hip).  Nothing computed in f32 reaches an output.  The environment
 * variable SPKM_NO_SCREEN=1 forces the all-exact kernels.
 * Work-saving state: all of it lives in the library's own buffers, per shard; nothing depends on what the caller
 * does with its output buffers between calls, and none of it changes an output.
 * Hints: from the second screen call on a shard, the screen compares the competition's partial sums with a
 * per-point estimate of the distance to the previous centroid (previous exact distance and that centroid's
 * movement): a group of 16 points stops after a quarter of its entries once the partial squared distance of every
 * other centroid of a tile already exceeds 1.5x the hinted distance squared (after half of them in the first three
 * hinted calls following spkm_shard_reset_policy, when the hints are still loose; SPKM_NO_LATE_SPLIT=1: always a
 * quarter).  A misleading hint costs time (and pauses the hints for a few calls), never correctness.  SPKM_NO_HINT=1
 * disables them.
 * Carried bounds: the library keeps, per shard, its own copy of the previous screen call's assignment, an
 * upper bound of every point's distance to its centroid and a lower bound of its distance to all others, plus
 * that call's centroids.  On the next call the centroids' movement (triangle inequality on the masked distances)
 * proves for most points of a converging run that their centroid is unchanged; 16-point steps of such points
 * skip the screen.  The exact pass still recomputes every point's distance to its centroid and all sums, so the
 * outputs are the same bit for bit.  Nothing here depends on buffers the caller owns.  SPKM_NO_BOUNDS=1 disables
 * the skipping; spkm_shard_reset_policy forgets the bounds. */
int spkm_assign_accumulate_dev(spkm_ctx *ctx, const spkm_shard *s, uint64_t K, const double *d_centers,
                               double gamma, int32_t *d_assign, double *d_mind, double *d_stats,
                               uint64_t *d_nk_u64, double *d_reduce);
/* d_mind == NULL in spkm_assign_accumulate_dev / spkm_lloyd_iter: the caller does not need the per-point distances of
 * THIS call.  Everything is computed as before -- every distance in reference arithmetic, obj2, the largest distance and
 * its index for EmptyAction='singleton', the library's bounds -- only the n doubles are not written (on this part a
 * gigabyte of stores costs as much as eight of loads: 1.1 ms per iteration at N = 1e8; and the unchanged-cluster
 * shortcut below applies only then).  A driver wants the
 * distances of its LAST iteration only (kmeans_sparsified.m:493-503,514-518 use `distances` after the loop); it gets them
 * from spkm_distances_dev:
 *   d_mind[i] = distance of point i to centroid d_assign[i] under d_centers / gamma -- the value findClusterAssignments
 *   returned for that point (private/findClusterAssignments.m:78,169), squared terms added in storage order.
 * d_centers must be the centres the assignment was computed WITH (the ones passed to the call that returned d_assign,
 * before its update).  Right after a fused call on the same shard this is one more streaming pass; otherwise a generic
 * kernel.  Blocks on the stream once. */
int spkm_distances_dev(spkm_ctx *ctx, const spkm_shard *s, uint64_t K, const double *d_centers, double gamma,
                       const int32_t *d_assign, double *d_mind);
/* The same, and d_stats double[3] (device, may be NULL) = { obj2 = sum_i d_mind[i]^2, max_i d_mind[i], its first index }:
 * what a fused call hands over in d_stats -- for hosts that asked for lazy statistics (spkm_shard_set_lazy_stats). */
int spkm_distances_stats_dev(spkm_ctx *ctx, const spkm_shard *s, uint64_t K, const double *d_centers, double gamma,
                             const int32_t *d_assign, double *d_mind, double *d_stats);
/* info[0] = path taken by the last spkm_assign_accumulate_dev (0 = exact tiles, 1 = screen + exact
 * confirmation), info[1] = points the screen could not certify.  Blocks on the stream. */
int spkm_last_path_info(spkm_ctx *ctx, int64_t info[2]);
/* After a screen call on the 4-lanes-per-point kernel: info[1] = rounds (of 4 stored entries) per column,
 * info[0] = rounds evaluated for ALL centroids.  info[0] < info[1]: the two-phase screen was used -- partial
 * sums (lower bounds) for every centroid, the remaining entries only for each tile's leader; the library
 * switches it on by itself when the previous call found almost no point with a runner-up within 2.25x of the
 * winner, and off again when it certifies poorly.  SPKM_NO_PRUNE=1 disables it; outputs never change.
 * Both 0 after any other path. */
int spkm_last_screen_rounds(spkm_ctx *ctx, int64_t info[2]);
/* info[0] = form of the last screen call: 0 plain, 1 two-phase, 2 hinted two-phase (-1: no screen);
 * info[1..4] = that call's counters: points listed for exact evaluation, points with a runner-up within 2.25x
 * (a bound: over-counts in the two-phase forms), (16-point step, centroid tile) pairs finished early by the
 * hinted form, 16-point steps skipped altogether on the bounds carried from the previous call (see
 * spkm_assign_accumulate_dev); info[5] = running total of skipped steps over all calls on this context;
 * info[6] = how the call got its per-cluster sums: 0 = the full accumulation pass with every point's distance, 3 = the
 * full pass WITHOUT distances (lazy statistics, spkm_shard_set_lazy_stats: a run's first call, or too many movers for
 * events; SPKM_NO_SUMS_ONLY=1: A/B switch), 2 = incrementally, by the points that changed cluster (events) -- for a call
 * that queued both forms and let the device choose (a run's second lazy call: no mover count is back yet; SPKM_NO_DUAL=1:
 * A/B switch) the value says which one the device opened; 4 = incrementally, the events applied one by one without
 * sorting them by cluster first (the previous call counted fewer than 2048 movers: one launch instead of three;
 * SPKM_NO_DIRECT_EVENTS=1: A/B switch);
 * info[7] = 2 if the
 * bounds were applied point by point (the list then names points; info[4] still counts the steps whose 16 points all
 * passed): the library switches to that form when the previous call's test passed >= 60 % of the points and whole
 * steps would leave several times as many points on the screen as failed, so that data in arbitrary order -- where a 16-point
 * step is rarely settled as a whole -- skips as much as cluster-contiguous data does.  (The listed points' entries are
 * then read from the record layout of the exact pass, 512 contiguous bytes per point at s = 51, when the shard has
 * one.)
 * Blocks on the stream. */
int spkm_last_screen_mode(spkm_ctx *ctx, int64_t info[8]);
/* How the last fused call moved the per-cluster sums when it did so incrementally (spkm_last_screen_mode info[6] = 2 or 4):
 * info[0] = 0 not an incremental call, 1 events sorted by cluster and applied through LDS slabs, 2 events applied one by
 * one (few movers); info[1] = 1 if the call recorded PAIR events -- one per mover, (point, new cluster, old cluster),
 * sorted by (new, old) pair so that a mover's entries are read once and go into the new cluster's sums and out of the
 * old one's together (K <= 128; SPKM_NO_PAIR_EVENTS=1: A/B switch) -- 0 if two events per mover, each applied on its
 * own (kmeans_sparsified.m:447-448's S and Cnt either way).  Does not block. */
int spkm_last_events_form(spkm_ctx *ctx, int64_t info[2]);

/* Unchanged-cluster shortcut of the fused call's exact pass.  A cluster (i) whose centroid is BITWISE the one the
 * previous fused call on this shard was given and (ii) that no point left or entered is not streamed again: every
 * member's distance to it, the per-cluster sums and counts (kmeans_sparsified.m:447-448), its share of obj2 and its
 * largest distance are exactly what the previous call produced and are taken from the library's per-shard cache.  What a
 * converging run looks like from the 9th iteration on at N = 1e8, K = 100: 4 clusters still exchange points, 96 clusters
 * with 98 % of the points are settled.  Outputs are the same as without it (bit for bit where they were reproducible
 * before: distances, assignments, counts; sums are atomics in no fixed order either way).  Not used when d_mind is
 * requested (the distances have to be written then) and never across spkm_shard_reset_policy, a change of K or gamma, or
 * any other entry point in between.  SPKM_NO_CLUSTER_SKIP=1 switches it off.
 * info[0] = running total (this context) of the points the exact pass actually streamed, info[1] = in the last call.
 * Blocks on the stream. */
int spkm_exact_pass_points(spkm_ctx *ctx, int64_t info[2]);
/* Work of the fused call's 4-lanes-per-point screen launches, in ROUNDS (4 stored entries of a 16-point step against the
 * centroids of one 32-centroid tile), running totals over this context: info[0] = rounds executed for all centroids of a
 * tile (a step skipped on the carried bounds contributes none, a (step, tile) pair that the two-phase forms finish early
 * only the rounds it ran for all centroids), info[1] = rounds of launches that do ALL the work (every step, every round).
 * Measurement aid: bench.py credits the timed window's launches with info[0] / info[1] of an iteration's algorithmic
 * bytes.  Blocks on the stream. */
int spkm_screen_work_totals(spkm_ctx *ctx, int64_t info[2]);

/* k-means++ seeding on the device (private/Arthur_initialization.m:38-69).  A round evaluates the distances to the NEWEST
 * centre only (spkm_assign_dev with K = 1 -> d_dist_new) and
 *   spkm_kpp_update_dev: d_run = min(d_run, d_dist_new) (first_round != 0: d_run = d_dist_new) -- the reference's
 *     [~,dist] = findClusterAssignments(X, all chosen centres) bit for bit, at 1/k of the work -- and d_cum = inclusive
 *     prefix sums of d_run.^2 in a fixed order; *total (host, may be NULL) = their sum (blocks on the stream);
 *   spkm_kpp_draw_dev: *index (host) = the first i with d_cum[i] > target: randsample(n,1,true,dist.^2) (:50) for
 *     target = u * total with the HOST's uniform random number u (blocks on the stream).
 * All buffers are n doubles on the device. */
int spkm_kpp_update_dev(spkm_ctx *ctx, uint64_t n, const double *d_dist_new, double *d_run, int first_round,
                        double *d_cum, double *total);
int spkm_kpp_draw_dev(spkm_ctx *ctx, uint64_t n, const double *d_cum, double target, int64_t *index);

/* centers(:,k) = gamma*S(:,k) ./ (Cnt(:,k) + 1e-16) for clusters with nk > 0
 * (kmeans_sparsified.m:448); empty clusters keep their column.  d_centers is updated in place;
 * d_out double[2] (device) = { ||old-new||_F^2, obj2 } (kmeans_sparsified.m:470-471 before sqrt). */
int spkm_finalize_dev(spkm_ctx *ctx, uint64_t p, uint64_t K, const double *d_reduce, double gamma,
                      double *d_centers, double *d_out);

/* y = hadamard(x) on device buffers (m x n, column-major). */
int spkm_fwht_dev(spkm_ctx *ctx, uint64_t m, uint64_t n, const double *d_x, double *d_y);
/* mix(X) = hadamard(D * [X*premul; 0]) / postdiv  (kmeans_sparsified.m:241-248,286-295):
 * d_x is p x n, d_y is p2 x n, d_sign is p2 doubles of +-1 (NULL = none), premul = 1+2*eps or 1,
 * postdiv = sqrt(p2) or 0 for none. */
int spkm_mix_dev(spkm_ctx *ctx, uint64_t p, uint64_t p2, uint64_t n, const double *d_x, const double *d_sign,
                 double premul, double postdiv, double *d_y);

/* Device sparsifier (the producer of the hot path's input):
 *   X = randsample_fixedNumberEntries(mix(X), s)    kmeans_sparsified.m:316-334,
 *                                                   private/randsample_fixedNumberEntries.m:30-64
 * d_x: p x n dense chunk (columns = points).  For column c (global index col0 + c): s distinct rows,
 * uniform without replacement (Philox4x32-10 keyed by (seed, col0 + c): independent of chunking and of
 * the number of GPUs), ascending, into d_ir_out[c*s .. c*s+s) (uint16 / uint32 by ir_bits) and the values
 * (mix(x)[row] ) / (s/p2) into d_x_out[c*s ..).  The dense mixed column never reaches HBM.
 * premul / postdiv / d_sign as in spkm_mix_dev.  Needs 16 <= p2 <= 16384. */
int spkm_mix_sample_dev(spkm_ctx *ctx, uint64_t p, uint64_t p2, uint64_t n, const double *d_x,
                        const double *d_sign, double premul, double postdiv, uint64_t s, uint64_t seed,
                        uint64_t col0, void *d_ir_out, int ir_bits, double *d_x_out);

/* The same sparsifier writing RECORDS -- the layout the fused call reads (a point's s values, then its s row ids, in
 * R = spkm_record_bytes(s, ir_bits) bytes, 16-byte aligned: 512 B at s = 51): column c (global index col0 + c) goes to
 * d_rec_out + c * R.  A shard made from such a buffer (spkm_shard_create_rec_dev) never holds the separate CSC arrays,
 * so the entries exist ONCE at every moment (a CSC shard holds them twice -- 166 GB against 100 at N = 1e8 -- from its
 * first fused call until spkm_shard_release_csc). */
uint64_t spkm_record_bytes(uint64_t s, int ir_bits);
int spkm_mix_sample_rec_dev(spkm_ctx *ctx, uint64_t p, uint64_t p2, uint64_t n, const double *d_x,
                            const double *d_sign, double premul, double postdiv, uint64_t s, uint64_t seed,
                            uint64_t col0, int ir_bits, void *d_rec_out);
/* A shard over n records of exactly s entries each that the caller holds on the device (and keeps alive): what
 * kmeans_sparsified.m:316-334 produces for one GPU, in the library's own layout.  Everything a CSC shard can do it can do:
 * an entry point that needs CSC arrays re-materialises library-owned ones from the records first.
 * The allocation behind d_rec must extend at least 256 BYTES past the last record (n * spkm_record_bytes(s, ir_bits) + 256):
 * the record kernels read whole 16-byte pieces and fetch a wave's batch ahead of its bounds check (the library gives its
 * own record buffers the same slack) -- checked against the allocation d_rec lies in (hipMemGetAddressRange):
 * SPKM_ERR_BAD_VALUE when it ends earlier.  1 <= s <= 64. */
int spkm_shard_create_rec_dev(spkm_ctx *ctx, uint64_t p, uint64_t n, uint64_t s, int ir_bits, const void *d_rec,
                              spkm_shard **out);

/* Widening copy in front of the sparsifier for streamed ingest (private/sampleAndMixFromLargeFile.m:100-113 reads a
 * chunk as doubles; a dataset of 1e9 points is stored narrower): d_dst[i] = (double) d_src[i], exact for every kind.
 * kind: 1 float32, 2 uint8, 3 int16, 4 int32.  Both buffers on the device, `count` elements. */
int spkm_widen_f64_dev(spkm_ctx *ctx, int kind, uint64_t count, const void *d_src, double *d_dst);

/* Dense (unsampled) data behind the reference's two-pass outputs.
 * d_X: n x p, point i at d_X + i*p (= column-major p x n); d_centers: K x p, centre k at d_centers + k*p.
 *
 * spkm_dense_assign_dev replaces findClusterAssignments(full(X), centers), dense branch, expanded quadratic
 * (private/findClusterAssignments.m:157-165,168-171; called from kmeans_sparsified.m:558 and
 * private/recalculateAssignmentLargeFile.m:96): d_dist[i] = min_k sqrt(|x_i|^2 - 2 x_i'c_k + |c_k|^2),
 * d_assign[i] = first k attaining it (0-based).  The Gram block runs on the f64 matrix cores.
 *
 * spkm_dense_accumulate_dev replaces the numerators / counts of mean(full(XFull(:,ind)),2)
 * (kmeans_sparsified.m:545-550; recalculateAssignmentLargeFile.m:100-107): d_sums[k*p + r] += sum of
 * X(r, i) over points with d_assign[i] == k, d_counts[k] += their number.  Both outputs ACCUMULATE so that
 * a file can be streamed through in chunks; zero them first.  d_assign values must lie in [0, K). */
int spkm_dense_assign_dev(spkm_ctx *ctx, uint64_t p, uint64_t n, const double *d_X, uint64_t K,
                          const double *d_centers, int32_t *d_assign, double *d_dist);
int spkm_dense_accumulate_dev(spkm_ctx *ctx, uint64_t p, uint64_t n, const double *d_X, uint64_t K,
                              const int32_t *d_assign, double *d_sums, double *d_counts);

/* ------------------------------------------------------------------------------------------
 * Part 3 -- one whole Lloyd iteration, and the data-parallel exchange
 *
 * The reference has no distributed code.  Points shard naturally over the GPUs of a node (one process per GPU);
 * the only exchange of an iteration is ONE SUM all-reduce of the reduce buffer [sums | counts | nk | obj2]
 * (kmeans_sparsified.m:447-448 needs sum X(:,ind) and sum spones(X)(:,ind) over ALL points of a cluster), after
 * which every rank finalises identical centres.  The collective is RCCL (ncclAllReduce, ncclDouble, ncclSum) over
 * xGMI, issued by the library on the context's stream -- no host round trip between the accumulation and the
 * finalisation.  librccl is bound at run time (the copy already loaded in the process -- PyTorch's -- or
 * $SPKM_RCCL_PATH, librccl.so, /opt/rocm/lib/librccl.so); a single-GPU user never loads it.
 * ------------------------------------------------------------------------------------------ */
#define SPKM_COMM_ID_BYTES 128
#define SPKM_ERR_COMM (-10)          /* librccl not loadable, or an RCCL call failed (text via spkm_ctx_last_error) */

/* Rendezvous token (ncclGetUniqueId): ONE rank creates it, the host ships the 128 bytes to the other ranks by
 * whatever it has (torch.distributed broadcast, MPI, a file). */
int spkm_comm_unique_id(uint8_t id[SPKM_COMM_ID_BYTES]);
/* Attach this context (its device, its stream) to the communicator of `nranks` processes as rank `rank`
 * (ncclCommInitRank; collective -- every rank must call it).  nranks == 1 is valid.  A context holds at most one
 * communicator; SPKM_ERR_BAD_VALUE if one is attached already. */
int spkm_comm_init(spkm_ctx *ctx, int nranks, int rank, const uint8_t id[SPKM_COMM_ID_BYTES]);
int spkm_comm_destroy(spkm_ctx *ctx);                       /* no-op without a communicator */
int spkm_comm_info(spkm_ctx *ctx, int *nranks, int *rank);  /* nranks = 0: none attached */
/* In-place SUM all-reduce of `count` doubles over the attached communicator, on the context's stream.  What
 * spkm_lloyd_iter issues; exported for the few other global sums of a run (SUMD, the two-pass means).  Without a
 * communicator: returns SPKM_OK and leaves the buffer as it is (one rank). */
int spkm_allreduce_f64_dev(spkm_ctx *ctx, double *d_buf, uint64_t count);
/* One full Lloyd iteration with dense centres (kmeans_sparsified.m:417-471) in one call, nothing returned to the host:
 *   spkm_assign_accumulate_dev  ->  all-reduce of d_reduce (if a communicator is attached)  ->  spkm_finalize_dev.
 * gamma = SparsityLevel (the factor of :448); unbiased != 0: distances to centers/gamma (unbiasedDistance, the
 * default, :369-370), else to the centres as they are.  d_centers is updated in place, d_out[2] =
 * { ||old-new||_F^2, obj2 } (both global).  Same outputs, bit for bit, as the three calls.  Empty clusters keep their column (d_reduce's nk block tells; EmptyAction is the host's). */
int spkm_lloyd_iter(spkm_ctx *ctx, const spkm_shard *s, uint64_t K, double *d_centers, double gamma, int unbiased,
                    int32_t *d_assign, double *d_mind, double *d_stats, uint64_t *d_nk_u64, double *d_reduce,
                    double *d_out);
/* The same iteration for a host that decides after every one of them, as the reference's driver does (kmeans_sparsified.m:
 * 432 "ind = find(~counts)", 470-487 "dff = norm(...); if dff < Tol, break"): besides everything spkm_lloyd_iter does, the
 * call WAITS until the iteration's results are in host memory and returns them in host_out[2 + K] =
 * { ||old-new||_F^2, obj2, nk[0..K-1] } (cluster sizes as doubles; all global).  No device-to-host copy and no stream
 * synchronisation is involved: the finalisation's last workgroup stores the values into pinned host memory that the device
 * maps, followed by a sequence number, and the call returns when the number has arrived (a stream that runs dry without it --
 * a platform without coherent host mappings -- is noticed after 0.25 s and the values are copied the ordinary way).
 * d_out receives the same two values on the device, as with spkm_lloyd_iter. */
int spkm_lloyd_iter_host(spkm_ctx *ctx, const spkm_shard *s, uint64_t K, double *d_centers, double gamma, int unbiased,
                         int32_t *d_assign, double *d_mind, double *d_stats, uint64_t *d_nk_u64, double *d_reduce,
                         double *d_out, double *host_out);
/* Text of the last HIP / RCCL failure recorded on this context ("" if none). */
const char *spkm_ctx_last_error(spkm_ctx *ctx);

/* Timing hooks for bench.py: hipEvents recorded on the context's stream around the dominant
 * kernel of the last spkm_assign_dev call.  Returns its duration in milliseconds (blocks). */
int spkm_last_assign_kernel_ms(spkm_ctx *ctx, double *ms);
/* Per-launch log of the same kernel: enable=1 starts (and clears) the log, every later
 * spkm_assign_dev records one event pair without any host sync; spkm_timing_read blocks on the
 * stream and returns up to cap durations (ms) and the number recorded.  enable=2: calls of
 * spkm_assign_accumulate_dev that take the screen path record two pairs each, the screen kernel and then the
 * exact accumulation kernel (k_exact_accumulate), alternating in the log. */
int spkm_timing_log(spkm_ctx *ctx, int enable);
/* Developer aid: per-workgroup (start, end) wall-clock stamps (100 MHz) of the tiled assignment
 * kernel.  enable=1 arms it; enable=0 copies up to cap pairs into out and reports the grid size. */
int spkm_debug_block_times(spkm_ctx *ctx, int enable, int64_t *out, int cap, int *nblocks);
int spkm_timing_read(spkm_ctx *ctx, double *ms, int cap, int *count);
/* Developer / test aid: the bounds a shard carries from its last screen call (csrc/screen.hip, k_center_drift), copied to
 * host buffers of n entries each (any may be NULL): ub = upper bound on each point's distance to its centroid, lb = lower
 * bound on its distance to every other centroid (the stored value minus the drift accumulated since), lib_assign = the
 * library's copy of the assignment.  Blocks on the stream.  SPKM_ERR_UNSUPPORTED when the shard holds no valid bounds.
 * tests/test_gpu_screen.py checks that they ARE bounds after every kind of call. */
int spkm_debug_shard_bounds(spkm_ctx *ctx, const spkm_shard *s, float *ub, double *lb, int32_t *lib_assign);

#ifdef __cplusplus
}
#endif
#endif /* SPKM_H */
