/*
 * mex gateway for the fused engine the reference does not have (include/spkm.h Part 2 / Part 3): the WHOLE pipeline of
 * kmeans_sparsified.m:316-486 on the device -- mix + sparsify, seeding, Lloyd iterations -- with per-point vectors
 * crossing PCIe once per run, not once per iteration.
 *
 *   spkm_lloyd('sparsify', X, p2, d, s, seed)   X dense p x n (double): chunk by chunk to the GPU, X*(1+2*eps), zero-pad
 *                                               to p2, D = diag(d), FWHT / sqrt(p2), s sampled rows per column scaled by
 *                                               p2 / s (kmeans_sparsified.m:292-334 + randsample_fixedNumberEntries.m:
 *                                               30-64 as ONE fused device pass, spkm_mix_sample_rec_dev: written in the
 *                                               library's record layout, so the entries exist once); the sparse
 *                                               p2 x n result stays resident and never exists on the host
 *   spkm_lloyd('upload', X)                     X sparse p x n, sparsified elsewhere: copied to the GPU once
 *   Y = spkm_lloyd('columns', idx)              sparse p x numel(idx): columns idx (1-based) of the resident data
 *                                               ('sample' start, X(:,iMax) of EmptyAction='singleton')
 *   [C, idx] = spkm_lloyd('kpp', K, gamma)      k-means++ seeding on the device (Arthur_initialization.m:24-69): running
 *                                               minimum of the distances to the newest centre, draws proportional to
 *                                               dist.^2 with MATLAB's own rand / randi as the random source, the
 *                                               400-retry duplicate rule; gamma = [] for the uncorrected distances
 *   [centers, dff, obj, nk] = spkm_lloyd('iterate', centers, gamma [, unbiased [, lazy]])
 *                                               one Lloyd iteration (kmeans_sparsified.m:420-471); centers full: the
 *                                               fused dense-centre call; centers sparse: the sparse-centres branch of
 *                                               findClusterAssignments.m:63-75 through the library's separate steps.
 *                                               assignment, per-cluster sums, ML-corrected centre update, dff and obj --
 *                                               ONE library call (spkm_lloyd_iter), 4 small outputs.  lazy ~= 0: obj may
 *                                               come back NaN (spkm_shard_set_lazy_stats); 'distances' delivers it
 *   a = spkm_lloyd('assignments')               1 x n, 1-based: the latest iteration's (fetched once, after the loop)
 *   [d, obj, iMax] = spkm_lloyd('distances', centersUsed, gamma [, unbiased])
 *                                               1 x n distances of the latest assignment under the centres it was
 *                                               computed WITH, their objective and the first index of the largest
 *   spkm_lloyd('reset')                         a new replicate / new start on the same data (spkm_shard_reset_policy)
 *   spkm_lloyd('release')                       frees the resident data and every device buffer
 *
 * The resident data is whatever the last 'sparsify' / 'upload' produced -- it is NOT looked up by the address of a
 * MATLAB array.  Device buffers are allocated when the shape (n, p, K) changes, not per call.
 *
 * NOT COMPILED IN THIS REPOSITORY (needs MATLAB's mex.h; the HIP runtime API header is used for the copies).
 * Build: see INTEGRATION.md section 2.  matlab/kmeans_sparsified.m is the host that drives it.
 */
#include <math.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "mex.h"
#include "spkm.h"
#include "spkm_mex_common.h"

static spkm_shard *g_shard = NULL;
static size_t g_n = 0, g_p = 0, g_s = 0;       /* g_s > 0: every column has exactly g_s entries ('sparsify') */
/* device arrays behind a shard made by 'sparsify' (adopted by the library; released once the record layout exists) */
static int64_t *g_jc = NULL;
static void *g_ir = NULL;
static double *g_x = NULL;
static void *g_rec = NULL;                      /* ... or its records (columns of <= 64 entries: the entries exist once) */
/* per-point outputs and per-(p, K) buffers stay allocated between calls */
static double *g_dmind = NULL, *g_dC = NULL, *g_dred = NULL, *g_dout = NULL, *g_dstats = NULL;
static int32_t *g_dassign = NULL;
static size_t g_buf_n = 0, g_buf_pk = 0, g_buf_rl = 0;
static int g_exit_registered = 0, g_csc_released = 0;
static int g_last_sparse = 0;   /* the latest 'iterate' ran the sparse-centres assignment (its distances are in g_dmind) */

static void *dmalloc(size_t bytes)
{
    void *q = NULL;
    if (hipMalloc(&q, bytes ? bytes : 8) != hipSuccess) mexErrMsgTxt("hipMalloc failed");
    return q;
}
static void dfree(void **q) { if (*q) { hipFree(*q); *q = NULL; } }
static void check(int st) { if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st)); }

static void release_all(void)
{
    spkm_shard_destroy(g_shard);
    g_shard = NULL;
    g_n = g_p = g_s = 0;
    g_csc_released = 0;
    g_last_sparse = 0;
    dfree((void **)&g_jc); dfree(&g_ir); dfree((void **)&g_x); dfree(&g_rec);
    dfree((void **)&g_dmind); dfree((void **)&g_dassign); dfree((void **)&g_dC); dfree((void **)&g_dred);
    dfree((void **)&g_dout); dfree((void **)&g_dstats);
    g_buf_n = g_buf_pk = g_buf_rl = 0;
}

static void per_point_buffers(size_t n)
{
    if (g_buf_n == n) return;
    dfree((void **)&g_dmind); dfree((void **)&g_dassign);
    g_dmind = (double *)dmalloc((n + 1) * 8);
    g_dassign = (int32_t *)dmalloc((n + 1) * 4);
    g_buf_n = n;
}

static void centre_buffers(size_t p, size_t K)
{
    const size_t pk = p * K, rl = (size_t)spkm_reduce_len(p, K);
    if (g_buf_pk != pk || g_buf_rl != rl) {
        dfree((void **)&g_dC); dfree((void **)&g_dred);
        g_dC = (double *)dmalloc(pk * 8);
        g_dred = (double *)dmalloc(rl * 8);
        g_buf_pk = pk;
        g_buf_rl = rl;
    }
    if (!g_dout) g_dout = (double *)dmalloc(16);
    if (!g_dstats) g_dstats = (double *)dmalloc(32);
}

/* columns idx (1-based, host) of the resident data as a MATLAB sparse p x cnt matrix: spkm_shard_get_column_host reads a
 * column from whichever layout holds it (CSC arrays, or the records once the arrays are released) */
static mxArray *fetch_columns(const double *idx1, size_t cnt)
{
    spkm_ctx *ctx = spkm_mex_ctx();
    size_t cap = g_s ? g_s : 64, nzmax = (g_s ? g_s : 64) * cnt + 1;
    uint64_t *irh = (uint64_t *)mxMalloc(cap * 8);
    double *xh = (double *)mxMalloc(cap * 8);
    mxArray *Y = mxCreateSparse(g_p, cnt, nzmax, mxREAL);
    size_t nz = 0;
    for (size_t c = 0; c < cnt; c++) {
        if (idx1[c] < 1.0 || (size_t)idx1[c] > g_n) mexErrMsgTxt("spkm_lloyd('columns'): index out of range");
        uint64_t have = 0;
        int st = spkm_shard_get_column_host(ctx, g_shard, (uint64_t)idx1[c] - 1, cap, irh, xh, &have);
        if (st == SPKM_ERR_BAD_VALUE && have > cap) {           /* a ragged shard's long column: make room, ask again */
            cap = have;
            irh = (uint64_t *)mxRealloc(irh, cap * 8);
            xh = (double *)mxRealloc(xh, cap * 8);
            st = spkm_shard_get_column_host(ctx, g_shard, (uint64_t)idx1[c] - 1, cap, irh, xh, &have);
        }
        check(st);
        if (nz + have > nzmax) {
            nzmax = 2 * (nz + have);
            mxSetNzmax(Y, nzmax);
            mxSetPr(Y, (double *)mxRealloc(mxGetPr(Y), nzmax * 8));
            mxSetIr(Y, (mwIndex *)mxRealloc(mxGetIr(Y), nzmax * sizeof(mwIndex)));
        }
        mxGetJc(Y)[c] = nz;
        for (size_t j = 0; j < have; j++)                        /* sparse() drops exact zeros (randsample_fixedNumberEntries.m:62) */
            if (xh[j] != 0.0) { mxGetPr(Y)[nz] = xh[j]; mxGetIr(Y)[nz] = (mwIndex)irh[j]; nz++; }
    }
    mxGetJc(Y)[cnt] = nz;
    mxFree(irh);
    mxFree(xh);
    return Y;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    char cmd[16] = {0};
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgTxt("first argument must be a command string");
    spkm_ctx *ctx = spkm_mex_ctx();
    if (!g_exit_registered) { mexAtExit(release_all); g_exit_registered = 1; }
    if (!strcmp(cmd, "release")) { release_all(); return; }
    if (!strcmp(cmd, "reset")) {
        if (g_shard) check(spkm_shard_reset_policy(g_shard));
        return;
    }
    if (!strcmp(cmd, "upload")) {
        if (nrhs != 2) mexErrMsgTxt("usage: spkm_lloyd('upload', X)");
        const mxArray *X = prhs[1];
        if (!mxIsSparse(X) || mxIsComplex(X) || !mxIsDouble(X)) mexErrMsgTxt("Requires first input to be a sparse matrix");
        release_all();
        check(spkm_shard_create_host(ctx, mxGetM(X), mxGetN(X), (const uint64_t *)mxGetJc(X), (const uint64_t *)mxGetIr(X),
                                     mxGetPr(X), &g_shard));
        g_p = mxGetM(X);
        g_n = mxGetN(X);
        return;
    }
    if (!strcmp(cmd, "sparsify")) {
        /* kmeans_sparsified.m:292 (X*(1+2*eps)), :241-245 (zero-pad), :286-289 (D), :248 (hadamard / sqrt(p2)),
         * :326-334 + randsample_fixedNumberEntries.m:30-64 (s rows per column, scaled by p2 / s): spkm_mix_sample_dev */
        if (nrhs != 6) mexErrMsgTxt("usage: spkm_lloyd('sparsify', X, p2, d, s, seed)");
        const mxArray *X = prhs[1];
        if (mxIsSparse(X) || mxIsComplex(X) || !mxIsDouble(X)) mexErrMsgTxt("'sparsify' takes a full real double matrix");
        const size_t p = mxGetM(X), n = mxGetN(X), p2 = (size_t)mxGetScalar(prhs[2]), s = (size_t)mxGetScalar(prhs[4]);
        const uint64_t seed = (uint64_t)mxGetScalar(prhs[5]);
        if (mxGetNumberOfElements(prhs[3]) != p2 || p2 > 65536 || s == 0 || s > p2) mexErrMsgTxt("'sparsify': bad p2 / d / s");
        release_all();
        double *d_sign = (double *)dmalloc(p2 * 8);
        hipMemcpy(d_sign, mxGetPr(prhs[3]), p2 * 8, hipMemcpyHostToDevice);
        /* the dense data crosses PCIe in chunks of at most 256 MB; only 10 B per kept entry stay on the device */
        const size_t chunk = (256u << 20) / (p * 8) ? (256u << 20) / (p * 8) : 1;
        double *d_chunk = (double *)dmalloc((chunk < n ? chunk : n) * p * 8);
        const double *xh = mxGetPr(X);
        const double premul = 1.0 + 2.0 * 2.220446049250313e-16, postdiv = sqrt((double)p2);
        if (s <= 64) {
            /* columns of at most 64 entries: the sparsifier writes the library's RECORD layout (a point's s values, then
             * its s row ids, in spkm_record_bytes(s, 16) bytes) and the shard adopts it -- the entries exist once on the
             * device, from the start (spkm_mix_sample_rec_dev, spkm_shard_create_rec_dev); an entry point that needs CSC
             * arrays (the K = 1 stream of 'kpp', sparse centres) re-materialises library-owned ones, released again below */
            const size_t R = (size_t)spkm_record_bytes(s, 16);
            g_rec = dmalloc(n * R + 256);
            for (size_t c0 = 0; c0 < n; c0 += chunk) {
                const size_t m = n - c0 < chunk ? n - c0 : chunk;
                hipMemcpy(d_chunk, xh + c0 * p, m * p * 8, hipMemcpyHostToDevice);
                check(spkm_mix_sample_rec_dev(ctx, p, p2, m, d_chunk, d_sign, premul, postdiv, s, seed, c0, 16, (char *)g_rec + c0 * R));
            }
            check(spkm_ctx_sync(ctx));
            check(spkm_shard_create_rec_dev(ctx, p2, n, s, 16, g_rec, &g_shard));
        } else {
            g_ir = dmalloc((n * s + 48) * 2);                    /* 48 entries of slack: the fixed-stride kernels */
            g_x = (double *)dmalloc((n * s + 48) * 8);
            hipMemset(g_ir, 0, (n * s + 48) * 2);
            hipMemset(g_x, 0, (n * s + 48) * 8);
            for (size_t c0 = 0; c0 < n; c0 += chunk) {
                const size_t m = n - c0 < chunk ? n - c0 : chunk;
                hipMemcpy(d_chunk, xh + c0 * p, m * p * 8, hipMemcpyHostToDevice);
                check(spkm_mix_sample_dev(ctx, p, p2, m, d_chunk, d_sign, premul, postdiv, s, seed, c0,
                                          (unsigned short *)g_ir + c0 * s, 16, g_x + c0 * s));
            }
            check(spkm_ctx_sync(ctx));
            int64_t *jch = (int64_t *)mxMalloc((n + 1) * 8);
            for (size_t i = 0; i <= n; i++) jch[i] = (int64_t)(i * s);
            g_jc = (int64_t *)dmalloc((n + 1) * 8);
            hipMemcpy(g_jc, jch, (n + 1) * 8, hipMemcpyHostToDevice);
            mxFree(jch);
            check(spkm_shard_create_dev(ctx, p2, n, n * s, g_jc, g_ir, 16, g_x, n * s + 48, &g_shard));
        }
        hipFree(d_chunk);
        hipFree(d_sign);
        g_p = p2; g_n = n; g_s = s;
        return;
    }
    if (!g_shard) mexErrMsgTxt("spkm_lloyd: no data resident; call spkm_lloyd('sparsify', ...) or spkm_lloyd('upload', X) first");
    const size_t p = g_p, n = g_n;
    if (!strcmp(cmd, "columns")) {
        if (nrhs != 2) mexErrMsgTxt("usage: Y = spkm_lloyd('columns', idx)");
        plhs[0] = fetch_columns(mxGetPr(prhs[1]), mxGetNumberOfElements(prhs[1]));
        return;
    }
    if (!strcmp(cmd, "kpp")) {
        /* Arthur_initialization.m:24-69 with the running minimum (one K = 1 distance evaluation per round) */
        if (nrhs != 3) mexErrMsgTxt("usage: [C, idx] = spkm_lloyd('kpp', K, gamma)");
        const size_t K = (size_t)mxGetScalar(prhs[1]);
        const double gamma = mxIsEmpty(prhs[2]) ? 0.0 : mxGetScalar(prhs[2]);
        per_point_buffers(n);
        double *d_run = (double *)dmalloc(n * 8), *d_cum = (double *)dmalloc(n * 8), *d_c = (double *)dmalloc(p * 8);
        double *chosen = (double *)mxMalloc(K * 8), *col = (double *)mxCalloc(p, 8);
        mxArray *rnd, *arg = mxCreateDoubleScalar((double)n);
        mexCallMATLAB(1, &rnd, 1, &arg, "randi");                /* randi(n,1) (:35) */
        chosen[0] = mxGetScalar(rnd);
        mxDestroyArray(rnd);
        for (size_t k = 1; k < K; k++) {
            /* the newest centre, densified: a column of the sparse data */
            mxArray *Yc = fetch_columns(&chosen[k - 1], 1);
            memset(col, 0, p * 8);
            for (mwIndex j = mxGetJc(Yc)[0]; j < mxGetJc(Yc)[1]; j++) col[mxGetIr(Yc)[j]] = mxGetPr(Yc)[j];
            mxDestroyArray(Yc);
            hipMemcpy(d_c, col, p * 8, hipMemcpyHostToDevice);
            check(spkm_assign_dev(ctx, g_shard, 1, d_c, gamma, g_dassign, g_dmind, NULL, NULL));
            double total = 0.0;
            check(spkm_kpp_update_dev(ctx, n, g_dmind, d_run, k == 1, d_cum, &total));
            size_t tries = 0;
            double pick = 0.0;
            for (;;) {                                           /* :50-65: redraw while the point is already a centre */
                if (total > 0.0) {
                    mexCallMATLAB(1, &rnd, 0, NULL, "rand");
                    int64_t idx = 0;
                    check(spkm_kpp_draw_dev(ctx, n, d_cum, mxGetScalar(rnd) * total, &idx));
                    mxDestroyArray(rnd);
                    pick = (double)idx + 1.0;
                } else {
                    mexCallMATLAB(1, &rnd, 1, &arg, "randi");
                    pick = mxGetScalar(rnd);
                    mxDestroyArray(rnd);
                }
                int dup = 0;
                for (size_t q = 0; q < k; q++) dup |= chosen[q] == pick;
                if (!dup) break;
                if (++tries >= 400) mexErrMsgTxt("Cannot sample with replacement with this distribution");
            }
            chosen[k] = pick;
        }
        plhs[0] = fetch_columns(chosen, K);                      /* centres = X(:, chosen), sparse (:36,68) */
        if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(1, K, mxREAL); memcpy(mxGetPr(plhs[1]), chosen, K * 8); }
        mxDestroyArray(arg);
        mxFree(chosen); mxFree(col);
        hipFree(d_run); hipFree(d_cum); hipFree(d_c);
        return;
    }
    if (!strcmp(cmd, "assignments")) {
        if (g_buf_n != n) mexErrMsgTxt("spkm_lloyd('assignments'): no iteration has run");
        plhs[0] = mxCreateDoubleMatrix(1, n, mxREAL);           /* 1-based, as MATLAB's min returns them */
        int32_t *ha = (int32_t *)mxMalloc((n + 1) * 4);
        hipMemcpy(ha, g_dassign, n * 4, hipMemcpyDeviceToHost);
        double *a = mxGetPr(plhs[0]);
        for (size_t i = 0; i < n; i++) a[i] = (double)ha[i] + 1.0;
        mxFree(ha);
        return;
    }
    if (!strcmp(cmd, "distances")) {
        /* `distances` of the latest assignment under the centres it was computed with (kmeans_sparsified.m:420), once
         * per run -- and with them obj (:471) and [~,iMax] = max(distances) (:436) */
        if (nrhs < 3 || nrhs > 4) mexErrMsgTxt("usage: [d, obj, iMax] = spkm_lloyd('distances', centersUsed, gamma [, unbiased])");
        const mxArray *C = prhs[1];
        if (mxIsComplex(C) || !mxIsDouble(C)) mexErrMsgTxt("centers must be a real double matrix");
        if (mxGetM(C) != p) mexErrMsgTxt(spkm_strerror(SPKM_ERR_CENTER_ROWS));
        const size_t K = mxGetN(C);
        const int unbiased = nrhs == 4 ? (mxGetScalar(prhs[3]) != 0.0) : 1;
        if (mxIsSparse(C)) {
            /* SPARSE centres: the iteration they were used in went through spkm_assign_sparse_centers_dev
             * (private/findClusterAssignments.m:63-75), which wrote every distance and the statistics already -- they are
             * still in g_dmind / g_dstats.  (The dense formula below would be the wrong one for them, and mxGetPr of a
             * sparse matrix holds nnz values, not p*K.) */
            if (!g_last_sparse)
                mexErrMsgTxt("spkm_lloyd('distances'): sparse centres, but the latest iteration did not use sparse centres");
        } else {
            centre_buffers(p, K);
            double *d_cu = (double *)dmalloc(p * K * 8);
            hipMemcpy(d_cu, mxGetPr(C), p * K * 8, hipMemcpyHostToDevice);
            check(spkm_distances_stats_dev(ctx, g_shard, K, d_cu, unbiased ? mxGetScalar(prhs[2]) : 0.0, g_dassign, g_dmind, g_dstats));
            check(spkm_ctx_sync(ctx));
            hipFree(d_cu);
        }
        plhs[0] = mxCreateDoubleMatrix(1, n, mxREAL);
        hipMemcpy(mxGetPr(plhs[0]), g_dmind, n * 8, hipMemcpyDeviceToHost);
        double st[3];
        hipMemcpy(st, g_dstats, 24, hipMemcpyDeviceToHost);
        if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(sqrt(st[0]));
        if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(st[2] + 1.0);
        return;
    }
    if (strcmp(cmd, "iterate") || nrhs < 3 || nrhs > 5)
        mexErrMsgTxt("usage: [C, dff, obj, nk] = spkm_lloyd('iterate', centers, gamma [, unbiased [, lazy]])");
    const mxArray *C = prhs[1];
    if (mxIsComplex(C) || !mxIsDouble(C)) mexErrMsgTxt("centers must be a real double matrix");
    const size_t K = mxGetN(C);
    if (mxGetM(C) != p) mexErrMsgTxt(spkm_strerror(SPKM_ERR_CENTER_ROWS));
    const double gamma = mxGetScalar(prhs[2]);
    const int unbiased = nrhs >= 4 ? (mxGetScalar(prhs[3]) != 0.0) : 1;
    const int lazy = nrhs >= 5 ? (mxGetScalar(prhs[4]) != 0.0) : 0;
    const size_t pk = p * K;
    int have_hres = 0;             /* the dense-centre call brought dff^2 / obj^2 / nk to the host itself */
    double hres_out[2] = {0.0, 0.0};
    per_point_buffers(n);
    centre_buffers(p, K);
    check(spkm_shard_set_lazy_stats(g_shard, lazy));
    if (mxIsSparse(C)) {
        /* SPARSE centres -- the first iteration(s) after a 'sample' / k-means++ start with the default denseCenters =
         * false: the distance runs over supp(x) n supp(c) with the separate scalings 1/gamma_c and 1/gamma
         * (private/findClusterAssignments.m:63-75).  Values + support mask go up; assignment, accumulation, exchange and
         * the centre update are the library's separate steps (the fused call is the dense-centre path) */
        double *hv = (double *)mxCalloc(pk, 8);
        unsigned char *hm = (unsigned char *)mxCalloc(pk, 1), *d_mask = (unsigned char *)dmalloc(pk);
        const mwIndex *cj = mxGetJc(C), *ci = mxGetIr(C);
        for (size_t k = 0; k < K; k++)
            for (mwIndex j = cj[k]; j < cj[k + 1]; j++) { hv[k * p + ci[j]] = mxGetPr(C)[j]; hm[k * p + ci[j]] = 1; }
        hipMemcpy(g_dC, hv, pk * 8, hipMemcpyHostToDevice);
        hipMemcpy(d_mask, hm, pk, hipMemcpyHostToDevice);
        mxFree(hv); mxFree(hm);
        check(spkm_assign_sparse_centers_dev(ctx, g_shard, K, g_dC, d_mask, unbiased ? gamma : 0.0, g_dassign, g_dmind, g_dstats, NULL));
        g_last_sparse = 1;   /* 'distances' with these (sparse) centres returns what this call wrote */
        check(spkm_accumulate_dev(ctx, g_shard, K, g_dassign, g_dred));
        check(spkm_allreduce_f64_dev(ctx, g_dred, spkm_reduce_len(p, K)));
        check(spkm_finalize_dev(ctx, p, K, g_dred, gamma, g_dC, g_dout));
        check(spkm_ctx_sync(ctx));
        hipFree(d_mask);
        goto outputs;
    }
    g_last_sparse = 0;
    if (hipMemcpy(g_dC, mxGetPr(C), pk * 8, hipMemcpyHostToDevice) != hipSuccess) mexErrMsgTxt("copy of the centres failed");
    /* assignment + accumulation (+ all-reduce when a communicator is attached) + centre update: the certified f32
     * screen with exact f64 confirmation where the data qualifies (every column the same length, as
     * randsample_fixedNumberEntries produces), the exact kernels otherwise -- same outputs either way.  d_mind = NULL:
     * the n distances are not written per iteration ('distances' delivers them once) */
    {   /* (spkm_lloyd_iter_host: dff^2, obj^2 and the cluster sizes are in host memory when the call returns -- through
         *  pinned memory the device maps; no copies of them below) */
        double *hres = (double *)mxMalloc((2 + K) * sizeof(double));
        check(spkm_lloyd_iter_host(ctx, g_shard, K, g_dC, gamma, unbiased, g_dassign, NULL, NULL, NULL, g_dred, g_dout, hres));
        have_hres = 1;
        hres_out[0] = hres[0]; hres_out[1] = hres[1];
        if (nlhs > 3) { plhs[3] = mxCreateDoubleMatrix(1, K, mxREAL); memcpy(mxGetPr(plhs[3]), hres + 2, K * sizeof(double)); }
        mxFree(hres);
    }
    if (g_s && !g_csc_released) {
        /* the first fused call has built the library's own layouts: the gateway's value / row arrays can go
         * (spkm_shard_release_csc; 'columns' keeps working: the library reads the records) */
        if (spkm_shard_release_csc(ctx, g_shard) == SPKM_OK) { dfree(&g_ir); dfree((void **)&g_x); }
        g_csc_released = 1;
    }
outputs:
    /* per iteration: p*K + K + 2 doubles back to the host, nothing of size n */
    plhs[0] = mxCreateDoubleMatrix(p, K, mxREAL);
    hipMemcpy(mxGetPr(plhs[0]), g_dC, pk * 8, hipMemcpyDeviceToHost);
    double out[2];
    if (have_hres) { out[0] = hres_out[0]; out[1] = hres_out[1]; }
    else hipMemcpy(out, g_dout, 16, hipMemcpyDeviceToHost);
    if (nlhs > 1) plhs[1] = mxCreateDoubleScalar(sqrt(out[0]));   /* norm(centersOld-centers,'fro') */
    if (nlhs > 2) plhs[2] = mxCreateDoubleScalar(sqrt(out[1]));   /* sqrt(sum(distances.^2)); NaN: not evaluated (lazy) */
    if (nlhs > 3 && !have_hres) { plhs[3] = mxCreateDoubleMatrix(1, K, mxREAL); hipMemcpy(mxGetPr(plhs[3]), g_dred + 2 * pk, K * 8, hipMemcpyDeviceToHost); }
}
