/*
 * mex gateway for the fused engine the reference does not have (include/spkm.h Part 2 / Part 3).
 *
 *   spkm_lloyd('upload', X)                     X sparse p x n: copied to the GPU once, kept resident
 *   [assignments, distances, centers, dff, obj, nk] = spkm_lloyd('iterate', centers, gamma [, unbiased])
 *                                               one Lloyd iteration with dense centres on the resident data
 *                                               (kmeans_sparsified.m:420-471): assignment, per-cluster sums, the
 *                                               ML-corrected centre update, dff and obj -- one library call
 *                                               (spkm_lloyd_iter); empty clusters keep their column and show nk == 0
 *   spkm_lloyd('reset')                         a new replicate / new start on the same data
 *                                               (spkm_shard_reset_policy)
 *   spkm_lloyd('release')                       frees the resident data and every device buffer
 *
 * The resident data is whatever the last 'upload' passed -- it is NOT looked up by the address of a MATLAB array
 * (MATLAB may hand a different matrix the same address after a free).  Device buffers are allocated when the
 * shape (n, p, K) changes, not per call.
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
static size_t g_n = 0, g_p = 0;
/* per-point outputs and per-(p, K) buffers stay allocated between calls */
static double *g_dmind = NULL, *g_dC = NULL, *g_dred = NULL, *g_dout = NULL;
static int32_t *g_dassign = NULL;
static size_t g_buf_n = 0, g_buf_pk = 0, g_buf_rl = 0;
static int g_exit_registered = 0;

static void *dmalloc(size_t bytes)
{
    void *q = NULL;
    if (hipMalloc(&q, bytes ? bytes : 8) != hipSuccess) mexErrMsgTxt("hipMalloc failed");
    return q;
}
static void dfree(void **q) { if (*q) { hipFree(*q); *q = NULL; } }

static void release_all(void)
{
    spkm_shard_destroy(g_shard);
    g_shard = NULL;
    g_n = g_p = 0;
    dfree((void **)&g_dmind); dfree((void **)&g_dassign); dfree((void **)&g_dC); dfree((void **)&g_dred); dfree((void **)&g_dout);
    g_buf_n = g_buf_pk = g_buf_rl = 0;
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    char cmd[16] = {0};
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgTxt("first argument must be a command string");
    spkm_ctx *ctx = spkm_mex_ctx();
    if (!g_exit_registered) { mexAtExit(release_all); g_exit_registered = 1; }
    int st;
    if (!strcmp(cmd, "release")) { release_all(); return; }
    if (!strcmp(cmd, "reset")) {
        if (g_shard && (st = spkm_shard_reset_policy(g_shard)) != SPKM_OK) mexErrMsgTxt(spkm_strerror(st));
        return;
    }
    if (!strcmp(cmd, "upload")) {
        if (nrhs != 2) mexErrMsgTxt("usage: spkm_lloyd('upload', X)");
        const mxArray *X = prhs[1];
        if (!mxIsSparse(X) || mxIsComplex(X) || !mxIsDouble(X)) mexErrMsgTxt("Requires first input to be a sparse matrix");
        spkm_shard_destroy(g_shard);
        g_shard = NULL;
        st = spkm_shard_create_host(ctx, mxGetM(X), mxGetN(X), (const uint64_t *)mxGetJc(X), (const uint64_t *)mxGetIr(X),
                                    mxGetPr(X), &g_shard);
        if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st));
        g_p = mxGetM(X);
        g_n = mxGetN(X);
        return;
    }
    if (strcmp(cmd, "iterate") || nrhs < 3 || nrhs > 4)
        mexErrMsgTxt("usage: [a, d, C, dff, obj, nk] = spkm_lloyd('iterate', centers, gamma [, unbiased])");
    if (!g_shard) mexErrMsgTxt("spkm_lloyd: no data resident; call spkm_lloyd('upload', X) first");
    const mxArray *C = prhs[1];
    if (mxIsSparse(C) || mxIsComplex(C) || !mxIsDouble(C)) mexErrMsgTxt("centers must be a full real double matrix");
    const size_t p = g_p, n = g_n, K = mxGetN(C);
    if (mxGetM(C) != p) mexErrMsgTxt(spkm_strerror(SPKM_ERR_CENTER_ROWS));
    const double gamma = mxGetScalar(prhs[2]);
    const int unbiased = nrhs == 4 ? (mxGetScalar(prhs[3]) != 0.0) : 1;
    const size_t pk = p * K, rl = (size_t)spkm_reduce_len(p, K);
    if (g_buf_n != n) {
        dfree((void **)&g_dmind); dfree((void **)&g_dassign);
        g_dmind = (double *)dmalloc((n + 1) * 8);
        g_dassign = (int32_t *)dmalloc((n + 1) * 4);
        g_buf_n = n;
    }
    if (g_buf_pk != pk || g_buf_rl != rl) {
        dfree((void **)&g_dC); dfree((void **)&g_dred);
        g_dC = (double *)dmalloc(pk * 8);
        g_dred = (double *)dmalloc(rl * 8);
        g_buf_pk = pk;
        g_buf_rl = rl;
    }
    if (!g_dout) g_dout = (double *)dmalloc(16);
    if (hipMemcpy(g_dC, mxGetPr(C), pk * 8, hipMemcpyHostToDevice) != hipSuccess) mexErrMsgTxt("copy of the centres failed");
    /* assignment + accumulation (+ all-reduce when a communicator is attached) + centre update: the certified f32
     * screen with exact f64 confirmation where the data qualifies (every column the same length, as
     * randsample_fixedNumberEntries produces), the exact kernels otherwise -- same outputs either way */
    st = spkm_lloyd_iter(ctx, g_shard, K, g_dC, gamma, unbiased, g_dassign, g_dmind, NULL, NULL, g_dred, g_dout);
    if (st == SPKM_OK) st = spkm_ctx_sync(ctx);
    if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st));
    plhs[0] = mxCreateDoubleMatrix(1, n, mxREAL);               /* 1-based, as MATLAB's min returns them */
    {
        int32_t *ha = (int32_t *)mxMalloc((n + 1) * 4);
        hipMemcpy(ha, g_dassign, n * 4, hipMemcpyDeviceToHost);
        double *a = mxGetPr(plhs[0]);
        for (size_t i = 0; i < n; i++) a[i] = (double)ha[i] + 1.0;
        mxFree(ha);
    }
    if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(1, n, mxREAL); hipMemcpy(mxGetPr(plhs[1]), g_dmind, n * 8, hipMemcpyDeviceToHost); }
    if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(p, K, mxREAL); hipMemcpy(mxGetPr(plhs[2]), g_dC, pk * 8, hipMemcpyDeviceToHost); }
    double out[2];
    hipMemcpy(out, g_dout, 16, hipMemcpyDeviceToHost);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(sqrt(out[0]));   /* norm(centersOld-centers,'fro') */
    if (nlhs > 4) plhs[4] = mxCreateDoubleScalar(sqrt(out[1]));   /* sqrt(sum(distances.^2))        */
    if (nlhs > 5) { plhs[5] = mxCreateDoubleMatrix(1, K, mxREAL); hipMemcpy(mxGetPr(plhs[5]), g_dred + 2 * pk, K * 8, hipMemcpyDeviceToHost); }
}
