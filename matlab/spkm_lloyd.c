/*
 * mex gateway for the fused engine the reference does not have:
 *   [assignments, distances, centers, dff, obj, nk] = spkm_lloyd('iterate', X, centers, gamma)
 * One call = one Lloyd iteration with dense centres (kmeans_sparsified.m:420-471): assignment, per-cluster
 * accumulation and the ML-corrected centre update run on the GPU; X is uploaded on the first call and kept
 * resident (spkm_lloyd('release') frees it).  NOT COMPILED HERE (needs MATLAB's mex.h and the HIP runtime
 * headers for the two small copies).  INTEGRATION.md shows the six-line patch to kmeans_sparsified.m.
 */
#include <math.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "mex.h"
#include "spkm.h"
#include "spkm_mex_common.h"

static spkm_shard *g_shard = NULL;
static const mxArray *g_shard_key = NULL; /* identity of the uploaded X (MATLAB shares data pointers) */
/* per-point outputs stay allocated between calls (no reallocation of 1.2 GB per iteration at N = 1e8) */
static double *g_dmind = NULL;
static int32_t *g_dassign = NULL;
static size_t g_npts = 0;

static void *dmalloc(size_t bytes) { void *p = NULL; if (hipMalloc(&p, bytes) != hipSuccess) mexErrMsgTxt("hipMalloc failed"); return p; }

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    char cmd[16] = {0};
    if (nrhs < 1 || mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgTxt("first argument must be a command string");
    spkm_ctx *ctx = spkm_mex_ctx();
    if (!strcmp(cmd, "release")) {
        spkm_shard_destroy(g_shard); g_shard = NULL; g_shard_key = NULL;
        hipFree(g_dmind); hipFree(g_dassign); g_dmind = NULL; g_dassign = NULL; g_npts = 0;
        return;
    }
    if (strcmp(cmd, "iterate") || nrhs != 4) mexErrMsgTxt("usage: spkm_lloyd('iterate', X, centers, gamma)");
    const mxArray *X = prhs[1], *C = prhs[2];
    if (!mxIsSparse(X)) mexErrMsgTxt("Requires first input to be a sparse matrix");
    const mwSize p = mxGetM(X), n = mxGetN(X), K = mxGetN(C);
    if (mxGetM(C) != p) mexErrMsgTxt(spkm_strerror(SPKM_ERR_CENTER_ROWS));
    const double gamma = mxGetScalar(prhs[3]);
    int st;
    if (!g_shard || g_shard_key != (const mxArray *)mxGetPr(X)) {   /* new data: upload once */
        spkm_shard_destroy(g_shard);
        st = spkm_shard_create_host(ctx, p, n, (const uint64_t *)mxGetJc(X), (const uint64_t *)mxGetIr(X), mxGetPr(X), &g_shard);
        if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st));
        g_shard_key = (const mxArray *)mxGetPr(X);
    }
    const size_t pk = (size_t)p * K, rl = (size_t)spkm_reduce_len(p, K);
    if (g_npts != (size_t)n) {
        hipFree(g_dmind); hipFree(g_dassign);
        g_dmind = (double *)dmalloc(((size_t)n + 1) * 8);
        g_dassign = (int32_t *)dmalloc(((size_t)n + 1) * 4);
        g_npts = (size_t)n;
    }
    double *dC = (double *)dmalloc(pk * 8), *dred = (double *)dmalloc(rl * 8), *dmind = g_dmind;
    double *dout = (double *)dmalloc(16);
    int32_t *dassign = g_dassign;
    hipMemcpy(dC, mxGetPr(C), pk * 8, hipMemcpyHostToDevice);
    /* assignment + accumulation in one call: the certified f32 screen with exact f64 confirmation where the
     * shard qualifies (every column the same length, as randsample_fixedNumberEntries produces), the exact
     * kernels otherwise -- same outputs either way */
    st = spkm_assign_accumulate_dev(ctx, g_shard, K, dC, gamma, dassign, dmind, NULL, NULL, dred);
    if (st == SPKM_OK) st = spkm_finalize_dev(ctx, p, K, dred, gamma, dC, dout);
    if (st == SPKM_OK) st = spkm_ctx_sync(ctx);
    if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st));
    plhs[0] = mxCreateDoubleMatrix(1, n, mxREAL);               /* 1-based, as MATLAB's min returns them */
    {
        int32_t *ha = (int32_t *)mxMalloc((n + 1) * 4);
        hipMemcpy(ha, dassign, n * 4, hipMemcpyDeviceToHost);
        double *a = mxGetPr(plhs[0]);
        for (mwSize i = 0; i < n; i++) a[i] = (double)ha[i] + 1.0;
        mxFree(ha);
    }
    if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(1, n, mxREAL); hipMemcpy(mxGetPr(plhs[1]), dmind, n * 8, hipMemcpyDeviceToHost); }
    if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(p, K, mxREAL); hipMemcpy(mxGetPr(plhs[2]), dC, pk * 8, hipMemcpyDeviceToHost); }
    double out[2];
    hipMemcpy(out, dout, 16, hipMemcpyDeviceToHost);
    if (nlhs > 3) plhs[3] = mxCreateDoubleScalar(sqrt(out[0]));   /* norm(centersOld-centers,'fro') */
    if (nlhs > 4) plhs[4] = mxCreateDoubleScalar(sqrt(out[1]));   /* sqrt(sum(distances.^2))        */
    if (nlhs > 5) { plhs[5] = mxCreateDoubleMatrix(1, K, mxREAL); hipMemcpy(mxGetPr(plhs[5]), dred + 2 * pk, K * 8, hipMemcpyDeviceToHost); }
    hipFree(dC); hipFree(dred); hipFree(dout);
}
