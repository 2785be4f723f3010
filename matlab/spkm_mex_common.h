/* One spkm context per MATLAB process (mex functions run on MATLAB's interpreter thread). */
#ifndef SPKM_MEX_COMMON_H
#define SPKM_MEX_COMMON_H
#include "mex.h"
#include "spkm.h"
static spkm_ctx *g_spkm_ctx = NULL;
static void spkm_mex_atexit(void) { if (g_spkm_ctx) { spkm_ctx_destroy(g_spkm_ctx); g_spkm_ctx = NULL; } }
static spkm_ctx *spkm_mex_ctx(void)
{
    if (!g_spkm_ctx) {
        int st = spkm_ctx_create(0, NULL, &g_spkm_ctx);
        if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st));
        mexAtExit(spkm_mex_atexit);
    }
    return g_spkm_ctx;
}
#endif
