/*
 * mex gateway: y = hadamard_pthreads(x)  -- drop-in for the reference's private/hadamard_pthreads.c
 * (gateway at :227-264; built by setup_kmeans.m:55-57 with -DNTHREADS=maxNumCompThreads()).
 *
 * The reference splits the columns of x over NTHREADS workers (:121-204), each running the same butterfly as
 * hadamard.c; a column's result does not depend on which worker took it.  Here every column goes through the one HIP
 * kernel (k_fwht_lds, csrc/fwht.hip) behind spkm_hadamard_pthreads_host, whose output is bit-identical to the
 * reference's worker for any NTHREADS (tests/test_gpu_ops.py against oracle/_ref's build of :57-119).  -DNTHREADS is
 * accepted and ignored, so setup_kmeans.m's mex line keeps working.
 * NOT COMPILED HERE (needs MATLAB's mex.h).
 *     mex -largeArrayDims -I<repo>/include hadamard_pthreads.c -L<repo>/sparsifiedkmeans_amd -lspkm -lamdhip64
 */
#include "mex.h"
#include "spkm.h"
#include "spkm_mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    /* argument checks in the reference's order and with its texts (hadamard_pthreads.c:233-252; the size checks of
     * checkPowerTwo, :207-223, come back as status codes from the library) */
    if (nrhs != 1)
        mexErrMsgTxt("One and only one input required; must be a column vector or matrix, with # rows a power of 2.");
    if (nlhs > 1) mexErrMsgTxt("Too many output arguments.");
    const mwSize m = mxGetM(prhs[0]), n = mxGetN(prhs[0]);
    if (m <= 1) mexErrMsgTxt(spkm_strerror(SPKM_ERR_LEN_LE_1));
    if (m & (m - 1)) mexErrMsgTxt(spkm_strerror(SPKM_ERR_NOT_POW2));
    if (mxIsComplex(prhs[0])) mexErrMsgTxt("Input must be real.");
    else if (mxIsSparse(prhs[0])) mexErrMsgTxt("Input must be a full matrix, not sparse.");
    else if (!mxIsDouble(prhs[0])) mexErrMsgTxt("Input must be of type double.");
    plhs[0] = mxCreateDoubleMatrix(m, n, mxREAL);
    int st = spkm_hadamard_pthreads_host(spkm_mex_ctx(), m, n, mxGetPr(prhs[0]), mxGetPr(plhs[0]));
    if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st));
}
