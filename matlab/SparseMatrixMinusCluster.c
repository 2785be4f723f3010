/*
 * mex gateway: dist = SparseMatrixMinusCluster(X, C [, beta])  -- drop-in for the reference's
 * private/SparseMatrixMinusCluster.c (same MATLAB name, arguments, output and error texts),
 * forwarding to libspkm.so (include/spkm.h).  NOT COMPILED IN THIS REPO'S CI: it needs MATLAB's
 * mex.h.  Build where MATLAB exists:
 *     mex -largeArrayDims -I<repo>/include SparseMatrixMinusCluster.c -L<repo>/sparsifiedkmeans_amd -lspkm -lamdhip64
 */
#include "mex.h"
#include "spkm.h"
#include "spkm_mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs != 2 && nrhs != 3)
        mexErrMsgIdAndTxt("MATLAB:mexFile:invalidNumInputs", "Two input arguments required.");
    if (nlhs > 1) mexErrMsgIdAndTxt("MATLAB:mexFile:maxlhs", "Too many output arguments.");
    if (!mxIsSparse(prhs[0])) mexErrMsgTxt("Requires first input to be a sparse matrix");
    const mwSize p = mxGetM(prhs[0]), n = mxGetN(prhs[0]), K = mxGetN(prhs[1]);
    double beta = 0.0;
    if (nrhs == 3) beta = mxGetScalar(prhs[2]);
    plhs[0] = mxCreateDoubleMatrix(K, n, mxREAL);
    /* mwIndex is a 64-bit unsigned integer under -largeArrayDims: passed through unchanged */
    int st = spkm_SparseMatrixMinusCluster_host(spkm_mex_ctx(), p, n, (const uint64_t *)mxGetJc(prhs[0]),
                                                (const uint64_t *)mxGetIr(prhs[0]), mxGetPr(prhs[0]),
                                                mxGetM(prhs[1]), K, mxGetPr(prhs[1]), nrhs == 3 ? &beta : NULL,
                                                mxGetPr(plhs[0]));
    if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st)); /* same texts as the reference's mexErrMsgTxt calls */
}
