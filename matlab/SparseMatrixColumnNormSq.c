/* mex gateway: nx2 = SparseMatrixColumnNormSq(X) -- drop-in for the reference's
 * private/SparseMatrixColumnNormSq.c.  NOT COMPILED HERE (needs MATLAB's mex.h). */
#include "mex.h"
#include "spkm.h"
#include "spkm_mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs != 1) mexErrMsgIdAndTxt("MATLAB:mexFile:invalidNumInputs", "One input arguments required.");
    if (nlhs != 1) mexErrMsgIdAndTxt("MATLAB:mexFile:maxlhs", "Too many output arguments, needs 1 output.");
    if (!mxIsSparse(prhs[0])) mexErrMsgTxt("Requires first input to be a sparse matrix");
    const mwSize n = mxGetN(prhs[0]);
    plhs[0] = mxCreateDoubleMatrix(1, n, mxREAL);
    int st = spkm_SparseMatrixColumnNormSq_host(spkm_mex_ctx(), n, (const uint64_t *)mxGetJc(prhs[0]),
                                                mxGetPr(prhs[0]), mxGetPr(plhs[0]));
    if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st));
}
