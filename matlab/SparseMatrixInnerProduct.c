/* mex gateway: [ip, nx2] = SparseMatrixInnerProduct(X, c) -- drop-in for the reference's
 * private/SparseMatrixInnerProduct.c.  NOT COMPILED HERE (needs MATLAB's mex.h). */
#include "mex.h"
#include "spkm.h"
#include "spkm_mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs != 2) mexErrMsgIdAndTxt("MATLAB:mexFile:invalidNumInputs", "Two input arguments required.");
    if (nlhs > 2) mexErrMsgIdAndTxt("MATLAB:mexFile:maxlhs", "Too many output arguments, needs 1 or 2 outputs.");
    if (!mxIsSparse(prhs[0])) mexErrMsgTxt("Requires first input to be a sparse matrix");
    const mwSize p = mxGetM(prhs[0]), n = mxGetN(prhs[0]);
    if (mxGetNumberOfElements(prhs[1]) < p) mexErrMsgTxt("Center vector must have at least p entries");
    plhs[0] = mxCreateDoubleMatrix(1, n, mxREAL);
    mxArray *nx2 = mxCreateDoubleMatrix(1, n, mxREAL);
    int st = spkm_SparseMatrixInnerProduct_host(spkm_mex_ctx(), p, n, (const uint64_t *)mxGetJc(prhs[0]),
                                                (const uint64_t *)mxGetIr(prhs[0]), mxGetPr(prhs[0]),
                                                mxGetPr(prhs[1]), mxGetPr(plhs[0]), mxGetPr(nx2));
    if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st));
    if (nlhs > 1) plhs[1] = nx2; else mxDestroyArray(nx2);
}
