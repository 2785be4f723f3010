/*
 * mex gateway: y = hadamard(x)  -- drop-in for the reference's private/hadamard.c (hadamard_pthreads.c next to this
 * file is the gateway for private/hadamard_pthreads.c: both names map to one HIP kernel whose output is bit-identical
 * to either).  NOT COMPILED HERE (needs MATLAB's mex.h).
 *     mex -largeArrayDims -I<repo>/include hadamard.c -L<repo>/sparsifiedkmeans_amd -lspkm -lamdhip64
 */
#include "mex.h"
#include "spkm.h"
#include "spkm_mex_common.h"

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs != 1) mexErrMsgTxt("Exactly one argument required.");          /* hadamard.c:122-124 */
    if (nlhs > 1) mexErrMsgTxt("Too many output arguments.");               /* :125-127 */
    if (mxIsComplex(prhs[0])) mexErrMsgTxt("Input must be real.");          /* :134-136 */
    if (mxIsSparse(prhs[0])) mexErrMsgTxt("Input must be full");            /* :137-140 */
    if (!mxIsDouble(prhs[0])) mexErrMsgTxt("Input must be of type double.");
    const mwSize m = mxGetM(prhs[0]), n = mxGetN(prhs[0]);
    plhs[0] = mxCreateDoubleMatrix(m, n, mxREAL);
    int st = spkm_hadamard_host(spkm_mex_ctx(), m, n, mxGetPr(prhs[0]), mxGetPr(plhs[0]));
    if (st != SPKM_OK) mexErrMsgTxt(spkm_strerror(st)); /* "Vector length must be power of 2." etc. */
}
