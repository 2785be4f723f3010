/* COMPILE CHECK ONLY -- not MATLAB's mex.h, not a stand-in for building or running anything.
 *
 * Declarations (no definitions) of the handful of documented MATLAB C Matrix / MEX API functions that OUR gateways under
 * matlab/ call, with the signatures MathWorks documents for -largeArrayDims builds (mwSize = mwIndex = size_t).
 * tests/test_abi_and_host.py runs `gcc -fsyntax-only -Wall -Werror` over matlab/ *.c against this file so that every
 * spkm_* call in the gateways is type-checked against include/spkm.h.  Nothing is ever
 * linked against it; the oracle does not see it; the reference's own C files are never compiled with it (the oracle's
 * reference build cuts mex-free line ranges instead: oracle/Makefile).  mxArray stays opaque on purpose. */
#ifndef SPKM_TESTS_MEX_DECLS_H
#define SPKM_TESTS_MEX_DECLS_H
#include <stddef.h>
#include <stdbool.h>

typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef size_t mwIndex;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;

/* gateway entry point every mex file defines */
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]);

/* mex* */
void mexErrMsgTxt(const char *msg);
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
int mexPrintf(const char *fmt, ...);
int mexAtExit(void (*fn)(void));
int mexCallMATLAB(int nlhs, mxArray *plhs[], int nrhs, mxArray *prhs[], const char *name);

/* queries */
size_t mxGetM(const mxArray *a);
size_t mxGetN(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
double *mxGetPr(const mxArray *a);
mwIndex *mxGetIr(const mxArray *a);
mwIndex *mxGetJc(const mxArray *a);
double mxGetScalar(const mxArray *a);
int mxGetString(const mxArray *a, char *buf, mwSize buflen);
bool mxIsSparse(const mxArray *a);
bool mxIsComplex(const mxArray *a);
bool mxIsDouble(const mxArray *a);
bool mxIsEmpty(const mxArray *a);

/* creation / ownership */
mxArray *mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateSparse(mwSize m, mwSize n, mwSize nzmax, mxComplexity flag);
void mxDestroyArray(mxArray *a);
void *mxMalloc(mwSize n);
void *mxCalloc(mwSize n, mwSize size);
void *mxRealloc(void *p, mwSize n);
void mxFree(void *p);
void mxSetPr(mxArray *a, double *pr);
void mxSetIr(mxArray *a, mwIndex *ir);
void mxSetNzmax(mxArray *a, mwSize nzmax);
#endif
