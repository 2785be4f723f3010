/*
 * oracle/ref_hadamard_shim.c -- CPU ORACLE support (test infrastructure, NOT product code).
 *
 * Export wrapper around an excerpt of the REFERENCE's own source, cut out at build time by oracle/Makefile
 * (target `_ref`) from where the file lies under /root/reference:
 *
 *     private/hadamard.c:57-92   hadamard_apply_vector, hadamard_apply_matrix
 *
 * Those lines contain no mx / mex call, so they compile without MATLAB's mex.h (which this image lacks) and without any
 * stand-in header.  The excerpt is written to oracle/_ref/ (git-ignored) and never enters the repository; this file
 * is ours and only gives the two functions a C-ABI name that ctypes can bind.  Flags: setup_kmeans.m:53
 * (-O3 -march=native -DNO_UCHAR) for libref_hadamard.so; a second, portable build (-O3 only) travels to the GPU box --
 * the transform is add/sub only, so the two must agree bit for bit (tests/test_oracle.py checks that here).
 *
 * What this pins: rows a13 / a14 of SURVEY section 8 (the butterfly and the column loop).  What it cannot pin: the
 * gateways (mexFunction, checkPowerTwo call mexErrMsgTxt) and hadamard_apply_matrix_threads (mxMalloc / mxFree).
 */
#include "_ref/hadamard_57_92.inc"

void ref_hadamard(unsigned m, unsigned n, const double *x, double *y)
{
    hadamard_apply_matrix(y, (double *)x, m, n);
}
