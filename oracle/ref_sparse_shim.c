/*
 * oracle/ref_sparse_shim.c -- CPU ORACLE support (test infrastructure, NOT product code).
 *
 * Export wrappers around excerpts of the REFERENCE's own source, cut out at build time by oracle/Makefile
 * (target `_ref`) from where the files lie under /root/reference:
 *
 *     private/SparseMatrixMinusCluster.c:121-129   the beta branch's column loop                       (row a3)
 *     private/SparseMatrixMinusCluster.c:131-183   `switch (K)`: K = 1 / 2 / 3 unrolled + the general K  (rows a1, a2)
 *     private/SparseMatrixInnerProduct.c:86-100    inner product + squared norm per column             (row a11)
 *     private/SparseMatrixColumnNormSq.c:70-77     squared norm per column                             (row a12)
 *
 * Those line ranges lie inside each file's mexFunction but contain no mx / mex call and no type from MATLAB's mex.h
 * (which this image lacks): they refer only to the function's locals  x, ir, jc, center, distance, dist, distHelper,
 * distArray, beta, innerProd, normX, nrm, inrProd, p, n, K, i, j, k.  Each wrapper below declares exactly those
 * locals, in C standard types -- mwIndex / mwSize are size_t under `mex -largeArrayDims` (setup_kmeans.m:19,26,33) --
 * fills the pointers from its arguments where the gateway fills them from mxGetPr / mxGetIr / mxGetJc
 * (SparseMatrixMinusCluster.c:91-114), and includes the excerpt as its body.  No stand-in header is involved; the
 * excerpts are written to oracle/_ref/ (git-ignored), checked against oracle/ref_excerpts.sha256 and deleted again once
 * the library is linked: only this file, which is ours, is in the repository.
 *
 * What this pins: the arithmetic and its order for rows a1-a3, a11, a12 of SURVEY section 8.  What it cannot pin: the
 * gateways' argument checks (mexErrMsgTxt), and everything that is MATLAB code (rows a4-a10, a16).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

/* SparseMatrixMinusCluster(X, center): distance is K x n, column-major.  Locals as in :48-55. */
void ref_dist_csc(size_t p_, size_t n_, size_t K_, const size_t *jc_, const size_t *ir_, const double *x_,
                  const double *center_, double *distance_)
{
    double *distance = distance_, *center = (double *)center_, *x = (double *)x_;
    register double dist;
    double *distArray = NULL;
    double distHelper[3];
    size_t *ir = (size_t *)ir_, *jc = (size_t *)jc_;
    size_t p = p_, n = n_, i, j, k, K = K_;
    if (K > 3) distArray = (double *)malloc(K * sizeof(double)); /* :108-110 uses mxMalloc */
    {
#include "_ref/smc_131_183.inc"
    }
    if (K > 3) free(distArray); /* :185-186 uses mxFree */
}

/* SparseMatrixMinusCluster(X, center, beta), K == 1 (:119-120 rejects anything else before the loop). */
void ref_dist_csc_beta(size_t n_, const size_t *jc_, const size_t *ir_, const double *x_, const double *center_,
                       double beta_, double *distance_)
{
    double *distance = distance_, *center = (double *)center_, *x = (double *)x_;
    register double dist;
    double beta = beta_;
    size_t *ir = (size_t *)ir_, *jc = (size_t *)jc_;
    size_t n = n_, i, j;
    {
#include "_ref/smc_121_129.inc"
    }
}

/* [innerProd, normX2] = SparseMatrixInnerProduct(X, c).  Locals as in :44-47. */
void ref_innerprod_csc(size_t n_, const size_t *jc_, const size_t *ir_, const double *x_, const double *center_,
                       double *innerProd_, double *normX_)
{
    double *innerProd = innerProd_, *normX = normX_, *center = (double *)center_, *x = (double *)x_;
    double nrm, inrProd;
    size_t *ir = (size_t *)ir_, *jc = (size_t *)jc_;
    size_t n = n_, i, j;
    {
#include "_ref/smip_86_100.inc"
    }
}

/* normX2 = SparseMatrixColumnNormSq(X). */
void ref_colnormsq_csc(size_t n_, const size_t *jc_, const double *x_, double *normX_)
{
    double *normX = normX_, *x = (double *)x_;
    double nrm;
    size_t *jc = (size_t *)jc_;
    size_t n = n_, i, j;
    {
#include "_ref/smcn_70_77.inc"
    }
}
