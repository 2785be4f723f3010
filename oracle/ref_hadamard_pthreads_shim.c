/*
 * oracle/ref_hadamard_pthreads_shim.c -- CPU ORACLE support (test infrastructure, NOT product code).
 *
 * Export wrapper around an excerpt of the REFERENCE's own source, cut out at build time by oracle/Makefile
 * (target `_ref`):
 *
 *     private/hadamard_pthreads.c:57-119   Info_t, hadamard_apply_vector, hadamard_apply_matrix, worker
 *
 * (no mx / mex call in those lines; compiled with setup_kmeans.m:55-57's flags: -pthread -O6 -DNTHREADS=n -UDEBUG
 * -DNO_UCHAR).  The reference's own thread driver, hadamard_apply_matrix_threads (:121-204), allocates with
 * mxMalloc / mxFree and therefore cannot be built here; the driver below is OURS and restates its partition --
 * n == 1 inline (:129-131), n <= NTHREADS one column per thread (:132-145), else floor(n / NTHREADS) columns for each
 * of NTHREADS workers plus one more worker for the remainder (:152-190), join all (:198-199) -- around the reference's
 * own `worker`.  So `worker` and the two kernels are the reference's bits; the partition is a restatement.
 */
#include <stdlib.h>
#include "_ref/hadamard_pthreads_57_119.inc"

void ref_hadamard_pthreads(unsigned m, unsigned n, const double *xc, double *y, unsigned nthreads)
{
    double *x = (double *)xc;
    if (nthreads < 1) nthreads = 1;
    if (n == 1) { hadamard_apply_vector(y, x, m); return; }
    const unsigned nt = n <= nthreads ? n : nthreads + 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
    Info_t *info = (Info_t *)malloc(sizeof(Info_t) * nt);
    unsigned used = 0;
    if (n <= nthreads) {
        for (unsigned j = 0; j < n; j++) {
            info[j].id = j; info[j].y = y + (size_t)j * m; info[j].x = x + (size_t)j * m; info[j].length = m; info[j].n = 1;
            pthread_create(&th[used++], NULL, worker, &info[j]);
        }
    } else {
        const unsigned nn = n / nthreads;
        unsigned j;
        for (j = 0; j < nthreads; j++) {
            info[j].id = j; info[j].y = y + (size_t)j * nn * m; info[j].x = x + (size_t)j * nn * m; info[j].length = m; info[j].n = nn;
            pthread_create(&th[used++], NULL, worker, &info[j]);
        }
        if (nn * nthreads < n) {
            info[j].id = j; info[j].y = y + (size_t)j * nn * m; info[j].x = x + (size_t)j * nn * m; info[j].length = m;
            info[j].n = n - nn * nthreads;
            pthread_create(&th[used++], NULL, worker, &info[j]);
        }
    }
    for (unsigned j = 0; j < used; j++) pthread_join(th[j], NULL);
    free(info);
    free(th);
}
