/*
 * oracle/orc_fwht.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's fast Walsh-Hadamard transform
 * (private/hadamard.c, private/hadamard_pthreads.c).  PARITY UNPINNED: see the
 * header of orc_sparse.c -- the reference C needs MATLAB's mex.h and cannot be
 * built here, and it ships no golden vectors.  Pinned by the cited source, by
 * oracle/numpy_ref.py (bit-for-bit) and by the identities the reference
 * documents (hadamard.c:8-11,17-23: symmetric, orthogonal up to 1/m, equals the
 * Sylvester/"hadamard"-ordered Walsh-Hadamard matrix times x).
 *
 * The transform is add/sub only, so compiler flags cannot change its bits
 * (reference builds it with -O3 -march=native, setup_kmeans.m:53).  Here:
 * gcc -O3 -pthread, with an AVX2 clone selected at load time so the prebuilt
 * .so is safe on a host CPU that differs from the build machine.
 */
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>

/* private/hadamard.c:57-78  ==  private/hadamard_pthreads.c:69-90.
 * Stage bit=1 out of place (pairs j, j+1), then stages bit=2,4,...,m/2 in
 * place: for every j with (bit & j)==0, k=j|bit: (y[j], y[k]) <- (y[j]+y[k], y[j]-y[k]).
 * Unnormalised; m must be a power of two > 1 (checked by the caller, :97-111). */
__attribute__((target_clones("avx2", "default")))
static void fwht_vector(double *y, const double *x, unsigned m)
{
    for (unsigned j = 0; j < m; j += 2) {
        const double a = x[j], b = x[j + 1];
        y[j] = a + b;
        y[j + 1] = a - b;
    }
    for (unsigned bit = 2; bit < m; bit <<= 1) {
        /* same pairs as the reference's `if ((bit & j)==0)` scan, visited in the same order */
        for (unsigned base = 0; base < m; base += 2 * bit) {
            for (unsigned j = base; j < base + bit; j++) {
                const unsigned k = j | bit;
                const double t = y[j];
                y[j] = t + y[k];
                y[k] = t - y[k];
            }
        }
    }
}

/* private/hadamard.c:86-92 (column loop).  size_t indexing: the reference's
 * `unsigned j*m` wraps at 2^32 elements (hadamard.c:90); callers of the oracle
 * stay below that so the two agree. */
void orc_fwht(unsigned m, size_t n, const double *x, double *y)
{
    for (size_t j = 0; j < n; j++) fwht_vector(y + j * (size_t)m, x + j * (size_t)m, m);
}

/* hadamard.c:97-111 / hadamard_pthreads.c:209-223: 0 ok, 1 = "must be greater than 1", 2 = "must be power of 2" */
int orc_check_pow2(unsigned m)
{
    if (m <= 1) return 1;
    while ((m & 1) == 0) m >>= 1;
    return m > 1 ? 2 : 0;
}

typedef struct { const double *x; double *y; unsigned m; size_t n; } job_t;
static void *worker(void *arg)
{
    job_t *jb = (job_t *)arg;
    orc_fwht(jb->m, jb->n, jb->x, jb->y);
    return NULL;
}

/* private/hadamard_pthreads.c:121-204: static column partition.
 *   n==1            -> inline                                      (:129-130)
 *   n<=NTHREADS     -> one thread per column                       (:132-145)
 *   else            -> NTHREADS workers of floor(n/NTHREADS) columns (:152,163-175)
 *                      plus one more for the n mod NTHREADS remainder (:179-190)
 * then join all (:198-199).  Output is identical to orc_fwht (disjoint columns). */
void orc_fwht_threads(unsigned m, size_t n, const double *x, double *y, unsigned nthreads)
{
    if (n == 1 || nthreads <= 1) { orc_fwht(m, n, x, y); return; }
    size_t nworkers, per;
    if (n <= nthreads) { nworkers = n; per = 1; }
    else { per = n / nthreads; nworkers = nthreads + ((n % nthreads) ? 1 : 0); }
    job_t *jobs = (job_t *)malloc(nworkers * sizeof(job_t));
    pthread_t *th = (pthread_t *)malloc(nworkers * sizeof(pthread_t));
    size_t col = 0;
    for (size_t t = 0; t < nworkers; t++) {
        size_t cnt = per;
        if (n > nthreads && t == nthreads) cnt = n - col; /* remainder worker */
        jobs[t].x = x + col * (size_t)m;
        jobs[t].y = y + col * (size_t)m;
        jobs[t].m = m;
        jobs[t].n = cnt;
        col += cnt;
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (size_t t = 0; t < nworkers; t++) pthread_join(th[t], NULL);
    free(jobs);
    free(th);
}

/* kmeans_sparsified.m:241-248,286-295: mix(X) = hadamard(DD*[X;0]) / sqrt(p2),
 * after X = X*(1+2*eps) (:292).  x is p x n (column-major), d is the +-1 sign
 * vector of length p2, y is p2 x n.  scale is sqrt(p2) computed by the caller.
 * DD*x is an exact multiply by +-1; the zero rows stay +0 (d*0 = +-0, and
 * -0 + anything nonzero behaves as +0; for an all-zero pad the first stage
 * gives 0+0 / 0-0 exactly as MATLAB's spdiags product would). */
void orc_mix(unsigned p, unsigned p2, size_t n, const double *x, const double *d,
             double premul, double scale, double *y, double *tmp /* p2 */)
{
    for (size_t j = 0; j < n; j++) {
        for (unsigned r = 0; r < p; r++) tmp[r] = d[r] * (x[j * (size_t)p + r] * premul);
        for (unsigned r = p; r < p2; r++) tmp[r] = d[r] * 0.;
        fwht_vector(y + j * (size_t)p2, tmp, p2);
        for (unsigned r = 0; r < p2; r++) y[j * (size_t)p2 + r] /= scale;
    }
}
