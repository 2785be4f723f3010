#!/usr/bin/env python3
"""Writes tests/golden/ref_fwht_*.npz, ref_dist_*.npz, ref_beta_*.npz, ref_ip_*.npz: seeded inputs + outputs of the
REFERENCE's own code (FWHT kernels and, since round 4, the three Sparse*.c loops).

Unlike make_fixtures.py (outputs of our oracle), these vectors come from the reference itself: oracle/_ref holds
private/hadamard.c:57-92 and private/hadamard_pthreads.c:57-119 compiled from where the files lie under
/root/reference with setup_kmeans.m:53,55-57's flags (oracle/Makefile; those line ranges need no mex.h).  They pin rows
a13 / a14 of SURVEY section 8 for the oracle AND -- through tests/test_gpu_ops.py -- for the HIP kernel, on machines
where /root/reference does not exist.  Shapes follow SURVEY 8(c)(i): m in {2, 8, 64, 1024, 4096}, n in {1, 3, 13, 17},
NTHREADS in {1, 4, 8}.

Round 4: ref_dist_* / ref_beta_* / ref_ip_* are outputs of private/SparseMatrixMinusCluster.c:131-183 (`switch (K)`) and
:121-129 (beta), SparseMatrixInnerProduct.c:86-100, SparseMatrixColumnNormSq.c:70-77 (oracle/ref_sparse_shim.c; built with
setup_kmeans.m:19,26,33's `-O`).  They pin rows a1-a3, a11, a12: K in {1, 2, 3, 4, 7, 10, 100} (every branch of the
switch), ragged and empty columns, duplicate centroids (exact ties for MATLAB's first-index `min`), fixed-stride
columns of s = 51 entries at p = 1024 (the benchmark's point shape: what the fused screen path takes).
Run from the repo root in the build container:  python tests/golden/make_ref_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    O.build(force=True)
    assert O.ref_available("native") and O.ref_available("pthreads"), "needs /root/reference (oracle/Makefile, _ref)"
    rng = np.random.default_rng(20260929)
    for m, n in [(2, 1), (8, 3), (64, 13), (1024, 17), (4096, 3)]:
        x = rng.standard_normal((m, n)) * np.exp(rng.uniform(-20, 20, (1, n)))     # columns of very different scale
        y = O.ref_fwht(x, "native")                                               # hadamard.c, -O3 -march=native
        for nt in (1, 4, 8):                                                       # hadamard_pthreads.c worker + kernels
            assert np.array_equal(O.ref_fwht(x, "pthreads", nt), y)
        np.savez_compressed(os.path.join(HERE, f"ref_fwht_{m}x{n}.npz"), kind="ref_fwht", x=x, out=y)
    sparse_fixtures()


def sparse_fixtures():
    sys.path.insert(0, os.path.dirname(HERE))
    from util import parts, random_csc

    assert O.ref_available("sparse") and O.ref_available("sparse_O2")
    #        p,    n,   K, nnz/col, ragged, empty columns, duplicate-centroid pairs (k_dup <- k_src)
    cases = [(2, 1, 1, 1, False, (), ()),
             (64, 257, 1, 7, True, (0, 256), ()),
             (64, 257, 2, 7, True, (3,), ((1, 0),)),
             (64, 257, 3, 7, True, (0,), ((2, 0),)),
             (512, 300, 4, 26, True, (17,), ((3, 1),)),
             (512, 257, 7, 26, False, (), ((5, 2),)),
             (1024, 257, 10, 51, False, (), ((9, 4),)),
             (1024, 300, 100, 51, False, (), ((20, 3), (99, 3), (17, 16)))]
    for p, n, K, s, ragged, empty, dups in cases:
        X = random_csc(p, n, s, seed=20260929 + 31 * p + K, ragged=ragged, empty_cols=empty)
        C = np.random.default_rng(K + p).standard_normal((p, K)) * 2.0
        for kd, ks in dups:
            C[:, kd] = C[:, ks]
        jc, ir, x = parts(X)
        D = O.ref_dist_csc(p, n, jc, ir, x, C)                                  # gcc -O   (setup_kmeans.m:19)
        assert np.array_equal(O.ref_dist_csc(p, n, jc, ir, x, C, "sparse_O2"), D)   # mex's stock -O2: same bits
        np.savez_compressed(os.path.join(HERE, f"ref_dist_p{p}_n{n}_K{K}.npz"), kind="ref_dist", p=p, n=n, K=K,
                            jc=X.indptr.astype(np.int64), ir=X.indices.astype(np.int32), x=x, C=C, dist=D)
    p, n = 128, 500
    X = random_csc(p, n, 9, seed=5, ragged=True, empty_cols=(11,))
    c = np.random.default_rng(1).standard_normal(p)
    jc, ir, x = parts(X)
    out = {f"dist_beta_{str(b).replace('.', 'p').replace('-', 'm')}": O.ref_dist_csc_beta(n, jc, ir, x, c, b)
           for b in (0.37, 1.0, -0.5)}
    np.savez_compressed(os.path.join(HERE, f"ref_beta_p{p}_n{n}.npz"), kind="ref_beta", p=p, n=n, betas=[0.37, 1.0, -0.5],
                        jc=X.indptr.astype(np.int64), ir=X.indices.astype(np.int32), x=x, c=c, **out)
    p, n = 700, 1000
    X = random_csc(p, n, 35, seed=11, ragged=True, empty_cols=(3, 999))
    c = np.random.default_rng(2).standard_normal(p)
    jc, ir, x = parts(X)
    ip, nx2 = O.ref_innerprod_csc(n, jc, ir, x, c)
    nsq = O.ref_colnormsq_csc(n, jc, x)
    np.savez_compressed(os.path.join(HERE, f"ref_ip_p{p}_n{n}.npz"), kind="ref_ip", p=p, n=n,
                        jc=X.indptr.astype(np.int64), ir=X.indices.astype(np.int32), x=x, c=c, ip=ip, nx2=nx2, nsq=nsq)


if __name__ == "__main__":
    main()
