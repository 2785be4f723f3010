#!/usr/bin/env python3
"""Writes tests/golden/ref_fwht_*.npz: seeded inputs + outputs of the REFERENCE's own FWHT code.

Unlike make_fixtures.py (outputs of our oracle), these vectors come from the reference itself: oracle/_ref holds
private/hadamard.c:57-92 and private/hadamard_pthreads.c:57-119 compiled from where the files lie under
/root/reference with setup_kmeans.m:53,55-57's flags (oracle/Makefile; those line ranges need no mex.h).  They pin rows
a13 / a14 of SURVEY section 8 for the oracle AND -- through tests/test_gpu_ops.py -- for the HIP kernel, on machines
where /root/reference does not exist.  Shapes follow SURVEY 8(c)(i): m in {2, 8, 64, 1024, 4096}, n in {1, 3, 13, 17},
NTHREADS in {1, 4, 8}.  Run from the repo root in the build container:  python tests/golden/make_ref_fixtures.py
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


if __name__ == "__main__":
    main()
