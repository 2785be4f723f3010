#!/usr/bin/env python3
"""Writes tests/golden/*.npz: seeded inputs + outputs of the CPU oracle.

These are REGRESSION fixtures of the oracle (and therefore of the HIP path, which the -m gpu
tests hold bit-exact to the oracle) -- not reference outputs: the reference ships no golden
vectors, and its C cannot be built here without MATLAB's mex.h (DESIGN.md "Oracle").
Run from the repo root:  python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402
from util import parts, random_csc  # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    for name, (p, n, K, s, ragged) in {"dist_k1": (64, 33, 1, 6, True), "dist_k3": (64, 33, 3, 6, True),
                                       "dist_k7": (512, 40, 7, 26, False), "dist_k100": (1024, 24, 100, 51, False)}.items():
        X = random_csc(p, n, s, seed=len(name) * 7 + K, ragged=ragged, empty_cols=(2,))
        C = rng.standard_normal((p, K))
        jc, ir, x = parts(X)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), kind="dist", p=p, n=n, jc=jc, ir=ir, x=x, C=C,
                            out=O.dist_csc(p, n, jc, ir, x, C))
    for m in (8, 64, 1024):
        x = rng.standard_normal((m, 3))
        np.savez_compressed(os.path.join(HERE, f"fwht_{m}.npz"), kind="fwht", x=x, out=O.fwht(x))
    p, n, K = 512, 300, 12
    X = random_csc(p, n, 26, seed=99)
    C = rng.standard_normal((p, K)) * 0.05
    C[:, 7] = C[:, 2]                      # a tie: first index must win
    jc, ir, x = parts(X)
    a, d = O.assign(p, n, jc, ir, x, C, 26 / 512)
    np.savez_compressed(os.path.join(HERE, "assign_ties.npz"), kind="assign", p=p, n=n, jc=jc, ir=ir, x=x, C=C,
                        gamma=26 / 512, assign=a, out=d)


if __name__ == "__main__":
    main()
