#!/usr/bin/env python3
"""The reference's demo (example_sparseKMeans.m) on the MI355X engine: p=512, n=5000, k=5 Gaussian mixture,
sparsified K-means with gamma = 0.05 and 20 replicates.  The reference compares against MATLAB's dense
kmeans (212 s) and its own dense code (16.5 s); those dense paths are outside this project's scope, so this
script reports the sparsified run only (reference figure: 0.79 s on the author's 2015 machine).

    python examples/example_sparseKMeans.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import synth  # noqa: E402
from sparsifiedkmeans_amd.kmeans import kmeans_sparsified  # noqa: E402


def main():
    p, n, k = 512, 5000, 5                         # example_sparseKMeans.m:13-15
    X, true_centres, labels = synth.gmm_dense(p, n, k, seed=234)   # rng(234), :12-22
    kmeans_sparsified(X.T, k, Sparsify=True, SparsityLevel=0.05, Replicates=1, rng=0)   # warm-up (library load)
    t0 = time.time()
    IDX, C, SUMD, D, OUT, C2, IDX2, D2, SUMD2 = kmeans_sparsified(
        X.T, k, ColumnSamples=False, Display="off", Replicates=20, Sparsify=True, SparsityLevel=0.05, rng=1,
        nargout=9)                                                                       # :60-65, two-pass outputs too
    dt = time.time() - t0
    # accuracy against the planted labels (best permutation)
    from scipy.optimize import linear_sum_assignment

    M = np.zeros((k, k))
    for a, b in zip(IDX - 1, labels):
        M[a, b] += 1
    r, c = linear_sum_assignment(-M)
    print(f"sparsified k-means, gamma=0.05, 20 replicates: {dt:.2f} s, objective {OUT['objectives'].min():.3e}, "
          f"accuracy {M[r, c].sum() / n:.4f}, iterations per replicate {OUT['iterations'].tolist()}")
    err = np.abs(C[r] - true_centres.T[c]).max()
    err2 = np.abs(C2[r] - true_centres.T[c]).max()
    print(f"max |centre - planted centre| = {err:.3f} one pass, {err2:.3f} after the second pass over the unsampled "
          f"data (noise level 0.1); the dense re-assignment moves {int(np.sum(IDX2 != IDX))} of {n} points")


if __name__ == "__main__":
    main()
