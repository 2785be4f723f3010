"""Shared input builders for the parity tests (seeded; sizes the oracle finishes in seconds)."""
import numpy as np
import scipy.sparse as sp


def random_csc(p, n, nnz_per_col, seed, ragged=False, empty_cols=(), dtype=np.float64):
    """p x n CSC with ascending rows.  ragged: per-column count uniform in [0, 2*nnz_per_col]."""
    rng = np.random.default_rng(seed)
    indptr = [0]
    idx, val = [], []
    for i in range(n):
        if i in empty_cols:
            c = 0
        elif ragged:
            c = int(rng.integers(0, min(p, 2 * nnz_per_col) + 1))
        else:
            c = min(p, nnz_per_col)
        rows = np.sort(rng.choice(p, c, replace=False))
        idx.append(rows)
        val.append(rng.standard_normal(c) * 3.0)
        indptr.append(indptr[-1] + c)
    X = sp.csc_matrix((np.concatenate(val) if val else np.zeros(0), np.concatenate(idx).astype(np.int64)
                       if idx else np.zeros(0, np.int64), np.array(indptr, np.int64)), shape=(p, n))
    return X


def parts(X):
    return X.indptr.astype(np.uint64), X.indices.astype(np.uint64), X.data.astype(np.float64)
