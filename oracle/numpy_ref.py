"""Second, independent restatement of the reference arithmetic in numpy.

TEST INFRASTRUCTURE ONLY (see oracle/orc_sparse.c header; PARITY UNPINNED).
Written to be obviously-correct rather than fast: each (column, k) accumulator
is updated entry by entry in storage order with separately rounded subtract,
multiply and add (numpy never fuses them).  tests/ require the C oracle to
match this bit-for-bit.
"""
from __future__ import annotations

import numpy as np


def dist_csc(p, n, jc, ir, x, Cmat):
    """private/SparseMatrixMinusCluster.c:117-184 -> K x n."""
    Cmat = np.asarray(Cmat, np.float64).reshape(p, -1)
    K = Cmat.shape[1]
    out = np.zeros((K, n))
    for i in range(n):
        acc = np.zeros(K)
        for j in range(int(jc[i]), int(jc[i + 1])):
            d = np.float64(x[j]) - Cmat[int(ir[j]), :]
            acc = acc + d * d
        out[:, i] = np.sqrt(acc)
    return out


def dist_csc_beta(n, jc, ir, x, c, beta):
    """private/SparseMatrixMinusCluster.c:118-129."""
    b = np.float64(beta) * np.float64(-2.0)
    out = np.zeros(n)
    c = np.asarray(c, np.float64).ravel()
    for i in range(n):
        acc = np.float64(0.0)
        for j in range(int(jc[i]), int(jc[i + 1])):
            xv, cv = np.float64(x[j]), c[int(ir[j])]
            acc = acc + (((xv * xv) + ((b * xv) * cv)) + (cv * cv))
        out[i] = np.sqrt(acc)
    return out


def innerprod_csc(n, jc, ir, x, c):
    """private/SparseMatrixInnerProduct.c:87-100."""
    c = np.asarray(c, np.float64).ravel()
    ip, nx2 = np.zeros(n), np.zeros(n)
    for i in range(n):
        a, b = np.float64(0.0), np.float64(0.0)
        for j in range(int(jc[i]), int(jc[i + 1])):
            xv = np.float64(x[j])
            a = a + xv * c[int(ir[j])]
            b = b + xv * xv
        ip[i], nx2[i] = a, b
    return ip, nx2


def colnormsq_csc(n, jc, x):
    """private/SparseMatrixColumnNormSq.c:71-77."""
    out = np.zeros(n)
    for i in range(n):
        b = np.float64(0.0)
        for j in range(int(jc[i]), int(jc[i + 1])):
            xv = np.float64(x[j])
            b = b + xv * xv
        out[i] = b
    return out


def min_cols(dist):
    """findClusterAssignments.m:169; first index on ties; 0-based index."""
    a = np.argmin(dist, axis=0).astype(np.int32)  # numpy argmin: first occurrence
    return dist[a, np.arange(dist.shape[1])], a


def fwht(x):
    """private/hadamard.c:57-92 -- stage order bit=1,2,4,...,m/2."""
    y = np.array(x, np.float64, copy=True)
    if y.ndim == 1:
        y = y[:, None]
    m, n = y.shape
    bit = 1
    while bit < m:
        v = y.reshape(m // (2 * bit), 2, bit, n)
        a, b = v[:, 0].copy(), v[:, 1].copy()
        v[:, 0] = a + b
        v[:, 1] = a - b
        bit *= 2
    return y


def sylvester(m):
    """Sylvester Hadamard matrix (hadamard.c:17-23: hadamard(x) == H_m x up to roundoff)."""
    H = np.array([[1.0]])
    while H.shape[0] < m:
        H = np.block([[H, H], [H, -H]])
    return H


def dense_assign(X, Cmat):
    """private/findClusterAssignments.m:157-165 (expanded quadratic, dense X) + :168-171 (min, first index):
    distances(k,:) = nrm2 - 2*(X'*c_k)' + norm(c_k)^2 ; sqrt ; [distances, assignments] = min(distances,[],1).
    X is p x n, Cmat p x K.  Returns (assignments 0-based, distances).  The summation order inside X'*c and
    norm() is the BLAS's in the reference (undefined): tests compare to a tolerance.  Rounded values below 0
    are clamped before the sqrt (MATLAB would go complex)."""
    X = np.asarray(X, np.float64)
    Cmat = np.asarray(Cmat, np.float64)
    nrm2 = np.sum(X * X, axis=0)
    K = Cmat.shape[1]
    d = np.empty((K, X.shape[1]))
    for k in range(K):
        c = Cmat[:, k]
        d[k] = nrm2 - 2.0 * (X.T @ c) + np.linalg.norm(c) ** 2
    d = np.sqrt(np.maximum(d, 0.0))
    a = np.argmin(d, axis=0)
    return a, d[a, np.arange(X.shape[1])], d


def two_pass_centers(X, assign0, K):
    """kmeans_sparsified.m:543-550: centers_twoPass(:,k) = mean(full(XFull(:,ind)),2), zeros for empty clusters."""
    X = np.asarray(X, np.float64)
    out = np.zeros((X.shape[0], K))
    for k in range(K):
        ind = np.flatnonzero(np.asarray(assign0) == k)
        if ind.size:
            out[:, k] = X[:, ind].mean(axis=1)
    return out
