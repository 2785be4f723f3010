"""Host-array front-ends with the reference's mex names and calling conventions.

Each function is what MATLAB code sees when it calls the mex file of the same name
(private/SparseMatrixMinusCluster.c, SparseMatrixInnerProduct.c, SparseMatrixColumnNormSq.c,
hadamard.c, hadamard_pthreads.c): same argument meaning, same output shapes, same error
conditions -- but the arithmetic runs in HIP kernels on the MI355X through libspkm.so.
There is no CPU fallback: without the library or a GPU these raise.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib


class Context:
    """One spkm_ctx (device + stream + scratch).  Not thread-safe; one per host thread."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        _lib.check(_lib.lib().spkm_ctx_create(int(device), C.c_void_p(stream or 0), C.byref(h)), "spkm_ctx_create")
        self.handle = h
        self.device = int(device)
        self.stream = int(stream or 0)          # the hipStream_t the library enqueues on (0 = the default stream)

    def sync(self):
        _lib.check(_lib.lib().spkm_ctx_sync(self.handle), "spkm_ctx_sync")

    def reload_switches(self):
        """re-read the SPKM_* A/B switches from the environment (they are read once, when the context is created)"""
        _lib.check(_lib.lib().spkm_ctx_reload_switches(self.handle), "spkm_ctx_reload_switches")

    def device_info(self) -> dict:
        a = (C.c_int64 * 4)()
        _lib.check(_lib.lib().spkm_device_info(self.handle, a))
        return dict(cus=a[0], lds_bytes=a[1], mem_bytes=a[2], wave=a[3])

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().spkm_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def _ptr(a: np.ndarray):
    return C.c_void_p(a.ctypes.data)


def _csc_parts(X):
    """-> (p, n, jc u64, ir u64, x f64) of a scipy CSC matrix in canonical (sorted, summed) form."""
    if not sp.issparse(X):
        # SparseMatrixMinusCluster.c:72-75 / InnerProduct.c:62-65 / ColumnNormSq.c:55-58
        raise TypeError("Requires first input to be a sparse matrix")
    X = sp.csc_matrix(X, dtype=np.float64)
    if not X.has_canonical_format:
        X = X.copy()
        X.sum_duplicates()
        X.sort_indices()
    p, n = X.shape
    return (p, n, np.ascontiguousarray(X.indptr, np.uint64), np.ascontiguousarray(X.indices, np.uint64),
            np.ascontiguousarray(X.data, np.float64))


def SparseMatrixMinusCluster(X, c, beta=None, ctx: Context | None = None) -> np.ndarray:
    """dist = SparseMatrixMinusCluster(X, C[, beta]) -> K x n Euclidean distances over supp(X(:,i))."""
    p, n, jc, ir, x = _csc_parts(X)   # argument checks come before any device work
    ctx = ctx or default_context()
    c = np.asarray(c, np.float64)
    if c.ndim == 1:
        c = c[:, None]
    K = c.shape[1]
    Cf = np.ascontiguousarray(c.T)  # column-major p x K
    out = np.zeros((n, K))          # column-major K x n
    b = None if beta is None else C.c_double(float(beta))
    st = _lib.lib().spkm_SparseMatrixMinusCluster_host(
        ctx.handle, p, n, _ptr(jc), _ptr(ir), _ptr(x), c.shape[0], K, _ptr(Cf),
        C.byref(b) if b is not None else None, _ptr(out))
    _lib.check(st, "SparseMatrixMinusCluster")
    return np.ascontiguousarray(out.T)


def SparseMatrixInnerProduct(X, c, ctx: Context | None = None):
    """[innerProd, normX2] = SparseMatrixInnerProduct(X, c) -> two length-n vectors."""
    p, n, jc, ir, x = _csc_parts(X)
    ctx = ctx or default_context()
    c = np.ascontiguousarray(np.asarray(c, np.float64).ravel())
    if c.size < p:
        raise ValueError("Center vector must have at least p entries")
    ip, nx2 = np.zeros(n), np.zeros(n)
    st = _lib.lib().spkm_SparseMatrixInnerProduct_host(ctx.handle, p, n, _ptr(jc), _ptr(ir), _ptr(x), _ptr(c),
                                                       _ptr(ip), _ptr(nx2))
    _lib.check(st, "SparseMatrixInnerProduct")
    return ip, nx2


def SparseMatrixColumnNormSq(X, ctx: Context | None = None) -> np.ndarray:
    """normX2 = SparseMatrixColumnNormSq(X) = sum(X.^2, 1)."""
    p, n, jc, ir, x = _csc_parts(X)
    ctx = ctx or default_context()
    out = np.zeros(n)
    st = _lib.lib().spkm_SparseMatrixColumnNormSq_host(ctx.handle, n, _ptr(jc), _ptr(x), _ptr(out))
    _lib.check(st, "SparseMatrixColumnNormSq")
    return out


def _hadamard(x, fn, ctx):
    if sp.issparse(x):
        raise TypeError("Input must be full")  # hadamard.c:137-140
    x = np.asarray(x)
    if np.iscomplexobj(x):
        raise TypeError("Input must be real")  # hadamard.c:134-136
    x = np.asarray(x, np.float64)
    vec = x.ndim == 1
    if vec:
        x = x[:, None]
    m, n = x.shape
    xin = np.ascontiguousarray(x.T)  # column-major m x n
    out = np.empty_like(xin)
    ctx = ctx or default_context()
    _lib.check(fn(ctx.handle, m, n, _ptr(xin), _ptr(out)), "hadamard")
    y = np.ascontiguousarray(out.T)
    return y[:, 0] if vec else y


def hadamard(x, ctx: Context | None = None) -> np.ndarray:
    """y = hadamard(x): unnormalised Walsh-Hadamard transform of each column (hadamard.c)."""
    return _hadamard(x, _lib.lib().spkm_hadamard_host, ctx)


def hadamard_pthreads(x, ctx: Context | None = None) -> np.ndarray:
    """y = hadamard_pthreads(x): same transform (hadamard_pthreads.c); one kernel serves both."""
    return _hadamard(x, _lib.lib().spkm_hadamard_pthreads_host, ctx)
