"""ctypes front-end to the CPU oracle (oracle/_build/liborc.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by sparsifiedkmeans_amd/.
PARITY: rows a1-a3, a11-a14 of SURVEY section 8 are PINNED -- the restatement is checked bit for bit against the
reference's own loops compiled from where they lie (oracle/_ref: ref_sparse_shim.c, ref_hadamard*_shim.c, the
ref_* functions at the bottom of this file) and against tests/golden/ref_*.npz written by them.  Rows a4-a10, a16 are
MATLAB code (no MATLAB / Octave in this image): restated from the cited lines, structurally unpinnable.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liborc.so")

_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_sz = C.c_size_t


def build(force: bool = False) -> str:
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_dist_csc.argtypes = [_sz, _sz, _sz, _u64p, _u64p, _f64p, _f64p, _f64p]
        L.orc_dist_csc_beta.argtypes = [_sz, _u64p, _u64p, _f64p, _f64p, C.c_double, _f64p]
        L.orc_innerprod_csc.argtypes = [_sz, _u64p, _u64p, _f64p, _f64p, _f64p, _f64p]
        L.orc_colnormsq_csc.argtypes = [_sz, _u64p, _f64p, _f64p]
        L.orc_min_cols.argtypes = [_sz, _sz, _f64p, _f64p, _i32p]
        L.orc_assign.argtypes = [_sz, _sz, _sz, _u64p, _u64p, _f64p, _f64p, C.c_double, _f64p, _i32p, _f64p]
        L.orc_dist_sparse_centers.argtypes = [_sz, _sz, _sz, _u64p, _u64p, _f64p, _u64p, _u64p, _f64p,
                                              C.c_double, _f64p]
        L.orc_accumulate.argtypes = [_sz, _sz, _sz, _u64p, _u64p, _f64p, _i32p, _f64p, _f64p, _i64p]
        L.orc_finalize_centers.argtypes = [_sz, _sz, _f64p, _f64p, _i64p, C.c_double, _f64p]
        L.orc_fro_diff.argtypes = [_sz, _f64p, _f64p]
        L.orc_fro_diff.restype = C.c_double
        L.orc_obj.argtypes = [_sz, _f64p]
        L.orc_obj.restype = C.c_double
        L.orc_lloyd.argtypes = [_sz, _sz, _sz, _u64p, _u64p, _f64p, C.c_double, C.c_int, C.c_int, C.c_double,
                                _f64p, _i32p, _f64p, _f64p, _f64p]
        L.orc_lloyd.restype = C.c_int
        L.orc_lloyd_ex.argtypes = [_sz, _sz, C.POINTER(C.c_size_t), _u64p, _u64p, _f64p, C.c_double, C.c_int, C.c_int,
                                   C.c_double, C.c_int, _f64p, _i32p, _f64p, _f64p, _f64p, C.POINTER(C.c_int)]
        L.orc_lloyd_ex.restype = C.c_int
        L.orc_lloyd_plain.argtypes = L.orc_lloyd_ex.argtypes
        L.orc_lloyd_plain.restype = C.c_int
        L.orc_finalize_plain_mean.argtypes = [_sz, _sz, _f64p, _i64p, _f64p]
        L.orc_lloyd_iter_threads.argtypes = [_sz, _sz, _sz, _u64p, _u64p, _f64p, C.c_double, C.c_int, _f64p, _i32p,
                                             _f64p, C.c_uint]
        L.orc_fwht.argtypes = [C.c_uint, _sz, _f64p, _f64p]
        L.orc_fwht_threads.argtypes = [C.c_uint, _sz, _f64p, _f64p, C.c_uint]
        L.orc_check_pow2.argtypes = [C.c_uint]
        L.orc_check_pow2.restype = C.c_int
        L.orc_mix.argtypes = [C.c_uint, C.c_uint, _sz, _f64p, _f64p, C.c_double, C.c_double, _f64p, _f64p]
        _lib = L
    return _lib


def _csc(jc, ir, x):
    return (np.ascontiguousarray(jc, np.uint64), np.ascontiguousarray(ir, np.uint64),
            np.ascontiguousarray(x, np.float64))


def _colmajor(M):
    """p x K array -> flat column-major buffer (MATLAB layout)."""
    return np.ascontiguousarray(np.asarray(M, np.float64).T).ravel()


def dist_csc(p, n, jc, ir, x, Cmat):
    """SparseMatrixMinusCluster(X, C): returns K x n array (as numpy [K, n])."""
    jc, ir, x = _csc(jc, ir, x)
    Cmat = np.asarray(Cmat, np.float64).reshape(p, -1)
    K = Cmat.shape[1]
    out = np.zeros(n * K)
    lib().orc_dist_csc(p, n, K, jc, ir, x, _colmajor(Cmat), out)
    return out.reshape(n, K).T.copy()


def dist_csc_beta(n, jc, ir, x, c, beta):
    jc, ir, x = _csc(jc, ir, x)
    out = np.zeros(n)
    lib().orc_dist_csc_beta(n, jc, ir, x, np.ascontiguousarray(c, np.float64).ravel(), float(beta), out)
    return out


def innerprod_csc(n, jc, ir, x, c):
    jc, ir, x = _csc(jc, ir, x)
    ip, nx2 = np.zeros(n), np.zeros(n)
    lib().orc_innerprod_csc(n, jc, ir, x, np.ascontiguousarray(c, np.float64).ravel(), ip, nx2)
    return ip, nx2


def colnormsq_csc(n, jc, x):
    jc = np.ascontiguousarray(jc, np.uint64)
    x = np.ascontiguousarray(x, np.float64)
    out = np.zeros(n)
    lib().orc_colnormsq_csc(n, jc, x, out)
    return out


def min_cols(dist):
    """MATLAB [d,a]=min(dist,[],1) on a K x n array; a is 0-based."""
    K, n = dist.shape
    flat = np.ascontiguousarray(dist.T).ravel()
    mind, a = np.zeros(n), np.zeros(n, np.int32)
    lib().orc_min_cols(K, n, flat, mind, a)
    return mind, a


def assign(p, n, jc, ir, x, Cmat, gamma=0.0):
    """findClusterAssignments dense-centres branch; returns (assign0 int32[n], mind f64[n])."""
    jc, ir, x = _csc(jc, ir, x)
    Cmat = np.asarray(Cmat, np.float64).reshape(p, -1)
    K = Cmat.shape[1]
    a, mind = np.zeros(n, np.int32), np.zeros(n)
    lib().orc_assign(p, n, K, jc, ir, x, _colmajor(Cmat), float(gamma or 0.0), np.zeros(p * K), a, mind)
    return a, mind


def dist_sparse_centers(p, n, jc, ir, x, cjc, cir, cx, K, gamma=0.0):
    jc, ir, x = _csc(jc, ir, x)
    cjc, cir, cx = _csc(cjc, cir, cx)
    out = np.zeros(n * K)
    lib().orc_dist_sparse_centers(p, n, K, jc, ir, x, cjc, cir, cx, float(gamma or 0.0), out)
    return out.reshape(n, K).T.copy()


def accumulate(p, n, K, jc, ir, x, a):
    jc, ir, x = _csc(jc, ir, x)
    sums, counts, nk = np.zeros(p * K), np.zeros(p * K), np.zeros(K, np.int64)
    lib().orc_accumulate(p, n, K, jc, ir, x, np.ascontiguousarray(a, np.int32), sums, counts, nk)
    return sums.reshape(K, p).T.copy(), counts.reshape(K, p).T.copy(), nk


def finalize_centers(sums, counts, nk, gamma, centers):
    p, K = sums.shape
    c = _colmajor(centers).copy()
    lib().orc_finalize_centers(p, K, _colmajor(sums), _colmajor(counts), np.ascontiguousarray(nk, np.int64),
                               float(gamma), c)
    return c.reshape(K, p).T.copy()


def finalize_plain_mean(sums, nk, centers):
    """'MLcorrection',false (kmeans_sparsified.m:449-451): centers(:,k) = mean(full(X(:,ind)),2) = sums(:,k) / nk(k)."""
    p, K = sums.shape
    c = _colmajor(centers).copy()
    lib().orc_finalize_plain_mean(p, K, _colmajor(sums), np.ascontiguousarray(nk, np.int64), c)
    return c.reshape(K, p).T.copy()


_EMPTY_ACTIONS = {"singleton": 0, "drop": 1, "error": 2}


def lloyd(p, n, jc, ir, x, centers, gamma, unbiased=True, maxiter=100, tol=1e-6, empty_action="singleton",
          mlcorrection=True):
    """Dense-centre Lloyd loop (kmeans_sparsified.m:417-486) with the reference's three EmptyAction choices
    (:432-445,454-459).  Returns dict; under 'drop' ``centers`` has the surviving columns only and ``assign`` is
    None when the last iteration dropped one (the reference leaves assignments = [] then, :457).
    mlcorrection=False: the plain-mean update of :449-451 instead of :447-448."""
    jc, ir, x = _csc(jc, ir, x)
    centers = np.asarray(centers, np.float64).reshape(p, -1)
    K = centers.shape[1]
    c = _colmajor(centers).copy()
    a, mind = np.zeros(n, np.int32), np.zeros(n)
    dff, obj = np.zeros(maxiter), np.zeros(maxiter)
    Kio, dropped = C.c_size_t(K), C.c_int(0)
    fn = lib().orc_lloyd_ex if mlcorrection else lib().orc_lloyd_plain
    its = fn(p, n, C.byref(Kio), jc, ir, x, float(gamma), int(bool(unbiased)), int(maxiter),
             float(tol), _EMPTY_ACTIONS[empty_action], c, a, mind, dff, obj, C.byref(dropped))
    if its < 0:
        raise RuntimeError("One cluster lost all its members")       # kmeans_sparsified.m:439
    Kf = int(Kio.value)
    return dict(iterations=its, centers=c[: p * Kf].reshape(Kf, p).T.copy(), assign=None if dropped.value else a,
                mind=mind, dff=dff[:its], obj=obj[:its], K=Kf)


def lloyd_iter_threads(p, n, jc, ir, x, centers, gamma, threads, unbiased=True):
    """ONE Lloyd iteration, points column-partitioned over ``threads`` workers as hadamard_pthreads partitions its
    columns (hadamard_pthreads.c:121-204) -- the all-cores CPU baseline.  Returns dict(centers, assign, mind)."""
    jc, ir, x = _csc(jc, ir, x)
    centers = np.asarray(centers, np.float64).reshape(p, -1)
    K = centers.shape[1]
    c = _colmajor(centers).copy()
    a, mind = np.zeros(n, np.int32), np.zeros(n)
    lib().orc_lloyd_iter_threads(p, n, K, jc, ir, x, float(gamma), int(bool(unbiased)), c, a, mind, int(threads))
    return dict(centers=c.reshape(K, p).T.copy(), assign=a, mind=mind)


def fwht(x, threads: int = 0):
    """hadamard(x) / hadamard_pthreads(x): x is m x n (numpy [m, n]); columns transformed."""
    x = np.asarray(x, np.float64)
    if x.ndim == 1:
        x = x[:, None]
    m, n = x.shape
    rc = lib().orc_check_pow2(m)
    if rc == 1:
        raise ValueError("Vector length must be greater than 1.")
    if rc == 2:
        raise ValueError("Vector length must be power of 2.")
    xin = np.ascontiguousarray(x.T).ravel()
    out = np.zeros_like(xin)
    if threads and threads > 1:
        lib().orc_fwht_threads(m, n, xin, out, threads)
    else:
        lib().orc_fwht(m, n, xin, out)
    return out.reshape(n, m).T.copy()


def mix(x, d, p2):
    """mix(X) of kmeans_sparsified.m:295 incl. the (1+2eps) pre-scale (:292)."""
    x = np.asarray(x, np.float64)
    p, n = x.shape
    xin = np.ascontiguousarray(x.T).ravel()
    out = np.zeros(n * p2)
    lib().orc_mix(p, p2, n, xin, np.ascontiguousarray(d, np.float64), 1.0 + 2 * np.finfo(np.float64).eps,
                  float(np.sqrt(np.float64(p2))), out, np.zeros(p2))
    return out.reshape(n, p2).T.copy()


# ------------------------------------------------------------------------------------------------
# oracle/_ref: the part of the REFERENCE's own C that builds here (oracle/Makefile, ref_hadamard*_shim.c,
# ref_sparse_shim.c): private/hadamard.c:57-92, private/hadamard_pthreads.c:57-119, SparseMatrixMinusCluster.c:121-129
# and :131-183, SparseMatrixInnerProduct.c:86-100, SparseMatrixColumnNormSq.c:70-77, compiled from where the files lie
# under /root/reference.  Present only where `make -C oracle` ran with the reference on disk (and, as prebuilt
# binaries, wherever the snapshot travelled).  Used to PIN orc_fwht / orc_fwht_threads (rows a13, a14) and
# orc_dist_csc / orc_dist_csc_beta / orc_innerprod_csc / orc_colnormsq_csc (rows a1-a3, a11, a12), to write
# tests/golden/ref_*.npz, and as the "reference" legs of bench.py's cpu_baseline.
# ------------------------------------------------------------------------------------------------
_REF_DIR = os.path.join(_HERE, "_ref")
_ref_libs: dict = {}
_REF_NAMES = {"native": "libref_hadamard.so", "portable": "libref_hadamard_portable.so",
              "pthreads": "libref_hadamard_pthreads.so", "sparse": "libref_sparse.so", "sparse_O2": "libref_sparse_O2.so"}


def ref_available(flavor: str = "portable") -> bool:
    """flavor: 'native' (setup_kmeans.m:53's -march=native build: only meaningful on the machine that built it),
    'portable' (the same excerpt, -O3 without -march), 'pthreads' (hadamard_pthreads.c's worker + kernels),
    'sparse' (the three Sparse*.c loops, `-O` as setup_kmeans.m:19,26,33), 'sparse_O2' (the same, -O2 -fwrapv)."""
    return os.path.exists(os.path.join(_REF_DIR, _REF_NAMES[flavor]))


def _ref(flavor: str):
    if flavor not in _ref_libs:
        L = C.CDLL(os.path.join(_REF_DIR, _REF_NAMES[flavor]))
        if flavor.startswith("sparse"):
            L.ref_dist_csc.argtypes = [_sz, _sz, _sz, _u64p, _u64p, _f64p, _f64p, _f64p]
            L.ref_dist_csc_beta.argtypes = [_sz, _u64p, _u64p, _f64p, _f64p, C.c_double, _f64p]
            L.ref_innerprod_csc.argtypes = [_sz, _u64p, _u64p, _f64p, _f64p, _f64p, _f64p]
            L.ref_colnormsq_csc.argtypes = [_sz, _u64p, _f64p, _f64p]
        elif flavor == "pthreads":
            L.ref_hadamard_pthreads.argtypes = [C.c_uint, C.c_uint, _f64p, _f64p, C.c_uint]
        else:
            L.ref_hadamard.argtypes = [C.c_uint, C.c_uint, _f64p, _f64p]
        _ref_libs[flavor] = L
    return _ref_libs[flavor]


def ref_fwht(x, flavor: str = "portable", threads: int = 4):
    """The reference's own hadamard_apply_matrix (hadamard.c:86-92 / hadamard_pthreads.c:101-107 behind `worker`)
    on an m x n array.  No size validation: the reference does that in its gateway, which needs mex.h."""
    x = np.asarray(x, np.float64)
    if x.ndim == 1:
        x = x[:, None]
    m, n = x.shape
    assert m > 1 and (m & (m - 1)) == 0 and m * n < 2 ** 32   # (`unsigned j*m`, hadamard.c:90)
    xin = np.ascontiguousarray(x.T).ravel()
    out = np.zeros_like(xin)
    if flavor == "pthreads":
        _ref(flavor).ref_hadamard_pthreads(m, n, xin, out, int(threads))
    else:
        _ref(flavor).ref_hadamard(m, n, xin, out)
    return out.reshape(n, m).T.copy()


def ref_dist_csc(p, n, jc, ir, x, Cmat, flavor: str = "sparse"):
    """The reference's own `switch (K)` (SparseMatrixMinusCluster.c:131-183): K x n distances (numpy [K, n])."""
    jc, ir, x = _csc(jc, ir, x)
    Cmat = np.asarray(Cmat, np.float64).reshape(p, -1)
    K = Cmat.shape[1]
    out = np.zeros(n * K)
    _ref(flavor).ref_dist_csc(p, n, K, jc, ir, x, _colmajor(Cmat), out)
    return out.reshape(n, K).T.copy()


def ref_dist_csc_flat(p, n, K, jc, ir, x, c_colmajor, out, flavor: str = "sparse"):
    """Same, on prepared buffers (bench.py's cpu_baseline: nothing but the reference's loop inside the timed call)."""
    _ref(flavor).ref_dist_csc(p, n, K, jc, ir, x, c_colmajor, out)


def ref_dist_csc_beta(n, jc, ir, x, c, beta, flavor: str = "sparse"):
    """The reference's own beta loop (SparseMatrixMinusCluster.c:121-129)."""
    jc, ir, x = _csc(jc, ir, x)
    out = np.zeros(n)
    _ref(flavor).ref_dist_csc_beta(n, jc, ir, x, np.ascontiguousarray(c, np.float64).ravel(), float(beta), out)
    return out


def ref_innerprod_csc(n, jc, ir, x, c, flavor: str = "sparse"):
    """The reference's own loop of SparseMatrixInnerProduct.c:86-100: (innerProd, normX2)."""
    jc, ir, x = _csc(jc, ir, x)
    ip, nx2 = np.zeros(n), np.zeros(n)
    _ref(flavor).ref_innerprod_csc(n, jc, ir, x, np.ascontiguousarray(c, np.float64).ravel(), ip, nx2)
    return ip, nx2


def ref_colnormsq_csc(n, jc, x, flavor: str = "sparse"):
    """The reference's own loop of SparseMatrixColumnNormSq.c:70-77."""
    out = np.zeros(n)
    _ref(flavor).ref_colnormsq_csc(n, np.ascontiguousarray(jc, np.uint64), np.ascontiguousarray(x, np.float64), out)
    return out
