"""Loader for libspkm.so (the C ABI of include/spkm.h) -- fails loudly, never falls back.

libspkm.so is built without a DT_NEEDED entry for the HIP runtime: its hip* symbols bind
to whichever libamdhip64 is already in the process.  When PyTorch is importable we load
PyTorch's bundled runtime first (RTLD_GLOBAL), so that ``tensor.data_ptr()`` values and the
kernels launched here belong to one and the same runtime; otherwise /opt/rocm's.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libspkm.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "spkm.h")

OK = 0
ERR_NULL_ARG, ERR_CENTER_ROWS, ERR_BETA_K, ERR_LEN_LE_1, ERR_NOT_POW2 = -1, -2, -3, -4, -5
ERR_BAD_CSC, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_BAD_VALUE, ERR_COMM = -6, -7, -8, -9, -10
COMM_ID_BYTES = 128


class SpkmError(RuntimeError):
    """Raised for every non-zero status of the C ABI; carries the reference's message text."""

    def __init__(self, status: int, where: str = ""):
        self.status = status
        msg = lib().spkm_strerror(status).decode()
        super().__init__(f"{where + ': ' if where else ''}{msg} (spkm status {status})")


def _preload_hip_runtime() -> str:
    cands = []
    try:
        import torch  # noqa: F401  (side effect: loads torch/lib/libamdhip64.so)

        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:  # torch absent: stand-alone use against the system runtime
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    errs = []
    for c in cands:
        try:
            C.CDLL(c, mode=C.RTLD_GLOBAL)
            return c
        except OSError as e:  # try the next candidate
            errs.append(f"{c}: {e}")
    raise ImportError("sparsifiedkmeans_amd: cannot load a HIP runtime (libamdhip64): " + "; ".join(errs))


def preload_rccl() -> str | None:
    """Make the RCCL that belongs to the process's HIP runtime visible to libspkm.so's run-time binding (PyTorch's
    bundled librccl.so when torch is importable: its extension modules load it RTLD_LOCAL).  Returns the path used,
    or None to let the library search by itself ($SPKM_RCCL_PATH, librccl.so, /opt/rocm/lib/librccl.so)."""
    if os.environ.get("SPKM_RCCL_PATH"):
        return None
    try:
        import torch

        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return cand
    except Exception:
        pass
    return None


_lib = None
hip_runtime_path = None

_u64 = C.c_uint64
_vp = C.c_void_p
_dbl = C.c_double


def declared_symbols() -> list[str]:
    """Every function name include/spkm.h declares (used by the symbol-coverage test)."""
    with open(HEADER) as f:
        txt = f.read()
    return sorted(set(re.findall(r"\b(spkm_[A-Za-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib, hip_runtime_path
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            f"sparsifiedkmeans_amd: {_SO} is missing -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or sparsifiedkmeans_amd/csrc/build.sh. "
            "There is no CPU fallback.")
    hip_runtime_path = _preload_hip_runtime()
    L = C.CDLL(_SO)
    L.spkm_strerror.restype = C.c_char_p
    L.spkm_strerror.argtypes = [C.c_int]
    L.spkm_version.restype = C.c_int
    L.spkm_ctx_create.argtypes = [C.c_int, _vp, C.POINTER(_vp)]
    L.spkm_ctx_destroy.argtypes = [_vp]
    L.spkm_ctx_destroy.restype = None
    L.spkm_ctx_sync.argtypes = [_vp]
    L.spkm_device_info.argtypes = [_vp, C.POINTER(C.c_int64)]
    L.spkm_SparseMatrixMinusCluster_host.argtypes = [_vp, _u64, _u64, _vp, _vp, _vp, _u64, _u64, _vp, _vp, _vp]
    L.spkm_SparseMatrixInnerProduct_host.argtypes = [_vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp]
    L.spkm_SparseMatrixColumnNormSq_host.argtypes = [_vp, _u64, _vp, _vp, _vp]
    L.spkm_hadamard_host.argtypes = [_vp, _u64, _u64, _vp, _vp]
    L.spkm_hadamard_pthreads_host.argtypes = [_vp, _u64, _u64, _vp, _vp]
    L.spkm_shard_create_host.argtypes = [_vp, _u64, _u64, _vp, _vp, _vp, C.POINTER(_vp)]
    L.spkm_shard_create_dev.argtypes = [_vp, _u64, _u64, _u64, _vp, _vp, C.c_int, _vp, _u64, C.POINTER(_vp)]
    L.spkm_shard_destroy.argtypes = [_vp]
    L.spkm_shard_destroy.restype = None
    L.spkm_shard_info.argtypes = [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(C.c_int)]
    L.spkm_shard_order_info.argtypes = [_vp, C.POINTER(C.c_int64)]
    L.spkm_reduce_len.argtypes = [_u64, _u64]
    L.spkm_reduce_len.restype = _u64
    L.spkm_assign_dev.argtypes = [_vp, _vp, _u64, _vp, _dbl, _vp, _vp, _vp, _vp]
    L.spkm_assign_sparse_centers_dev.argtypes = [_vp, _vp, _u64, _vp, _vp, _dbl, _vp, _vp, _vp, _vp]
    L.spkm_accumulate_dev.argtypes = [_vp, _vp, _u64, _vp, _vp]
    L.spkm_assign_accumulate_dev.argtypes = [_vp, _vp, _u64, _vp, _dbl, _vp, _vp, _vp, _vp, _vp]
    L.spkm_last_path_info.argtypes = [_vp, C.POINTER(C.c_int64)]
    L.spkm_finalize_dev.argtypes = [_vp, _u64, _u64, _vp, _dbl, _vp, _vp]
    L.spkm_fwht_dev.argtypes = [_vp, _u64, _u64, _vp, _vp]
    L.spkm_mix_dev.argtypes = [_vp, _u64, _u64, _u64, _vp, _vp, _dbl, _dbl, _vp]
    L.spkm_mix_sample_dev.argtypes = [_vp, _u64, _u64, _u64, _vp, _vp, _dbl, _dbl, _u64, _u64, _u64, _vp, C.c_int, _vp]
    L.spkm_record_bytes.argtypes = [_u64, C.c_int]
    L.spkm_record_bytes.restype = _u64
    L.spkm_mix_sample_rec_dev.argtypes = [_vp, _u64, _u64, _u64, _vp, _vp, _dbl, _dbl, _u64, _u64, _u64, C.c_int, _vp]
    L.spkm_shard_create_rec_dev.argtypes = [_vp, _u64, _u64, _u64, C.c_int, _vp, C.POINTER(_vp)]
    L.spkm_last_screen_rounds.argtypes = [_vp, C.POINTER(C.c_int64)]
    L.spkm_last_screen_mode.argtypes = [_vp, C.POINTER(C.c_int64)]
    L.spkm_last_events_form.argtypes = [_vp, C.POINTER(C.c_int64)]
    L.spkm_shard_reset_policy.argtypes = [_vp]
    L.spkm_dense_assign_dev.argtypes = [_vp, _u64, _u64, _vp, _u64, _vp, _vp, _vp]
    L.spkm_dense_accumulate_dev.argtypes = [_vp, _u64, _u64, _vp, _u64, _vp, _vp, _vp]
    L.spkm_last_assign_kernel_ms.argtypes = [_vp, C.POINTER(_dbl)]
    L.spkm_timing_log.argtypes = [_vp, C.c_int]
    L.spkm_timing_read.argtypes = [_vp, C.POINTER(_dbl), C.c_int, C.POINTER(C.c_int)]
    L.spkm_debug_block_times.argtypes = [_vp, C.c_int, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]
    L.spkm_debug_shard_bounds.argtypes = [_vp, _vp, _vp, _vp, _vp]
    L.spkm_exact_pass_points.argtypes = [_vp, C.POINTER(C.c_int64)]
    L.spkm_screen_work_totals.argtypes = [_vp, C.POINTER(C.c_int64)]
    L.spkm_distances_dev.argtypes = [_vp, _vp, _u64, _vp, _dbl, _vp, _vp]
    L.spkm_distances_stats_dev.argtypes = [_vp, _vp, _u64, _vp, _dbl, _vp, _vp, _vp]
    L.spkm_shard_set_lazy_stats.argtypes = [_vp, C.c_int]
    L.spkm_shard_release_csc.argtypes = [_vp, _vp]
    L.spkm_shard_get_column_host.argtypes = [_vp, _vp, _u64, _u64, _vp, _vp, C.POINTER(C.c_uint64)]
    L.spkm_kpp_update_dev.argtypes = [_vp, _u64, _vp, _vp, C.c_int, _vp, C.POINTER(C.c_double)]
    L.spkm_kpp_draw_dev.argtypes = [_vp, _u64, _vp, C.c_double, C.POINTER(C.c_int64)]
    L.spkm_ctx_reload_switches.argtypes = [_vp]
    L.spkm_widen_f64_dev.argtypes = [_vp, C.c_int, _u64, _vp, _vp]
    L.spkm_comm_unique_id.argtypes = [_vp]
    L.spkm_comm_init.argtypes = [_vp, C.c_int, C.c_int, _vp]
    L.spkm_comm_destroy.argtypes = [_vp]
    L.spkm_comm_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.spkm_allreduce_f64_dev.argtypes = [_vp, _vp, _u64]
    L.spkm_lloyd_iter.argtypes = [_vp, _vp, _u64, _vp, _dbl, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
    L.spkm_lloyd_iter_host.argtypes = [_vp, _vp, _u64, _vp, _dbl, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.spkm_ctx_last_error.argtypes = [_vp]
    L.spkm_ctx_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(status: int, where: str = ""):
    if status != 0:
        raise SpkmError(status, where)
