"""Device-resident Lloyd engine: thin Python plumbing over Part 2 of include/spkm.h.

PyTorch is used for what it is good at here -- device allocations, the current HIP stream and
``torch.distributed`` (RCCL) -- and nothing else: every kernel is ours (libspkm.so).

One iteration (kmeans_sparsified.m:417-486 with dense centres):
    assign      findClusterAssignments(X, centers, [], gamma)         :420
    accumulate  per-cluster sums / counts of the local shard           :430-431,447-448
    all-reduce  ONE RCCL SUM over [sums | counts | nk | obj2]          (new: data-parallel over points)
    finalize    centers = gamma*S./(Cnt+1e-16); dff; obj               :448,470-471
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import Context


def _p(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


class Shard:
    """A block of points (columns of a p x n CSC matrix) resident in HBM."""

    def __init__(self, ctx: Context, handle, keep=()):
        self.ctx = ctx
        self.handle = handle
        self._keep = keep  # tensors backing an adopted shard
        p, n, nnz, bits = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        _lib.check(_lib.lib().spkm_shard_info(handle, C.byref(p), C.byref(n), C.byref(nnz), C.byref(bits)))
        self.p, self.n, self.nnz, self.ir_bits = p.value, n.value, nnz.value, bits.value

    @classmethod
    def from_scipy(cls, ctx: Context, X) -> "Shard":
        from .ops import _csc_parts, _ptr

        p, n, jc, ir, x = _csc_parts(X)
        h = C.c_void_p()
        _lib.check(_lib.lib().spkm_shard_create_host(ctx.handle, p, n, _ptr(jc), _ptr(ir), _ptr(x), C.byref(h)),
                   "spkm_shard_create_host")
        return cls(ctx, h)

    @classmethod
    def from_device(cls, ctx: Context, p: int, jc: torch.Tensor, ir: torch.Tensor, x: torch.Tensor,
                    nnz: int | None = None) -> "Shard":
        """Adopt device tensors: jc int64[n+1], ir int16/uint16 or int32, x float64.  ``nnz`` is the
        number of stored entries; ir / x may be longer (>= nnz + 16 unlocks the fixed-stride exact kernel, >= nnz + 48 the screen)."""
        assert jc.dtype == torch.int64 and x.dtype == torch.float64 and jc.is_cuda and x.is_cuda and ir.is_cuda
        bits = ir.element_size() * 8
        n = jc.numel() - 1
        nnz = x.numel() if nnz is None else int(nnz)
        cap = min(x.numel(), ir.numel())
        h = C.c_void_p()
        _lib.check(_lib.lib().spkm_shard_create_dev(ctx.handle, p, n, nnz, _p(jc), _p(ir), bits, _p(x), cap,
                                                    C.byref(h)), "spkm_shard_create_dev")
        return cls(ctx, h, keep=(jc, ir, x))

    @classmethod
    def from_records(cls, ctx: Context, p: int, n: int, s: int, rec: torch.Tensor, ir_bits: int = 16) -> "Shard":
        """Adopt a device buffer of n records (``mix_sample_records_device``'s output: a point's s float64 values, then its
        s row ids, in ``record_bytes(s, ir_bits)`` bytes) -- the layout the fused call reads; the separate CSC arrays never
        exist (spkm_shard_create_rec_dev)."""
        assert rec.is_cuda and rec.dtype == torch.uint8 and rec.is_contiguous()
        # (n records + 256 bytes: the record kernels fetch whole 16-byte pieces and a wave's batch of records ahead of its
        #  bounds check -- spkm.h, spkm_shard_create_rec_dev)
        assert rec.numel() >= n * record_bytes(s, ir_bits) + 256, "the record buffer needs 256 bytes of slack behind its last record"
        h = C.c_void_p()
        _lib.check(_lib.lib().spkm_shard_create_rec_dev(ctx.handle, p, n, s, ir_bits, _p(rec), C.byref(h)),
                   "spkm_shard_create_rec_dev")
        return cls(ctx, h, keep=(rec,))

    def set_lazy_stats(self, on: bool = True):
        """Allow fused calls without distances to leave obj2 / the largest distance unevaluated (NaN) and to move the
        per-cluster sums by the points that changed cluster (spkm_shard_set_lazy_stats); LloydEngine.distances() then
        delivers them for the iteration that needs them."""
        _lib.check(_lib.lib().spkm_shard_set_lazy_stats(self.handle, 1 if on else 0), "spkm_shard_set_lazy_stats")

    def column(self, i: int) -> tuple[np.ndarray, np.ndarray]:
        """(row ids int64 ascending, values float64) of column ``i`` -- from the CSC arrays or, once they are released,
        from the record layout (spkm_shard_get_column_host)."""
        cap = 64
        while True:
            ir = np.zeros(cap, np.uint64)
            x = np.zeros(cap, np.float64)
            cnt = C.c_uint64()
            st = _lib.lib().spkm_shard_get_column_host(self.ctx.handle, self.handle, int(i), cap, C.c_void_p(ir.ctypes.data),
                                                       C.c_void_p(x.ctypes.data), C.byref(cnt))
            if st == _lib.ERR_BAD_VALUE and cnt.value > cap:
                cap = int(cnt.value)
                continue
            _lib.check(st, "spkm_shard_get_column_host")
            return ir[: cnt.value].astype(np.int64), x[: cnt.value].copy()

    def release_csc(self) -> bool:
        """Let go of the CSC value / row-id arrays once the record layout and the screen copy exist
        (spkm_shard_release_csc): the tensors an adopted shard was built from are dropped here, so that their memory
        returns to the allocator unless the caller still holds them.  Returns False when the shard does not qualify
        (ragged, columns longer than 64, no room for the records) -- nothing changes then."""
        st = _lib.lib().spkm_shard_release_csc(self.ctx.handle, self.handle)
        if st == _lib.ERR_UNSUPPORTED:
            return False
        _lib.check(st, "spkm_shard_release_csc")
        if self._keep:
            self._keep = (self._keep[0],)      # jc stays referenced by the library; ir / x do not
        return True

    def debug_bounds(self) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(ub, lb, lib_assign) the shard carries from its last screen call (spkm_debug_shard_bounds; test aid)."""
        ub, lb, a = np.zeros(self.n, np.float32), np.zeros(self.n), np.zeros(self.n, np.int32)
        _lib.check(_lib.lib().spkm_debug_shard_bounds(self.ctx.handle, self.handle, ub.ctypes.data, lb.ctypes.data,
                                                      a.ctypes.data), "spkm_debug_shard_bounds")
        return ub, lb, a

    def order_info(self) -> tuple[bool, bool]:
        """(the library keeps this shard's points in an order of its own, it regrouped them since the last reset_policy) --
        spkm_shard_order_info: data in arbitrary order is regrouped by cluster inside the library; nothing the caller sees moves."""
        a = (C.c_int64 * 2)()
        _lib.check(_lib.lib().spkm_shard_order_info(self.handle, a), "spkm_shard_order_info")
        return bool(a[0]), bool(a[1])

    def reset_policy(self):
        """New start / new replicate: drop the adaptive state of the fused call (spkm_shard_reset_policy)."""
        _lib.check(_lib.lib().spkm_shard_reset_policy(self.handle), "spkm_shard_reset_policy")

    def close(self):
        if self.handle:
            _lib.lib().spkm_shard_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def torch_context(device: int | None = None) -> Context:
    """A Context bound to torch's current HIP stream on ``device``."""
    if not torch.cuda.is_available():
        raise RuntimeError("sparsifiedkmeans_amd: no HIP device visible to PyTorch; there is no CPU fallback")
    device = torch.cuda.current_device() if device is None else device
    torch.cuda.set_device(device)
    return Context(device, torch.cuda.current_stream(device).cuda_stream)


def comm_size(ctx: Context) -> int:
    """ranks of the RCCL communicator attached to ``ctx`` inside libspkm.so (0 = none)."""
    n, r = C.c_int(), C.c_int()
    _lib.check(_lib.lib().spkm_comm_info(ctx.handle, C.byref(n), C.byref(r)))
    return n.value


def comm_info(ctx: Context) -> tuple[int, int]:
    """(ranks, this rank) of the RCCL communicator attached to ``ctx`` inside libspkm.so ((0, 0) = none): spkm_comm_info."""
    n, r = C.c_int(), C.c_int()
    _lib.check(_lib.lib().spkm_comm_info(ctx.handle, C.byref(n), C.byref(r)))
    return n.value, r.value


def attach_rccl(ctx: Context, group=None) -> int:
    """Attach ``ctx`` to an RCCL communicator owned by libspkm.so, spanning the ranks of ``group`` (default: the
    world of torch.distributed, which only carries the 128-byte rendezvous token here -- any backend).  From then on
    LloydEngine.iterate() is ONE library call per iteration, the all-reduce issued by the library on the context's
    stream (spkm_lloyd_iter).  Collective: every rank calls it.  Returns the communicator size."""
    import torch.distributed as dist

    if comm_size(ctx):
        return comm_size(ctx)
    if not (dist.is_available() and dist.is_initialized()):
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)

    def agree(ok: bool, what: str, why: str = ""):
        """every rank learns whether ALL ranks got through a step -- a rank that cannot bind RCCL must not leave the
        others waiting inside ncclCommInitRank"""
        if world > 1:
            votes = [None] * world
            dist.all_gather_object(votes, (bool(ok), why), group=group)
        else:
            votes = [(bool(ok), why)]
        bad = [(r, w) for r, (o, w) in enumerate(votes) if not o]
        if bad:
            raise _lib.SpkmError(_lib.ERR_COMM, f"{what} failed on rank(s) " + "; ".join(f"{r}: {w}" for r, w in bad))

    # step 1 (local): RCCL can be loaded and bound here -- every rank asks for a token, only rank 0's is used
    ident = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    why = ""
    try:
        _lib.preload_rccl()
        st = _lib.lib().spkm_comm_unique_id(ident)
        if st != 0:
            why = _lib.lib().spkm_strerror(st).decode()
    except Exception as e:                                      # noqa: BLE001 -- reported to every rank below
        st, why = -1, repr(e)
    agree(st == 0, "binding RCCL (spkm_comm_unique_id)", why)
    # step 2 (collective): the communicator
    if world > 1:
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(box[0])
    st = _lib.lib().spkm_comm_init(ctx.handle, world, rank, ident)
    why = "" if st == 0 else _lib.lib().spkm_ctx_last_error(ctx.handle).decode()
    try:
        agree(st == 0, "spkm_comm_init", why)
    except _lib.SpkmError:
        if st == 0:
            _lib.lib().spkm_comm_destroy(ctx.handle)
        raise
    return world


def detach_rccl(ctx: Context) -> None:
    _lib.check(_lib.lib().spkm_comm_destroy(ctx.handle), "spkm_comm_destroy")


class LloydEngine:
    """State of one Lloyd run over one local shard (one process per GPU).

    centers: torch float64 [K, p] (row k = centre k, i.e. MATLAB's p x K in column-major bytes).
    ``group`` is a torch.distributed process group (or None = use the default group when
    torch.distributed is initialised, single-GPU otherwise).
    """

    def __init__(self, shard: Shard, K: int, gamma: float, unbiased: bool = True, group=None):
        self.shard, self.ctx, self.K, self.p = shard, shard.ctx, int(K), int(shard.p)
        self.gamma = float(gamma)
        self.unbiased = bool(unbiased)
        dev = torch.device("cuda", self.ctx.device)
        n = shard.n
        self.assign = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
        self.mind = torch.empty(max(n, 1), dtype=torch.float64, device=dev)[:n]
        self.stats = torch.zeros(3, dtype=torch.float64, device=dev)
        self.nk = torch.zeros(self.K, dtype=torch.int64, device=dev)
        self.reduce = torch.zeros(int(_lib.lib().spkm_reduce_len(self.p, self.K)), dtype=torch.float64, device=dev)
        self.out = torch.zeros(2, dtype=torch.float64, device=dev)
        self._host_res = None      # iterate_host's [dff^2, obj^2, nk] on the host
        self.group = group
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized()

    # -- steps ---------------------------------------------------------------------------
    def assign_step(self, centers: torch.Tensor):
        assert centers.dtype == torch.float64 and centers.is_contiguous() and tuple(centers.shape) == (self.K, self.p)
        g = self.gamma if self.unbiased else 0.0
        _lib.check(_lib.lib().spkm_assign_dev(self.ctx.handle, self.shard.handle, self.K, _p(centers), g,
                                              _p(self.assign), _p(self.mind), _p(self.stats), _p(self.nk)),
                   "spkm_assign_dev")

    def assign_sparse_step(self, centers: torch.Tensor, mask: torch.Tensor):
        """Sparse-centres branch (findClusterAssignments.m:63-75): ``centers`` [K, p] dense values,
        ``mask`` [K, p] uint8 support of each centre."""
        assert centers.dtype == torch.float64 and centers.is_contiguous() and tuple(centers.shape) == (self.K, self.p)
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and tuple(mask.shape) == (self.K, self.p)
        g = self.gamma if self.unbiased else 0.0
        _lib.check(_lib.lib().spkm_assign_sparse_centers_dev(self.ctx.handle, self.shard.handle, self.K, _p(centers),
                                                             _p(mask), g, _p(self.assign), _p(self.mind),
                                                             _p(self.stats), _p(self.nk)),
                   "spkm_assign_sparse_centers_dev")

    def accumulate_step(self):
        _lib.check(_lib.lib().spkm_accumulate_dev(self.ctx.handle, self.shard.handle, self.K, _p(self.assign),
                                                  _p(self.reduce)), "spkm_accumulate_dev")

    def allreduce_step(self):
        """the one exchange of an iteration: through the library's own RCCL communicator when one is attached to the
        context (attach_rccl), else torch.distributed (RCCL under backend "nccl", gloo in the CPU tests)"""
        if comm_size(self.ctx) > 0:
            _lib.check(_lib.lib().spkm_allreduce_f64_dev(self.ctx.handle, _p(self.reduce), self.reduce.numel()),
                       "spkm_allreduce_f64_dev")
            return
        from .distributed import allreduce_

        allreduce_(self.reduce, self.group)

    def finalize_step(self, centers: torch.Tensor):
        _lib.check(_lib.lib().spkm_finalize_dev(self.ctx.handle, self.p, self.K, _p(self.reduce), self.gamma,
                                                _p(centers), _p(self.out)), "spkm_finalize_dev")

    def assign_accumulate_step(self, centers: torch.Tensor, want_mind: bool = True):
        """assign_step + accumulate_step in one call (same assignments, counts and distances bit for bit, sums to the order of
        summation -- with lazy statistics they are moved by the points that changed cluster; lets the library take its
        certified-screen fast path when the shard qualifies).  want_mind=False: the per-point distances of this call
        are not written (self.mind keeps whatever it held); ``distances(centers_used)`` produces them on demand."""
        assert centers.dtype == torch.float64 and centers.is_contiguous() and tuple(centers.shape) == (self.K, self.p)
        g = self.gamma if self.unbiased else 0.0
        _lib.check(_lib.lib().spkm_assign_accumulate_dev(self.ctx.handle, self.shard.handle, self.K, _p(centers), g,
                                                         _p(self.assign), _p(self.mind) if want_mind else None,
                                                         _p(self.stats), _p(self.nk), _p(self.reduce)),
                   "spkm_assign_accumulate_dev")

    def distances(self, centers_used: torch.Tensor) -> torch.Tensor:
        """self.mind <- the reference's distance of every point to centroid self.assign[i] under ``centers_used``
        (the centres the last assignment was computed with, i.e. BEFORE their update), and self.stats <- [sum of their
        squares (LOCAL obj2), the largest, its first index] -- spkm_distances_stats_dev."""
        assert centers_used.dtype == torch.float64 and centers_used.is_contiguous() and tuple(centers_used.shape) == (self.K, self.p)
        g = self.gamma if self.unbiased else 0.0
        _lib.check(_lib.lib().spkm_distances_stats_dev(self.ctx.handle, self.shard.handle, self.K, _p(centers_used), g,
                                                       _p(self.assign), _p(self.mind), _p(self.stats)),
                   "spkm_distances_stats_dev")
        return self.mind

    def last_path_info(self) -> tuple[int, int]:
        a = (C.c_int64 * 2)()
        _lib.check(_lib.lib().spkm_last_path_info(self.ctx.handle, a))
        return int(a[0]), int(a[1])

    def exact_pass_points(self) -> tuple[int, int]:
        """(running total of the points the exact pass streamed on this context, points it streamed in the last call):
        fewer than n once clusters are settled (spkm_exact_pass_points)."""
        a = (C.c_int64 * 2)()
        _lib.check(_lib.lib().spkm_exact_pass_points(self.ctx.handle, a))
        return int(a[0]), int(a[1])

    def screen_work_totals(self) -> tuple[int, int]:
        """(rounds the screen launches on this context executed for all centroids of a tile, rounds of launches doing all the
        work) -- running totals (spkm_screen_work_totals); a round = 4 stored entries of a 16-point step against one tile."""
        a = (C.c_int64 * 2)()
        _lib.check(_lib.lib().spkm_screen_work_totals(self.ctx.handle, a))
        return int(a[0]), int(a[1])

    def last_screen_rounds(self) -> tuple[int, int]:
        """(rounds evaluated for all centroids, rounds per column) of the last screen call; the first is smaller
        when the two-phase screen was used (spkm_last_screen_rounds)."""
        a = (C.c_int64 * 2)()
        _lib.check(_lib.lib().spkm_last_screen_rounds(self.ctx.handle, a))
        return int(a[0]), int(a[1])

    def last_screen_mode(self) -> tuple[int, ...]:
        """(form of the last screen call: 0 plain / 1 two-phase / 2 hinted / -1 none, then its counters:
        listed points, ambiguous points, early-finished (step, tile) pairs, steps skipped on the carried bounds,
        the running total of skipped steps on this context, how the sums were formed (0 full pass with distances, 3 full
        pass without, 2 events, 4 events applied one by one), 2 if the bounds list named points; blocks) -- spkm_last_screen_mode."""
        a = (C.c_int64 * 8)()
        _lib.check(_lib.lib().spkm_last_screen_mode(self.ctx.handle, a))
        return tuple(int(v) for v in a)

    def last_events_form(self) -> tuple[int, int]:
        """(0 not incremental / 1 events sorted by cluster / 2 applied one by one, 1 if pair events: one per mover, its
        record read once) of the last fused call -- spkm_last_events_form."""
        a = (C.c_int64 * 2)()
        _lib.check(_lib.lib().spkm_last_events_form(self.ctx.handle, a))
        return int(a[0]), int(a[1])

    def iterate(self, centers: torch.Tensor, want_mind: bool = True):
        """One full Lloyd iteration in place on ``centers``; returns the device tensor
        [dff^2, obj^2] (no host sync).  One library call (spkm_lloyd_iter: fused assignment + accumulation, the
        all-reduce over the library's RCCL communicator if one is attached, finalisation) unless the exchange has to
        go through torch.distributed (a process group without attach_rccl)."""
        from .distributed import is_distributed

        if is_distributed() and comm_size(self.ctx) == 0:
            self.assign_accumulate_step(centers, want_mind)
            self.allreduce_step()
            self.finalize_step(centers)
            return self.out
        assert centers.dtype == torch.float64 and centers.is_contiguous() and tuple(centers.shape) == (self.K, self.p)
        _lib.check(_lib.lib().spkm_lloyd_iter(self.ctx.handle, self.shard.handle, self.K, _p(centers), self.gamma,
                                              1 if self.unbiased else 0, _p(self.assign),
                                              _p(self.mind) if want_mind else None, _p(self.stats),
                                              _p(self.nk), _p(self.reduce), _p(self.out)), "spkm_lloyd_iter")
        return self.out

    def iterate_host(self, centers: torch.Tensor, want_mind: bool = True) -> np.ndarray:
        """One full Lloyd iteration in place on ``centers`` for a host that decides after each of them (the reference's
        driver: kmeans_sparsified.m:432, 470-487): returns the host array [dff^2, obj^2, nk[0..K-1]] (global values) --
        spkm_lloyd_iter_host: the results arrive through pinned host memory the device maps, no copy, no stream
        synchronisation.  Falls back to iterate() + one device-to-host read when the exchange has to go through
        torch.distributed (a process group without attach_rccl)."""
        from .distributed import is_distributed

        if is_distributed() and comm_size(self.ctx) == 0:
            self.iterate(centers, want_mind)
            pk = self.p * self.K
            return torch.cat([self.out, self.reduce[2 * pk: 2 * pk + self.K]]).cpu().numpy()
        assert centers.dtype == torch.float64 and centers.is_contiguous() and tuple(centers.shape) == (self.K, self.p)
        if self._host_res is None or self._host_res.size != 2 + self.K:
            self._host_res = np.zeros(2 + self.K, np.float64)
        _lib.check(_lib.lib().spkm_lloyd_iter_host(self.ctx.handle, self.shard.handle, self.K, _p(centers), self.gamma,
                                                   1 if self.unbiased else 0, _p(self.assign),
                                                   _p(self.mind) if want_mind else None, _p(self.stats),
                                                   _p(self.nk), _p(self.reduce), _p(self.out),
                                                   self._host_res.ctypes.data), "spkm_lloyd_iter_host")
        return self._host_res.copy()

    # -- helpers -------------------------------------------------------------------------
    def global_nk(self) -> torch.Tensor:
        pk = self.p * self.K
        return self.reduce[2 * pk: 2 * pk + self.K]

    def last_assign_kernel_ms(self) -> float:
        ms = C.c_double()
        _lib.check(_lib.lib().spkm_last_assign_kernel_ms(self.ctx.handle, C.byref(ms)))
        return ms.value


def fwht_device(ctx: Context, x: torch.Tensor) -> torch.Tensor:
    """hadamard() on a device tensor laid out [n, m] (each row = one column of the m x n matrix)."""
    assert x.dtype == torch.float64 and x.is_contiguous() and x.dim() == 2
    y = torch.empty_like(x)
    _lib.check(_lib.lib().spkm_fwht_dev(ctx.handle, x.shape[1], x.shape[0], _p(x), _p(y)), "spkm_fwht_dev")
    return y


def mix_device(ctx: Context, x: torch.Tensor, p2: int, sign: torch.Tensor | None, premul: float,
               postdiv: float) -> torch.Tensor:
    """mix(X) (kmeans_sparsified.m:295) on a device tensor [n, p] -> [n, p2]."""
    assert x.dtype == torch.float64 and x.is_contiguous() and x.dim() == 2
    n, p = x.shape
    y = torch.empty((n, p2), dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib().spkm_mix_dev(ctx.handle, p, p2, n, _p(x), _p(sign) if sign is not None else None,
                                       float(premul), float(postdiv), _p(y)), "spkm_mix_dev")
    return y


def mix_sample_device(ctx: Context, x: torch.Tensor, p2: int, sign: torch.Tensor | None, premul: float,
                      postdiv: float, s: int, seed: int, col0: int, ir_out: torch.Tensor, x_out: torch.Tensor):
    """Fused mix + sparsify of a dense device chunk [n, p] (kmeans_sparsified.m:316-334): writes the s sampled
    row ids of every point into ``ir_out`` (int16 viewed as uint16, or int32) and the values into ``x_out``
    (both flat, n*s entries).  The sample of a point depends only on (seed, col0 + index)."""
    assert x.dtype == torch.float64 and x.is_contiguous() and x.dim() == 2
    n, p = x.shape
    assert ir_out.numel() >= n * s and x_out.numel() >= n * s and x_out.dtype == torch.float64
    bits = ir_out.element_size() * 8
    _lib.check(_lib.lib().spkm_mix_sample_dev(ctx.handle, p, p2, n, _p(x), _p(sign) if sign is not None else None,
                                              float(premul), float(postdiv), int(s), int(seed) & (2**64 - 1),
                                              int(col0), _p(ir_out), bits, _p(x_out)), "spkm_mix_sample_dev")


def record_bytes(s: int, ir_bits: int = 16) -> int:
    """bytes of one record of s entries (spkm_record_bytes): s float64 values + s row ids, rounded up to 16"""
    return int(_lib.lib().spkm_record_bytes(int(s), int(ir_bits)))


def mix_sample_records_device(ctx: Context, x: torch.Tensor, p2: int, sign: torch.Tensor | None, premul: float,
                              postdiv: float, s: int, seed: int, col0: int, rec_out: torch.Tensor, ir_bits: int = 16):
    """mix_sample_device writing RECORDS: point i of the chunk goes to ``rec_out`` (uint8, at byte i * record_bytes(s)):
    the layout the fused call reads (spkm_mix_sample_rec_dev).  Same samples, same values as the CSC form."""
    assert x.dtype == torch.float64 and x.is_contiguous() and x.dim() == 2
    n, p = x.shape
    assert rec_out.dtype == torch.uint8 and rec_out.is_contiguous() and rec_out.numel() >= n * record_bytes(s, ir_bits)
    _lib.check(_lib.lib().spkm_mix_sample_rec_dev(ctx.handle, p, p2, n, _p(x), _p(sign) if sign is not None else None,
                                                  float(premul), float(postdiv), int(s), int(seed) & (2**64 - 1),
                                                  int(col0), int(ir_bits), _p(rec_out)), "spkm_mix_sample_rec_dev")


def dense_assign_device(ctx: Context, x: torch.Tensor, centers: torch.Tensor):
    """[assignments, distances] = findClusterAssignments(full(X), centers), dense branch / expanded quadratic
    (private/findClusterAssignments.m:157-171) for a dense device chunk ``x`` [n, p] and ``centers`` [K, p].
    Returns (assign int32 0-based [n], dist float64 [n])."""
    assert x.dtype == torch.float64 and x.is_contiguous() and x.dim() == 2
    assert centers.dtype == torch.float64 and centers.is_contiguous() and centers.shape[1] == x.shape[1]
    n, p = x.shape
    a = torch.empty(n, dtype=torch.int32, device=x.device)
    d = torch.empty(n, dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib().spkm_dense_assign_dev(ctx.handle, p, n, _p(x), centers.shape[0], _p(centers), _p(a), _p(d)),
               "spkm_dense_assign_dev")
    return a, d


def dense_accumulate_device(ctx: Context, x: torch.Tensor, assign: torch.Tensor, sums: torch.Tensor,
                            counts: torch.Tensor):
    """sums[k] += sum of the rows of ``x`` [n, p] with assign == k, counts[k] += their number: the numerators
    and denominators of mean(full(XFull(:,ind)),2) (kmeans_sparsified.m:545-550), chunk by chunk."""
    assert x.dtype == torch.float64 and x.is_contiguous() and x.dim() == 2
    assert assign.dtype == torch.int32 and assign.is_contiguous() and assign.numel() == x.shape[0]
    assert sums.dtype == torch.float64 and sums.is_contiguous() and sums.shape[1] == x.shape[1]
    assert counts.dtype == torch.float64 and counts.numel() == sums.shape[0]
    n, p = x.shape
    _lib.check(_lib.lib().spkm_dense_accumulate_dev(ctx.handle, p, n, _p(x), sums.shape[0], _p(assign), _p(sums),
                                                    _p(counts)), "spkm_dense_accumulate_dev")


_WIDEN_KIND = {torch.float32: 1, torch.uint8: 2, torch.int16: 3, torch.int32: 4}
def _copy_threads() -> int:
    """threads of the host-side staging copy (SPKM_COPY_THREADS overrides).  Measured on the benchmark box (256 hardware
    threads, tools/ingest_probe2.py): torch's own parallel copy moves 93 GB/s into pinned memory with 16 threads, 21 GB/s
    with its default of 128 (oversubscribed), and a Python thread pool of sliced copies 14 GB/s whatever its size (every
    slice's copy_ fans out over all intra-op threads again)."""
    import os

    if os.environ.get("SPKM_COPY_THREADS"):
        return max(1, int(os.environ["SPKM_COPY_THREADS"]))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 8
    return max(1, min(16, cores))


def _parallel_host_copy(dst: torch.Tensor, src: torch.Tensor, threads: int | None = None) -> None:
    """dst.copy_(src) for large host tensors with a bounded number of intra-op threads (see _copy_threads); PCIe takes
    57 GB/s out of the pinned buffer afterwards, so the staging copy must not be the slower of the two."""
    if dst.numel() * dst.element_size() < (8 << 20):
        dst.copy_(src)
        return
    want = threads if threads is not None else _copy_threads()
    before = torch.get_num_threads()
    if before != want:
        torch.set_num_threads(want)
    try:
        dst.copy_(src)
    finally:
        if before != want:
            torch.set_num_threads(before)


class StreamingSparsifier:
    """One-pass ingest of a dense dataset that never fits in HBM at once: chunk -> X*(1+2eps) -> mix ->
    sample -> append to the resident sparse shard (private/sampleAndMixFromLargeFile.m:79-129).  Only the
    sparse form (10 B per kept entry) stays on the device; the dense intermediate of a chunk lives in one
    reusable buffer and the mixed chunk never leaves LDS.

    Chunks may arrive as float64 / float32 / uint8 / int16 / int32 (a 1e9-point dataset is not stored as doubles);
    narrower types cross PCIe as they are and are widened on the device (spkm_widen_f64_dev, exact).  Host chunks go
    through PINNED memory and a copy stream with two device staging buffers: the transfer of chunk c+1 overlaps the
    transform + sampling of chunk c.  A chunk that already is a pinned torch tensor is sent from where it lies; numpy
    arrays and pageable tensors are first copied into one of two pinned staging buffers (that copy is the "read").

    ``first`` is the global index of this rank's first point (the sample of a point depends on
    (seed, global index) only, so any chunking / sharding yields the same dataset)."""

    def __init__(self, ctx: Context, p: int, n_local: int, s: int, seed: int, sign: torch.Tensor | None,
                 first: int = 0, sketch: bool = True, layout: str = "csc"):
        self.ctx, self.p, self.n, self.s, self.seed, self.first = ctx, int(p), int(n_local), int(s), int(seed), int(first)
        self.p2 = (1 << max(1, int(np.ceil(np.log2(p))))) if sketch else int(p)
        if not sketch:
            raise NotImplementedError("the fused sampler sits behind the Hadamard sketch (power-of-two row count)")
        dev = torch.device("cuda", ctx.device)
        self.sign = sign
        # layout = "records": the chunks are appended in the library's record layout (Shard.from_records; columns of at most
        # 64 entries) -- the resident shard then holds the entries once, from the start; "csc": the reference's format
        self.records = layout == "records" and self.s <= 64
        self.ir_bits = 16 if self.p2 <= 65536 else 32
        if self.records:
            self.R = record_bytes(self.s, self.ir_bits)
            self.rec = torch.empty(self.n * self.R + 256, dtype=torch.uint8, device=dev)
            self.ir = self.x = None
        else:
            self.ir = torch.zeros(self.n * self.s + 48, dtype=torch.int16 if self.p2 <= 65536 else torch.int32, device=dev)
            self.x = torch.zeros(self.n * self.s + 48, dtype=torch.float64, device=dev)
        self._dev = dev
        self.filled = 0
        self._buf = None                       # float64 chunk on the device (input of the transform)
        self._stage = [None, None]             # device staging buffers in the source's dtype
        self._pin = [None, None]               # pinned host staging buffers (for pageable sources)
        self._ev_copied = [torch.cuda.Event(), torch.cuda.Event()]
        self._ev_free = [None, None]           # recorded on the main stream when a staging buffer has been consumed
        self._ev_pin_free = [None, None]       # recorded on the copy stream when a pinned buffer has been sent
        self._turn = 0
        self._copy_stream = torch.cuda.Stream(device=dev)
        self.bytes_in = 0                      # bytes that crossed PCIe (for the ingest-rate report)
        self._src_inflight = None              # (storage pointer, event) of the last PINNED source chunk sent in place

    def wait_source(self) -> None:
        """Block until the last pinned chunk handed to append() has left host memory.  A pinned source is transferred
        from where it lies (no staging copy), asynchronously: refilling that buffer before this returns would race with
        the DMA engine.  Pageable / numpy sources are copied to the sparsifier's own pinned buffers inside append()."""
        if self._src_inflight is not None:
            self._src_inflight[1].synchronize()
            self._src_inflight = None

    def append(self, chunk) -> None:
        """chunk: [m, p] points as rows (numpy array or torch tensor, host or device; float64 / float32 / uint8 /
        int16 / int32).  A PINNED host tensor is read asynchronously after this returns: call wait_source() before
        overwriting it."""
        dev = self._dev
        t = torch.from_numpy(np.ascontiguousarray(chunk)) if isinstance(chunk, np.ndarray) else chunk
        if t.dtype not in _WIDEN_KIND and t.dtype != torch.float64:
            t = t.to(torch.float64)
        t = t.contiguous()
        m = t.shape[0]
        assert t.dim() == 2 and t.shape[1] == self.p and self.filled + m <= self.n
        # the stream the library launches on (the context's), not whatever torch's current stream happens to be now
        main = torch.cuda.ExternalStream(self.ctx.stream, device=dev) if self.ctx.stream else torch.cuda.default_stream(dev)
        if self._buf is None or self._buf.shape[0] < m:
            self._buf = torch.empty((m, self.p), dtype=torch.float64, device=dev)
        buf = self._buf[:m]
        if t.is_cuda:
            src = t
        else:
            b = self._turn
            self._turn ^= 1
            if self._stage[b] is None or self._stage[b].shape[0] < m or self._stage[b].dtype != t.dtype:
                if self._ev_free[b] is not None:
                    self._ev_free[b].synchronize()           # kernels on the context's stream may still read the old block
                self._stage[b] = torch.empty((m, self.p), dtype=t.dtype, device=dev)
                self._ev_free[b] = None
            host = t
            if not t.is_pinned():
                if self._pin[b] is None or self._pin[b].shape[0] < m or self._pin[b].dtype != t.dtype:
                    self._pin[b] = torch.empty((m, self.p), dtype=t.dtype, pin_memory=True)
                    self._ev_pin_free[b] = None
                if self._ev_pin_free[b] is not None:
                    self._ev_pin_free[b].synchronize()       # the previous transfer out of this pinned buffer is done
                _parallel_host_copy(self._pin[b][:m], t)      # the host-side "read" of the chunk
                host = self._pin[b][:m]
            with torch.cuda.stream(self._copy_stream):
                if self._ev_free[b] is not None:
                    self._copy_stream.wait_event(self._ev_free[b])   # the kernels that read this staging buffer are done
                self._stage[b][:m].copy_(host, non_blocking=True)
                self._ev_copied[b].record(self._copy_stream)
                if not t.is_pinned():
                    self._ev_pin_free[b] = torch.cuda.Event()
                    self._ev_pin_free[b].record(self._copy_stream)
                else:
                    # a pinned source is sent from where it lies: the caller must not refill it before this event
                    # (wait_source())
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                    self._src_inflight = (t.untyped_storage().data_ptr(), ev)
            main.wait_event(self._ev_copied[b])
            src = self._stage[b][:m]
            self.bytes_in += m * self.p * t.element_size()
        if src.dtype == torch.float64:
            fin = src if src.is_contiguous() else src.contiguous()
        else:
            _lib.check(_lib.lib().spkm_widen_f64_dev(self.ctx.handle, _WIDEN_KIND[src.dtype], m * self.p, _p(src), _p(buf)),
                       "spkm_widen_f64_dev")
            fin = buf
        o = self.filled * self.s
        if self.records:
            mix_sample_records_device(self.ctx, fin, self.p2, self.sign, 1.0 + 2.0 * float(np.finfo(np.float64).eps),
                                      float(np.sqrt(np.float64(self.p2))), self.s, self.seed, self.first + self.filled,
                                      self.rec[self.filled * self.R:], self.ir_bits)
        else:
            mix_sample_device(self.ctx, fin, self.p2, self.sign, 1.0 + 2.0 * float(np.finfo(np.float64).eps),
                              float(np.sqrt(np.float64(self.p2))), self.s, self.seed, self.first + self.filled,
                              self.ir[o:], self.x[o:])
        if not t.is_cuda:
            self._ev_free[b] = torch.cuda.Event()
            self._ev_free[b].record(main)
        self.filled += m

    def finish(self) -> Shard:
        assert self.filled == self.n, f"expected {self.n} points, got {self.filled}"
        self._stage = [None, None]
        self._pin = [None, None]
        self._buf = None
        if self.records:
            return Shard.from_records(self.ctx, self.p2, self.n, self.s, self.rec, self.ir_bits)
        jc = torch.arange(0, (self.n + 1) * self.s, self.s, dtype=torch.int64, device=self._dev)
        return Shard.from_device(self.ctx, self.p2, jc, self.ir, self.x, nnz=self.n * self.s)
