"""Host driver: the reference's entry point ``kmeans_sparsified`` and its helper
``findClusterAssignments`` (same names, options and error behaviour), with the hot loops on
the MI355X through libspkm.so.

Scope: the sparsified path -- 'Sparsify',true with the Hadamard sketch or no
sketch -- including the two-pass outputs (nargout 6..9).  What the reference does with MATLAB toolboxes
outside that path (matfile containers, function-handle sketches) raises NotImplementedError naming the option,
rather than silently doing something else.  The DCT sketch ('auto' picks it when p is not a power of two) is a
p x p orthonormal matrix applied with a library GEMM.  'Sparsify',false -- the reference's default -- runs plain Lloyd on the dense data with the
dense kernels of the two-pass outputs (one GPU, data resident in HBM).  'MLcorrection',false (plain means of the sparse columns,
kmeans_sparsified.m:449-451) runs on the same accumulation with a different final division.

MATLAB's RNG cannot be reproduced here; every random product (sign vector, sampled rows, initial
centres) comes from ``rng`` (a numpy Generator or seed), so runs are reproducible per seed but
not bit-comparable with a MATLAB run.  Given the same random products the Lloyd iteration itself
is: assignments bit-exact, centroids within 1e-6 relative (tests/).

Indices follow the reference: IDX is 1-based (values 1..K).

Multi-GPU (one process per GPU, torch.distributed initialised): every rank calls kmeans_sparsified with ITS
block of points, the same ``rng`` seed, ``first`` = global index of its first point and ``n_total``.  Random
products are drawn identically on all ranks; the sample of a point depends on (seed, global index) only, so
the run clusters exactly the dataset a single process would.  IDX / D come back for the local block, C and
SUMD are global.
"""
from __future__ import annotations

import ctypes as C
import time
import warnings

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from . import distributed as D_
from . import synth
from .engine import (LloydEngine, Shard, StreamingSparsifier, dense_accumulate_device, dense_assign_device, mix_device,
                     torch_context)

EPS = np.finfo(np.float64).eps

# kmeans_sparsified.m:130-155 (name, default)
_DEFAULTS = dict(
    Replicates=1, Start="Arthur", MaxIter=100, Display=False, PrintEvery=10, Tol=1e-6, Sparsify=False,
    SparsityLevel=0.01, SketchType="auto", EmptyAction="singleton", ColumnSamples=False, MLcorrection=True,
    DataFile=None, MB_limit=500, DataFileVerbose=False, SparsityIgnoreUpsampling=False, FORCE_BUG=False,
    tryBuiltinMex=True, unbiasedDistance=True, unbiasedInitialization=True, denseCenters=False)
_EXTRA = dict(rng=None, device=None, nargout=5, first=0, n_total=None)  # Python-side additions (not reference options)


def _parse(opts: dict) -> dict:
    canon = {k.lower(): k for k in list(_DEFAULTS) + list(_EXTRA)}
    out = dict(_DEFAULTS)
    out.update(_EXTRA)
    for k, v in opts.items():
        ck = canon.get(k.lower())
        if ck is None:
            raise TypeError(f"'{k}' is not a recognized parameter")  # inputParser behaviour
        out[ck] = v
    if isinstance(out["Display"], str) and out["Display"].lower() not in ("off", "iter", "final"):
        raise ValueError("Display must be 'off', 'iter' or 'final'")  # kmeans_sparsified.m:136-137
    if not (0 < out["SparsityLevel"] <= 1):
        raise ValueError("SparsityLevel must satisfy 0 < x <= 1")     # :141
    if str(out["EmptyAction"]).lower() not in ("singleton", "error", "drop"):
        raise ValueError("invalid EmptyAction choice")                # :143-144
    return out


def _nextpow2(p: int) -> int:
    return 1 << max(0, int(np.ceil(np.log2(p)))) if p > 1 else 1


class _Sketch:
    """mix / unmix of kmeans_sparsified.m:238-296 on device tensors laid out [n, p]."""

    def __init__(self, ctx, kind: str, p: int, sign: np.ndarray | None):
        self.ctx, self.kind, self.p = ctx, kind, p
        self.p2 = _nextpow2(p) if kind == "hadamard" else p
        self.sign = None if sign is None else torch.tensor(sign, dtype=torch.float64, device=f"cuda:{ctx.device}")
        if kind == "dct":
            # orthonormal DCT-II as MATLAB's dct(): y(k) = w(k) sum_n x(n) cos(pi (2n-1)(k-1) / (2N)), w(1) = 1/sqrt(N),
            # w(k>1) = sqrt(2/N); idct is its transpose (kmeans_sparsified.m:256-258).  A p x p matrix applied with a
            # library GEMM: one pass over the data, and the reference's own dct is a toolbox FFT whose rounding is
            # not specified either (tolerance parity).
            k = torch.arange(p, dtype=torch.float64, device=f"cuda:{ctx.device}")[:, None]
            nn_ = torch.arange(p, dtype=torch.float64, device=f"cuda:{ctx.device}")[None, :]
            M = torch.cos(np.pi * (2.0 * nn_ + 1.0) * k / (2.0 * p)) * np.sqrt(2.0 / p)
            M[0] = M[0] / np.sqrt(2.0)
            self.M = M                                                            # [p, p]: y = M x

    def mix(self, x: torch.Tensor, premul: float = 1.0) -> torch.Tensor:
        if self.kind == "none":
            return x * premul if premul != 1.0 else x
        if self.kind == "dct":
            xs = x * self.sign if premul == 1.0 else (x * premul) * self.sign     # DD*X (:283-291), rows = points
            return (xs @ self.M.T).contiguous()
        # H(DD*upsample(x)) with H(x) = hadamard(x)/sqrt(p2)   (:241-248,286-295)
        return mix_device(self.ctx, x.contiguous(), self.p2, self.sign, premul, float(np.sqrt(np.float64(self.p2))))

    def unmix(self, y: torch.Tensor) -> torch.Tensor:
        if self.kind == "none":
            return y
        if self.kind == "dct":
            return ((y @ self.M) * self.sign).contiguous()                        # DD*idct(Y) (:296)
        # downsample(DD*Ht(y)), Ht = H (:255,296)
        z = mix_device(self.ctx, y.contiguous(), self.p2, None, 1.0, float(np.sqrt(np.float64(self.p2))))
        return (z * self.sign)[:, : self.p].contiguous()


def findClusterAssignments(X, centers, tryBuiltinMex=None, gamma=None, ctx=None):
    """[assignments, distances] = findClusterAssignments(X, centers, tryBuiltinMex, gamma)
    (private/findClusterAssignments.m).  Sparse X (p x n scipy matrix): dense centres use the tiled HIP
    kernel, sparse centres (scipy matrix) the sparse-centres kernel.  Dense X (p x n array): the expanded
    quadratic of :157-165 on the f64 matrix cores (gamma is ignored there, as in the reference).
    assignments are 1-based."""
    if not sp.issparse(X):
        ctx = ctx or torch_context()
        Xd = np.asarray(X, np.float64)
        Cd = np.asarray(centers.toarray() if sp.issparse(centers) else centers, np.float64)
        if Cd.shape[0] != Xd.shape[0]:
            raise ValueError("Array of centers not of correct size")  # :55
        dev = f"cuda:{ctx.device}"
        a, d = dense_assign_device(ctx, torch.tensor(np.ascontiguousarray(Xd.T), device=dev),
                                   torch.tensor(np.ascontiguousarray(Cd.T), device=dev))
        return a.cpu().numpy().astype(np.int64) + 1, d.cpu().numpy()
    ctx = ctx or torch_context()
    p, n = X.shape
    if centers.shape[0] != p:
        raise ValueError("Array of centers not of correct size")  # :55
    K = centers.shape[1]
    eng = LloydEngine(Shard.from_scipy(ctx, X), K, gamma if gamma else 1.0, unbiased=bool(gamma))
    dev = f"cuda:{ctx.device}"
    if sp.issparse(centers):
        Cd = np.ascontiguousarray(centers.toarray().T)
        M = np.ascontiguousarray((centers != 0).toarray().T.astype(np.uint8))
        eng.assign_sparse_step(torch.tensor(Cd, device=dev), torch.tensor(M, device=dev))
    else:
        eng.assign_step(torch.tensor(np.ascontiguousarray(np.asarray(centers, np.float64).T), device=dev))
    return eng.assign.cpu().numpy().astype(np.int64) + 1, eng.mind.cpu().numpy()


def _weighted_draw(rng, w):
    """randsample(n,1,true,w) (Arthur_initialization.m:50): one index, probability ∝ w."""
    c = np.cumsum(w)
    return int(min(np.searchsorted(c, rng.random() * c[-1], side="right"), len(w) - 1))


def kmeans_sparsified(X, K, **options):
    """[IDX, C, SUMD, D, OUTPUT] = kmeans_sparsified(X, K, 'Name', value, ...)   (kmeans_sparsified.m:1)

    X: n x p array (points are rows; 'ColumnSamples',True for p x n).  Returns the tuple
    (IDX, C, SUMD, D, OUTPUT); IDX is 1-based.  With nargout=6..9 the two-pass outputs follow:
    (..., C_twoPass, IDX_twoPass, D_twoPass, SUMD_twoPass)[:nargout].  See module docstring for scope."""
    t0 = time.time()
    o = _parse(options)
    nargout = int(o["nargout"])
    if not 1 <= nargout <= 9:
        raise ValueError("nargout must be between 1 and 9")
    if isinstance(X, str):
        o["DataFile"], X = X, None                                                # kmeans_sparsified.m:179-183
    LoadFromDisk = o["DataFile"] is not None
    if not o["Sparsify"]:
        return _kmeans_dense(X, K, o, nargout, t0)
    MLcorrection = bool(o["MLcorrection"]) and bool(o["Sparsify"])   # :171
    rng = o["rng"] if isinstance(o["rng"], np.random.Generator) else np.random.default_rng(o["rng"])
    ctx = torch_context(o["device"])
    dev = f"cuda:{ctx.device}"
    Display = o["Display"] if isinstance(o["Display"], str) else "off"
    OUTPUT = dict(LoadFromDisk=LoadFromDisk, Options=dict(o), Sparsify=True)

    if LoadFromDisk:
        # the reference streams a MATLAB v7.3 (HDF5) file through matfile(); here: a .npy file, memory-mapped
        # and read MB_limit megabytes at a time (private/sampleAndMixFromLargeFile.m:79-129)
        t1 = time.time()
        fn = o["DataFile"] if str(o["DataFile"]).endswith(".npy") else str(o["DataFile"]) + ".npy"
        try:
            Xmm = np.load(fn, mmap_mode="r")
        except FileNotFoundError:
            raise FileNotFoundError("Cannot find specified data file to load")      # :190
        if Xmm.ndim != 2:
            raise ValueError("Error reading file; returned bad size for matrix")    # :210-212
        p, n = (Xmm.shape if o["ColumnSamples"] else Xmm.shape[::-1])
        OUTPUT["TimeToReadSizeOfFile"] = time.time() - t1
        X = None
    else:
        if np.iscomplexobj(X):
            raise ValueError("Code and distance computations require real data")   # :312-314
        X = np.asarray(X)
        # float32 / uint8 / int16 data stays as it is on the host (it crosses PCIe narrow and is widened on the device,
        # exactly); everything else becomes float64 as in MATLAB
        keep_narrow = X.dtype in (np.float32, np.uint8, np.int16) and str(o["SketchType"]).lower() in ("auto", "hadamard")
        if not keep_narrow:
            X = np.asarray(X, np.float64)
        if not o["ColumnSamples"]:
            X = X.T                                                                 # :214-216 (points become columns)
        p, n = X.shape
    dist_on = D_.is_distributed()
    first = int(o["first"]) if dist_on else 0
    n_glob = int(o["n_total"]) if (dist_on and o["n_total"]) else n
    if dist_on and o["n_total"] is None:
        raise ValueError("distributed run: pass n_total (and first) so that all ranks agree on the dataset")
    if n_glob < K:
        raise ValueError("X must have more samples than the number of clusters.")  # :219-221

    # ---- sketch (:224-296) ----
    sk = o["SketchType"]
    if isinstance(sk, (list, tuple)):
        raise NotImplementedError("function-handle sketches run in MATLAB, not here")
    sk = str(sk).lower()
    if sk == "auto":
        sk = "hadamard" if p == _nextpow2(p) else "dct"                          # :226-231
        OUTPUT["SketchType"] = "Hadamard" if sk == "hadamard" else "DCT"
    if sk == "dct":
        d = np.sign(rng.random(p)) if o["FORCE_BUG"] else np.sign(rng.standard_normal(p))   # :283-287 (p2 = p here)
        d[d == 0] = 1.0
        if p > 16384:
            raise NotImplementedError("the DCT sketch is applied as a p x p matrix; p is too large for that")
        sketch = _Sketch(ctx, "dct", p, d)
    elif sk in ("nothing", "none"):
        sketch = _Sketch(ctx, "none", p, None)
    elif sk == "hadamard":
        p2 = _nextpow2(p)
        d = np.sign(rng.random(p2)) if o["FORCE_BUG"] else np.sign(rng.standard_normal(p2))  # :283-287
        d[d == 0] = 1.0
        sketch = _Sketch(ctx, "hadamard", p, d)
        OUTPUT["SlowHadamard"] = False
    else:
        raise ValueError('bad type for "SketchType"')                            # :273
    p2 = sketch.p2

    small_p = synth.small_p_of(o["SparsityLevel"], p2)                           # :324-326
    gamma = small_p / p                                                          # :329 (divides by p, not p2)
    sample_seed = int(rng.integers(0, 2**63 - 1))
    Y = None
    if sk == "hadamard" and 16 <= p2 <= 16384:
        # device sparsifier: chunk -> X*(1+2eps) -> mix -> sample -> resident CSC (kmeans_sparsified.m:292-334;
        # for 'DataFile': sampleAndMixFromLargeFile.m:100-129).  The dense mixed data never reaches HBM.
        t1 = time.time()
        sp_ = StreamingSparsifier(ctx, p, n, small_p, sample_seed, sketch.sign, first=first)
        nn = max(1, min(n, int(o["MB_limit"] * 2**20 // (8 * p))))               # sampleAndMixFromLargeFile.m:82-84
        if LoadFromDisk and o["DataFileVerbose"]:
            print(f"Splitting {p} x {n} matrix into {-(-n // nn)} {p} x {nn} chunks")
        for c0 in range(0, n, nn):
            if LoadFromDisk:
                blk = Xmm[:, c0:c0 + nn].T if o["ColumnSamples"] else Xmm[c0:c0 + nn, :]
            else:
                blk = X[:, c0:c0 + nn].T
            sp_.append(np.ascontiguousarray(blk))
        shard = sp_.finish()
        vals_ = sp_.x[: n * small_p]
        # one Inf / NaN entry makes its whole mixed column non-finite (every output of the transform is a signed sum
        # of all inputs), so the sampled values tell.  The reference has no such check: its run ends in
        # error('Found NaN in centers') (:480-484) a few iterations later; here the data is refused up front, because
        # the argmin kernels are specified for finite distances only.
        if not bool(torch.isfinite(vals_).all().item()):
            raise ValueError("X must be finite (Inf / NaN entries found)")
        nnz = n * small_p
        nzm_ = vals_ != 0
        if not bool(nzm_.all().item()):
            # sparse(i,j,v) drops entries that are exactly 0 (randsample_fixedNumberEntries.m:62): they are absent from
            # the masked distance and from spones(X) (the Cnt denominator, :352-355).  All-zero points, or exact +-
            # cancellation in the transform of integer data.  Rebuild the shard as a ragged CSC without them.
            cnt_ = nzm_.view(n, small_p).sum(dim=1)
            jc_ = torch.zeros(n + 1, dtype=torch.int64, device=dev)
            jc_[1:] = torch.cumsum(cnt_, 0)
            nnz = int(jc_[-1].item())
            x2_ = torch.zeros(nnz + 48, dtype=torch.float64, device=dev)
            ir2_ = torch.zeros(nnz + 48, dtype=sp_.ir.dtype, device=dev)
            x2_[:nnz] = vals_[nzm_]
            ir2_[:nnz] = sp_.ir[: n * small_p][nzm_]
            shard = Shard.from_device(ctx, p2, jc_, ir2_, x2_, nnz=nnz)
        del nzm_, vals_
        torch.cuda.synchronize()
        OUTPUT["TimeToSketch"] = OUTPUT["TimeToSample"] = time.time() - t1       # fused: one number for both
    else:
        # DCT / no sketch: mix on the device (a GEMM or nothing), sample on the host (randsample_fixedNumberEntries,
        # :334), MB_limit columns at a time -- the same generator runs through all chunks, so a 'DataFile' run
        # draws exactly the samples of the in-memory run (sampleAndMixFromLargeFile.m:100-129)
        nn = max(1, min(n, int(o["MB_limit"] * 2**20 // (8 * p)))) if LoadFromDisk else n
        # the stream depends on (seed, offset of this rank's block): ranks of a distributed run draw different row
        # patterns, a single process draws what it always drew
        srng = np.random.default_rng(sample_seed if first == 0 else [sample_seed, first])
        t_mix = t_smp = 0.0
        parts_ = []
        for c0 in range(0, n, nn):
            if LoadFromDisk:
                blk = Xmm[:, c0:c0 + nn].T if o["ColumnSamples"] else Xmm[c0:c0 + nn, :]
            else:
                blk = X[:, c0:c0 + nn].T
            t1 = time.time()
            Xmixed = sketch.mix(torch.tensor(np.ascontiguousarray(blk, dtype=np.float64), device=dev),
                                premul=1.0 + 2.0 * EPS)                          # :292,295 (X*(1+2eps) then mix)
            torch.cuda.synchronize()
            t_mix += time.time() - t1
            t1 = time.time()
            parts_.append(synth.sparsify_dense(Xmixed.cpu().numpy().T, small_p, srng))
            t_smp += time.time() - t1
        OUTPUT["TimeToSketch"], OUTPUT["TimeToSample"] = t_mix, t_smp
        Y = parts_[0] if len(parts_) == 1 else sp.hstack(parts_, format="csc")
        if not np.all(np.isfinite(Y.data)):
            raise ValueError("X must be finite (Inf / NaN entries found)")
        shard = Shard.from_scipy(ctx, Y)
        nnz = Y.nnz
    # obj = sqrt(sum(distances.^2)) is evaluated in every iteration (:471) but used only by Display='iter' (:472-475) and,
    # after the loop, for the iteration that turned out to be the last (:489-503): unless it is displayed per iteration
    # the library may leave it (and the exact pass that produces it) out of a fused call; it is then obtained on demand
    # together with the distances (spkm_shard_set_lazy_stats, LloydEngine.distances)
    lazy_stats = Display != "iter"
    shard.set_lazy_stats(lazy_stats)
    if Display in ("iter", "final"):
        print(f"Randomly mixing of type {sk}")
        print(f"Randomly taking {100 * gamma:.1f}% of the data; actual dataset is {100 * nnz / (p2 * n):.1f}% sparse")

    def local_column(i):
        """dense copy of local sparse column i"""
        if Y is not None:
            return Y[:, i].toarray().ravel()
        rows, vals = shard.column(i)         # from the CSC arrays or, once they are released, from the record layout
        col = np.zeros(p2)
        col[rows] = vals
        return col

    def column(gi):
        """dense copy of the sparse column with GLOBAL index gi ('sample' / k-means++ starts,
        EmptyAction='singleton'); in a distributed run its owner broadcasts it"""
        if not dist_on:
            return local_column(gi)
        mine = first <= gi < first + n
        flag = torch.tensor([float(torch.distributed.get_rank()) if mine else -1.0], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        col = torch.tensor(local_column(gi - first), device=dev) if mine else torch.zeros(p2, dtype=torch.float64, device=dev)
        D_.broadcast_column(col, int(flag.item()))
        return col.cpu().numpy()

    unbiased = bool(o["unbiasedDistance"])                                       # :369-373
    start = o["Start"]
    Replicates = int(o["Replicates"])
    OUTPUT.update(iterations=np.zeros(Replicates, int), stoppingDiff=np.zeros(Replicates),
                  objectives=np.zeros(Replicates), replicateTimes=np.zeros(Replicates),
                  replicateTimesJustInitialization=np.zeros(Replicates))
    if isinstance(start, str) and start.lower() == "uniform":
        vals = Y.data if Y is not None else sp_.x[: n * small_p].cpu().numpy()
        mn, mx = float(vals.min(initial=0.0)), float(vals.max(initial=0.0))      # full(min(X(:))) incl. implicit zeros
        if nnz < p2 * n:
            mn, mx = min(mn, 0.0), max(mx, 0.0)
        if dist_on:
            mm = torch.tensor([-mn, mx], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(mm, op=torch.distributed.ReduceOp.MAX)
            mn, mx = -float(mm[0].item()), float(mm[1].item())

    best = dict(obj=np.inf)
    distances = None
    K_start = K                               # K of a 'Start' matrix
    for trial in range(Replicates):
        t1 = time.time()
        Kc = K                                # (after an EmptyAction='drop' the reference keeps the smaller K, :458)
        sparse_mask = None                    # [K, p2] uint8 while the centres are sparse
        if isinstance(start, str):
            s = start.lower()
            if s == "sample":
                ind = rng.choice(n_glob, K, replace=False)                       # randsample(n,K) (:387)
                centers_np = np.stack([column(int(i)) for i in ind], axis=1)
                sparse_mask = (centers_np != 0).astype(np.uint8)
            elif s == "uniform":
                centers_np = (mx - mn) * rng.random((p2, K)) - mn                # :390 (the reference subtracts mn)
            elif s in ("arthur", "++", "kmeans++", "k-means++", "k-means-++"):
                g_init = gamma if o["unbiasedInitialization"] else None          # :392-396
                centers_np, sparse_mask = _arthur(ctx, shard, column, n, K, g_init, rng, first, n_glob, dist_on)
            else:
                raise ValueError('cannot handle other types of "Start" values')  # :398
        else:
            S = np.asarray(start, np.float64)
            if not o["ColumnSamples"]:
                S = S.T                                                          # want p x K (:403-405)
            if S.shape != (p, K_start):
                raise ValueError("Start matrix must be K x p (or p x K with ColumnSamples)")
            Kc = K_start
            centers_np = sketch.mix(torch.tensor(np.ascontiguousarray(S.T), device=dev)).cpu().numpy().T  # :406
            if Replicates > 1:
                warnings.warn("initialization is specified, so running more than 1 replicate is not helpful")
        if o["denseCenters"]:
            sparse_mask = None                                                   # centers = full(centers) (:412-414)
        OUTPUT["replicateTimesJustInitialization"][trial] = time.time() - t1

        shard.reset_policy()                                                     # new start: nothing learned carries over
        eng = LloydEngine(shard, Kc, gamma, unbiased=unbiased)
        centers = torch.tensor(np.ascontiguousarray(centers_np.T), device=dev)   # [K, p2]
        if not bool(torch.isfinite(centers).all().item()):
            raise ValueError("initial centers must be finite")                   # (see the check on X above)
        mask_t = None if sparse_mask is None else torch.tensor(np.ascontiguousarray(sparse_mask.T), device=dev)
        its = 0
        dff = obj = np.nan
        assignments = None
        csc_released = False
        dist_t = eng.mind                     # min-distances of the latest iteration (survives a 'drop' re-build of eng)
        mind_pending, eng_used, centers_used = False, eng, centers
        fused_iters = 0
        for its in range(1, int(o["MaxIter"]) + 1):
            # [assignments,distances] = findClusters(X,centers) (:420) and the per-cluster sums of :430-453
            host_res = None
            if mask_t is not None:
                eng.assign_sparse_step(centers, mask_t)                          # findClusterAssignments.m:63-75
                eng.accumulate_step()
            else:
                # dense centres: the fused call -- the library's fast path (certified screen, carried bounds) when the
                # shard qualifies, the exact kernels otherwise; same assignments, counts and distances bit for bit, per-cluster sums to
                # the order of summation (findClusterAssignments.m:76-82)
                # (per-point distances are not stored per iteration -- a gigabyte of stores per 1e8 points -- but
                #  produced once after the loop for the iteration that turned out to be the last: spkm_distances_dev)
                old = centers.clone()
                if MLcorrection:
                    # assignment + accumulation + all-reduce + gamma*S./(Cnt+1e-16) + dff + obj: ONE library call
                    # (spkm_lloyd_iter; kmeans_sparsified.m:420-471)
                    # (the results this loop decides on -- dff^2, obj^2, cluster sizes -- arrive in host memory with the
                    #  call: spkm_lloyd_iter_host)
                    host_res = eng.iterate_host(centers, want_mind=False)
                else:
                    eng.assign_accumulate_step(centers, want_mind=False)
                fused_iters += 1
                if Y is None and not csc_released and fused_iters == 1:
                    # the first fused call has built the library's record layout and screen copy: the CSC value / row
                    # arrays of the device-built shard can go (spkm_shard_release_csc; an entry point that needs them
                    # -- the k-means++ rounds of the next replicate -- brings them back from the records)
                    torch.cuda.synchronize()
                    if shard.release_csc():
                        sp_.x = sp_.ir = None
                        torch.cuda.empty_cache()
                    csc_released = True
            if mask_t is not None:
                old = centers.clone()
            mind_pending = mask_t is None          # the distances of THIS iteration still have to be materialised
            eng_used, centers_used = eng, old      # ... by this engine, under these centres (kept across a 'drop')
            dist_t = eng.mind
            pk_ = p2 * Kc
            if MLcorrection and mask_t is None:
                dff2_t = eng.out[0:1]                                            # (finalised inside the call above)
            elif MLcorrection:
                eng.allreduce_step()
                eng.finalize_step(centers)                                       # gamma*S./(Cnt+1e-16)  (:447-448)
                dff2_t = eng.out[0:1]
            else:
                eng.allreduce_step()
                # centers(:,ki) = mean(full(X(:,ind)),2) (:449-451): plain mean of the sparse columns, zeros included
                nk_ = eng.reduce[2 * pk_: 2 * pk_ + Kc]
                mean_ = eng.reduce[:pk_].view(Kc, p2) / torch.clamp(nk_, min=1.0)[:, None]
                centers.copy_(torch.where((nk_ > 0)[:, None], mean_, centers))
                dff2_t = ((old - centers) ** 2).sum().reshape(1)
            # ONE small device-to-host read per iteration: [dff^2 | obj^2 | cluster sizes] (none when the fused call brought them)
            if host_res is not None:
                host = host_res
            else:
                host = torch.cat([dff2_t, eng.reduce[2 * pk_ + Kc: 2 * pk_ + Kc + 1], eng.reduce[2 * pk_: 2 * pk_ + Kc]]).cpu().numpy()
            dff2, obj2, nk = float(host[0]), float(host[1]), host[2:]
            empty = np.flatnonzero(nk == 0)
            if empty.size and np.isnan(obj2) and mind_pending:
                # a lazy call (no objective, no largest distance) and EmptyAction is about to need them: this iteration's
                # distances now, under the centres its assignment was computed with
                eng.distances(old)
                mind_pending, dist_t = False, eng.mind
                o2 = eng.stats[0:1].clone()
                if dist_on:
                    torch.distributed.all_reduce(o2, op=torch.distributed.ReduceOp.SUM)
                obj2 = float(o2.item())
            obj = float(np.sqrt(obj2))                                           # sqrt(sum(distances.^2)) (:471); NaN: not evaluated yet
            dropped = False
            if empty.size:
                warnings.warn("cluster has lost all its members")                # :433
                act = str(o["EmptyAction"]).lower()
                if act == "error":
                    raise RuntimeError("One cluster lost all its members")      # :439
                if act == "singleton":
                    st = eng.stats.cpu().numpy()                                 # [~,iMax] = max(distances) (:436)
                    _, imax, _ = D_.global_first_argmax(float(st[1]), int(st[2]), first)
                    col = torch.tensor(column(imax), device=dev)
                    for ki in empty:
                        centers[ki] = col                                        # centers(:,ki) = X(:,iMax) (:437)
                    # the finalize kernel's dff covers the clusters it updated; the replaced columns come on top
                    dff2 = float(((old - centers) ** 2).sum().item())
                else:                                                            # 'drop' (:441,454-459)
                    keep = np.setdiff1d(np.arange(Kc), empty)
                    keep_t = torch.tensor(keep, device=dev)
                    centers = centers[keep_t].contiguous()
                    old = old[keep_t].contiguous()
                    Kc = keep.size
                    if isinstance(start, str):
                        K = Kc    # the reference overwrites K itself (:458), so later replicates start with fewer clusters
                    # distances / obj of THIS iteration stay what they are (dist_t, obj above: :471 reads `distances`);
                    # only the engine's per-K buffers are rebuilt for the next one
                    eng = LloydEngine(shard, Kc, gamma, unbiased=unbiased)
                    dropped = True
                    dff2 = float(((old - centers) ** 2).sum().item())            # over the kept columns (:455-456,470)
            if mask_t is not None:
                # after the first ML update the columns are (almost) full: issparse && nnz > .99 -> full (:460-464)
                filled = float((centers != 0).double().mean().item())
                if filled > 0.99:
                    mask_t = None
                else:
                    mask_t = (centers != 0).to(torch.uint8).contiguous()
            dff = float(np.sqrt(dff2))                                           # norm(centersOld-centers,'fro') (:470)
            assignments = None if dropped else eng.assign
            if Display == "iter" and its % int(o["PrintEvery"]) == 0:
                print(f"Iter: {its:3d}; change in cluster centers: {dff:.2e}; objective: {obj:.2e}")
            if dff < o["Tol"]:
                break                                                            # :476-478
            if not np.isfinite(dff2) and bool(torch.isnan(centers).any().item()):
                raise RuntimeError("Found NaN in centers")                       # :480-484 (a NaN centre makes dff NaN)
        last_path = eng.last_path_info()[0] if fused_iters else 0
        if its > 0 and mind_pending:
            eng_used.distances(centers_used)                                     # `distances` of the last iteration (:420)
            dist_t = eng_used.mind
            if np.isnan(obj):                                                    # ... and its objective (:471), left out by a lazy call
                o2 = eng_used.stats[0:1].clone()
                if dist_on:
                    torch.distributed.all_reduce(o2, op=torch.distributed.ReduceOp.SUM)
                obj = float(np.sqrt(o2.item()))
        OUTPUT["replicateTimes"][trial] = time.time() - t1
        OUTPUT["stoppingDiff"][trial], OUTPUT["objectives"][trial], OUTPUT["iterations"][trial] = dff, obj, its
        # (not reference fields) how many iterations went through the fused call, and which path the library took for
        # the last of them: 1 = certified screen + exact confirmation, 0 = all-exact kernels
        OUTPUT.setdefault("fusedIterations", np.zeros(Replicates, int))[trial] = fused_iters
        OUTPUT.setdefault("lastPath", np.zeros(Replicates, int))[trial] = last_path
        distances = dist_t.cpu().numpy()
        if obj < best["obj"]:                                                    # :493-503
            best = dict(obj=obj, K=Kc, centers=centers.clone(),
                        assign=None if assignments is None else assignments.cpu().numpy().astype(np.int64) + 1,
                        dist=distances.copy())
        if Display == "iter" or (Display == "final" and best["obj"] == obj):
            print(f"Trial {trial + 1:3d} of {Replicates:3d} total, objective {obj:.2e}")

    OUTPUT["TimeInitialization"] = float(OUTPUT["replicateTimesJustInitialization"].sum())
    OUTPUT["TimeAlgo_wo_initialization"] = float(OUTPUT["replicateTimes"].sum()) - OUTPUT["TimeInitialization"]
    Kb = best["K"]
    IDX = best["assign"] if best["assign"] is not None else np.zeros(0, np.int64)
    # :514-518, SUMD(ki) = sum(distances(IDX==ki).^2) with the LAST trial's distances -- in one pass over the points
    # instead of K (at n = 1e7, K = 100 the K masked sums took a second, five times the Lloyd loop)
    SUMD = np.zeros(Kb)
    if IDX.size:
        ok = (IDX >= 1) & (IDX <= Kb)                                            # ('drop' blanks assignments to 0)
        SUMD = np.bincount(IDX[ok] - 1, weights=distances[ok] ** 2, minlength=Kb)[:Kb].astype(np.float64)
    if dist_on:                                                                  # SUMD is a sum over all points
        sd = torch.tensor(SUMD, dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(sd, op=torch.distributed.ReduceOp.SUM)
        SUMD = sd.cpu().numpy()
    OUTPUT["TimeOverall_OnePass"] = time.time() - t0
    Cdev = sketch.unmix(best["centers"])                                         # [K, p] (:523)
    Cout = Cdev.cpu().numpy().T                                                  # p x K
    D = best["dist"]
    extra = ()
    if nargout > 5:
        # ---- two-pass outputs (:525-571): a second pass over the UNSAMPLED data, streamed in MB_limit chunks ----
        #  centers_twoPass(:,k) = mean(full(XFull(:,ind)),2) over the one-pass assignments (:545-550;
        #                         DataFile: recalculateAssignmentLargeFile.m:100-113)
        #  [assignments_twoPass, distances_twoPass] = findClusters(full(XFull), bestCenters)  (:558; dense branch,
        #                         which ignores gamma) -- for 'DataFile' always computed, as the reference does (:531)
        #  SUMD_twoPass(k) = sum(distances(:, assignments_twoPass==k).^2) with the ONE-pass distances of the last
        #                         replicate (:564-568; a reference quirk kept as is)
        # Reference quirk NOT kept: recalculateAssignmentLargeFile.m:105 overwrites newCenters(:,ki) with each
        # chunk's sum instead of adding to it, while the counter (:103) and the final division (:111-113) span
        # all chunks; the sums are accumulated here, which is identical whenever the file fits one chunk.
        if LoadFromDisk:
            warnings.warn("Requires a second pass over the dataset")            # :529
        t1 = time.time()
        want_assign = LoadFromDisk or nargout > 6
        sums = torch.zeros((Kb, p), dtype=torch.float64, device=dev)
        counts = torch.zeros(Kb, dtype=torch.float64, device=dev)
        a2 = torch.empty(n, dtype=torch.int32, device=dev) if want_assign else None
        d2 = torch.empty(n, dtype=torch.float64, device=dev) if want_assign else None
        idx0 = None if not IDX.size else torch.tensor((IDX - 1).astype(np.int32), device=dev)
        nn = max(1, min(n, int(o["MB_limit"] * 2**20 // (8 * p))))               # recalculateAssignmentLargeFile.m:66
        t_read = 0.0
        for c0 in range(0, n, nn):
            tr = time.time()
            if LoadFromDisk:
                blk = Xmm[:, c0:c0 + nn].T if o["ColumnSamples"] else Xmm[c0:c0 + nn, :]
            else:
                blk = X[:, c0:c0 + nn].T
            blk = np.ascontiguousarray(blk, dtype=np.float64)
            t_read += time.time() - tr
            xb = torch.tensor(blk, device=dev)
            if idx0 is not None:
                dense_accumulate_device(ctx, xb, idx0[c0:c0 + nn].contiguous(), sums, counts)
            if want_assign:
                a_, d_ = dense_assign_device(ctx, xb, Cdev)
                a2[c0:c0 + nn], d2[c0:c0 + nn] = a_, d_
        if dist_on:
            torch.distributed.all_reduce(sums, op=torch.distributed.ReduceOp.SUM)
            torch.distributed.all_reduce(counts, op=torch.distributed.ReduceOp.SUM)
        cnt = counts.cpu().numpy()
        C2 = sums.cpu().numpy().T                                                # p x K
        if LoadFromDisk:
            with np.errstate(divide="ignore", invalid="ignore"):
                C2 = C2 * (1.0 / cnt)[None, :]                                   # bsxfun(@times, newCenters, 1./counter)
        else:
            nz = cnt > 0
            C2[:, nz] = C2[:, nz] / cnt[nz][None, :]                             # mean(...,2); empty clusters stay 0 (:544)
        torch.cuda.synchronize()
        OUTPUT["TimeSecondPass_Overall" if LoadFromDisk else "TimeSecondPass_Centers"] = time.time() - t1
        if LoadFromDisk:
            OUTPUT["TimeSecondPass_JustRead"] = t_read
        extra = (C2 if o["ColumnSamples"] else C2.T,)
        if want_assign:
            IDX2 = a2.cpu().numpy().astype(np.int64) + 1
            D2 = d2.cpu().numpy()
            extra += (IDX2, D2)
            if nargout >= 9:
                t1 = time.time()
                S2 = np.bincount(IDX2 - 1, weights=distances ** 2, minlength=Kb)[:Kb].astype(np.float64)   # (one pass, as SUMD)
                if dist_on:
                    sd = torch.tensor(S2, dtype=torch.float64, device=dev)
                    torch.distributed.all_reduce(sd, op=torch.distributed.ReduceOp.SUM)
                    S2 = sd.cpu().numpy()
                OUTPUT["TimeSecondPass_SUMD"] = time.time() - t1
                extra += (S2,)
    if not o["ColumnSamples"]:
        Cout = Cout.T                                                            # K x p like MATLAB's kmeans (:586-590)
    OUTPUT["TimeOverall"] = time.time() - t0
    return ((IDX, Cout, SUMD, D, OUTPUT) + extra)[:max(nargout, 5)]


def _kmeans_dense(X, K, o, nargout, t0):
    """'Sparsify',false -- the reference's DEFAULT: plain Lloyd on the dense data, no sketch, no sampling
    (kmeans_sparsified.m:362-364,378-486 with findClusterAssignments.m:124-171 and the plain mean of :449-451).
    Assignment = spkm_dense_assign_dev (expanded quadratic on the f64 matrix cores), centres = per-cluster means
    from spkm_dense_accumulate_dev.  The data must fit in HBM as one n x p float64 tensor; 'DataFile' is not
    offered on this branch (neither does the reference's code path load it)."""
    if o["DataFile"] is not None:
        raise NotImplementedError("'DataFile' needs 'Sparsify',true (kmeans_sparsified.m:298-307 only loads it there)")
    if D_.is_distributed():
        raise NotImplementedError("'Sparsify',false runs on one GPU")
    rng = o["rng"] if isinstance(o["rng"], np.random.Generator) else np.random.default_rng(o["rng"])
    ctx = torch_context(o["device"])
    dev = f"cuda:{ctx.device}"
    X = np.asarray(X, np.float64)
    if np.iscomplexobj(X):
        raise ValueError("Code and distance computations require real data")
    if o["ColumnSamples"]:
        X = X.T                                                                   # here: points as rows [n, p]
    n, p = X.shape
    if n < K:
        raise ValueError("X must have more samples than the number of clusters.")  # :219-221
    free, _ = torch.cuda.mem_get_info()
    if n * p * 8 > 0.8 * free:
        raise NotImplementedError("'Sparsify',false keeps the dense data on the GPU; it does not fit")
    Xd = torch.tensor(np.ascontiguousarray(X), device=dev)
    Display = o["Display"] if isinstance(o["Display"], str) else "off"
    Replicates = int(o["Replicates"])
    OUTPUT = dict(LoadFromDisk=False, Options=dict(o), Sparsify=False, iterations=np.zeros(Replicates, int),
                  stoppingDiff=np.zeros(Replicates), objectives=np.zeros(Replicates), replicateTimes=np.zeros(Replicates),
                  replicateTimesJustInitialization=np.zeros(Replicates))
    start = o["Start"]
    best = dict(obj=np.inf)
    distances = None
    for trial in range(Replicates):
        t1 = time.time()
        if isinstance(start, str):
            sl = start.lower()
            if sl == "sample":
                centers = Xd[torch.tensor(rng.choice(n, K, replace=False), device=dev)].clone()       # :387
            elif sl == "uniform":
                mn, mx = float(Xd.min().item()), float(Xd.max().item())
                centers = torch.tensor((mx - mn) * rng.random((K, p)) - mn, device=dev)   # :390 (subtracts mn)
            elif sl in ("arthur", "++", "kmeans++", "k-means++", "k-means-++"):
                chosen = [int(rng.integers(n))]                                   # Arthur_initialization.m:35
                dist = None
                for k in range(1, K):
                    _, dnew = dense_assign_device(ctx, Xd, Xd[chosen[-1]][None, :].contiguous())
                    dist = dnew if dist is None else torch.minimum(dist, dnew)    # same dist vector as :39
                    cum = torch.cumsum(dist * dist, 0)
                    tot = float(cum[-1].item())

                    def draw():
                        if tot <= 0.0:
                            return int(rng.integers(n))
                        t = torch.tensor([rng.random() * tot], dtype=torch.float64, device=dev)
                        return int(min(torch.searchsorted(cum, t, right=True).item(), n - 1))
                    i, counter = draw(), 1
                    while i in chosen and counter < 400:                          # :54-61
                        i, counter = draw(), counter + 1
                    if i in chosen:
                        raise RuntimeError("Cannot sample with replacement with this distribution")
                    chosen.append(i)
                centers = Xd[torch.tensor(chosen, device=dev)].clone()
            else:
                raise ValueError('cannot handle other types of "Start" values')  # :398
        else:
            S = np.asarray(start, np.float64)
            if o["ColumnSamples"]:
                S = S.T
            if S.shape != (K, p):
                raise ValueError("Start matrix must be K x p (or p x K with ColumnSamples)")
            centers = torch.tensor(np.ascontiguousarray(S), device=dev)
            if Replicates > 1:
                warnings.warn("initialization is specified, so running more than 1 replicate is not helpful")
        OUTPUT["replicateTimesJustInitialization"][trial] = time.time() - t1
        Kc = K
        its, dff, obj, assign, dmin, dropped = 0, np.nan, np.nan, None, None, False
        for its in range(1, int(o["MaxIter"]) + 1):
            assign, dmin = dense_assign_device(ctx, Xd, centers.contiguous())     # findClusterAssignments.m:124-171
            old = centers.clone()
            sums = torch.zeros((Kc, p), dtype=torch.float64, device=dev)
            cnt = torch.zeros(Kc, dtype=torch.float64, device=dev)
            dense_accumulate_device(ctx, Xd, assign, sums, cnt)
            nz = cnt > 0
            centers[nz] = sums[nz] / cnt[nz, None]                                # mean(full(X(:,ind)),2) (:449-451)
            empty = torch.nonzero(~nz).flatten().cpu().numpy()
            dropped = False
            if empty.size:
                warnings.warn("cluster has lost all its members")                # :433
                act = str(o["EmptyAction"]).lower()
                if act == "error":
                    raise RuntimeError("One cluster lost all its members")      # :439
                if act == "singleton":
                    centers[torch.tensor(empty, device=dev)] = Xd[int(torch.argmax(dmin).item())]   # :436-437
                else:                                                            # 'drop' (:441,454-459)
                    keep = torch.nonzero(nz).flatten()
                    centers, old, Kc, dropped = centers[keep].contiguous(), old[keep].contiguous(), int(keep.numel()), True
            dff = float(torch.linalg.norm(old - centers).item())                 # :470
            obj = float(torch.sqrt((dmin * dmin).sum()).item())                  # :471
            if Display == "iter" and its % int(o["PrintEvery"]) == 0:
                print(f"Iter: {its:3d}; change in cluster centers: {dff:.2e}; objective: {obj:.2e}")
            if dff < o["Tol"]:
                break
            if bool(torch.isnan(centers).any().item()):
                raise RuntimeError("Found NaN in centers")
        OUTPUT["replicateTimes"][trial] = time.time() - t1
        OUTPUT["stoppingDiff"][trial], OUTPUT["objectives"][trial], OUTPUT["iterations"][trial] = dff, obj, its
        distances = dmin.cpu().numpy()
        if obj < best["obj"]:
            best = dict(obj=obj, K=Kc, centers=centers.clone(), dist=distances.copy(),
                        assign=None if dropped else assign.cpu().numpy().astype(np.int64) + 1)
        if Display == "iter" or (Display == "final" and best["obj"] == obj):
            print(f"Trial {trial + 1:3d} of {Replicates:3d} total, objective {obj:.2e}")
    OUTPUT["TimeInitialization"] = float(OUTPUT["replicateTimesJustInitialization"].sum())
    OUTPUT["TimeAlgo_wo_initialization"] = float(OUTPUT["replicateTimes"].sum()) - OUTPUT["TimeInitialization"]
    Kb = best["K"]
    IDX = best["assign"] if best["assign"] is not None else np.zeros(0, np.int64)
    SUMD = np.array([np.sum(distances[IDX == ki + 1] ** 2) if IDX.size else 0.0 for ki in range(Kb)])   # :514-518
    OUTPUT["TimeOverall_OnePass"] = time.time() - t0
    Cout = best["centers"].cpu().numpy()                                          # K x p
    extra = ()
    if nargout > 5:
        warnings.warn("There is no sparsification, so the twoPass variables are the same")   # :575-576
        extra = (Cout.T if o["ColumnSamples"] else Cout, IDX, best["dist"], SUMD)             # :577-582
    if o["ColumnSamples"]:
        Cout = Cout.T
    OUTPUT["TimeOverall"] = time.time() - t0
    return ((IDX, Cout, SUMD, best["dist"], OUTPUT) + extra)[:max(nargout, 5)]


def _arthur(ctx, shard, column, n, K, gamma, rng, first=0, n_glob=None, dist_on=False):
    """K-means++ seeding, private/Arthur_initialization.m:24-69: first centre uniform; then K-1 rounds of
    [~,dist] = findClusterAssignments(X, full(centres), [], gamma) and a draw ∝ dist.^2 with the 400-retry
    duplicate rule.  The reference recomputes the distances to ALL chosen centres every round (K^2/2
    centre evaluations); since min() is exact and each (point, centre) distance does not depend on the other
    centres, keeping a running minimum and evaluating only the NEW centre gives the same ``dist`` vector bit for
    bit at 1/K of the work.  Everything stays on the device; the draw is a cumulative sum + binary search
    (distributed: the rank is picked from the all-gathered block totals first, with the same random number).
    Returns (p2 x K dense values, p2 x K support mask): the centres are columns of the sparse X (:36,68)."""
    if K < 1:
        raise ValueError("K must be >= 1")
    n_glob = n if n_glob is None else n_glob
    dev = f"cuda:{ctx.device}"
    chosen = [int(rng.integers(n_glob))]                                         # randi(n,1) (:35)
    cols = [column(chosen[0])]
    eng = LloydEngine(shard, 1, gamma if gamma else 1.0, unbiased=bool(gamma))
    dist = None
    for k in range(1, K):
        c_new = torch.tensor(cols[-1][None, :], device=dev)                      # newest centre only
        if gamma:
            eng.assign_step(c_new)                                               # findDist(X, full(ref), [], gamma) (:31)
        else:
            # gamma empty: findClusterAssignments(X, ref) with the SPARSE column (:29) -> sparse-centres branch,
            # distance over supp(x) n supp(c), no scaling (findClusterAssignments.m:71-75)
            eng.assign_sparse_step(c_new, (c_new != 0).to(torch.uint8).contiguous())
        # running minimum + prefix sums of dist.^2 in the library (spkm_kpp_update_dev)
        first_round = dist is None
        if first_round:
            dist = torch.empty_like(eng.mind)
            cum = torch.empty_like(eng.mind)
        tot = C.c_double(0.0)
        _lib.check(_lib.lib().spkm_kpp_update_dev(ctx.handle, n, C.c_void_p(eng.mind.data_ptr()), C.c_void_p(dist.data_ptr()),
                                                  1 if first_round else 0, C.c_void_p(cum.data_ptr()), C.byref(tot)),
                   "spkm_kpp_update_dev")
        local_total = float(tot.value) if n > 0 else 0.0
        if dist_on:
            world = torch.distributed.get_world_size()
            tt = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            torch.distributed.all_gather(tt, torch.tensor([local_total], dtype=torch.float64, device=dev))
            totals = np.array([float(t.item()) for t in tt])
        else:
            totals = np.array([local_total])
        total = float(totals.sum())
        edges = np.concatenate([[0.0], np.cumsum(totals)])

        def draw():
            if total > 0:                                                        # norm(dist) > 0 (:49)
                u = rng.random() * total                                         # same number on every rank
                r = int(min(np.searchsorted(edges, u, side="right") - 1, len(totals) - 1))
                def local_draw(target):
                    idx = C.c_int64(0)
                    _lib.check(_lib.lib().spkm_kpp_draw_dev(ctx.handle, n, C.c_void_p(cum.data_ptr()), float(target),
                                                            C.byref(idx)), "spkm_kpp_draw_dev")
                    return int(idx.value)

                if not dist_on:
                    return local_draw(u)
                gi = torch.zeros(1, dtype=torch.float64, device=dev)
                if torch.distributed.get_rank() == r:
                    gi[0] = float(first + local_draw(u - edges[r]))
                torch.distributed.broadcast(gi, src=r)
                return int(gi.item())
            return int(rng.integers(n_glob))

        i = draw()
        counter = 1
        while i in chosen and counter < 400:                                     # :54-61
            i = draw()
            counter += 1
        if i in chosen:
            raise RuntimeError("Cannot sample with replacement with this distribution")  # :62-65
        chosen.append(i)
        cols.append(column(i))
    Cd = np.stack(cols, axis=1)
    return Cd, (Cd != 0).astype(np.uint8)
