"""Synthetic Gaussian-mixture workloads in the reference's own input format.

Follows example_sparseKMeans.m:12-22 (centres ~ N(0,I_p), contiguous equal cluster blocks,
noise 0.1*N(0,I_p)) and then the reference's preprocessing pipeline
(kmeans_sparsified.m:286-334): X*(1+2eps) -> sign flip -> zero-pad to p2 -> FWHT/sqrt(p2)
-> keep s = max(1, round(gamma*p2)) uniformly random rows per column, ascending, scaled by
p2/s (private/randsample_fixedNumberEntries.m:30-31,62).

The RNG is numpy's / torch's, not MATLAB's: RNG-dependent products (sign vector, sampled
rows, initial centres) are *inputs* to the parity tests, never compared with MATLAB's.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

EPS = np.finfo(np.float64).eps


def small_p_of(gamma: float, p2: int) -> int:
    """kmeans_sparsified.m:324-326: small_p = max(1, round(SparsityLevel*p2)) (MATLAB round: half away from 0)."""
    return max(1, int(np.floor(gamma * p2 + 0.5)))


def gmm_dense(p: int, n: int, K: int, seed: int = 234, noise: float = 0.1):
    """Dense p x n mixture as in example_sparseKMeans.m:17-22; returns (X, true_centres, labels)."""
    rng = np.random.default_rng(seed)
    centres = rng.standard_normal((p, K))
    labels = (np.arange(n) * K) // n
    X = centres[:, labels] + noise * rng.standard_normal((p, n))
    return X, centres, labels


def sample_rows(rng: np.random.Generator, p2: int, s: int, n: int) -> np.ndarray:
    """s distinct uniformly random rows per column, ascending: [n, s] int array
    (private/randsample_block.m:44-92 -- any exact without-replacement sampler is equivalent)."""
    keys = rng.random((n, p2))
    idx = np.argpartition(keys, s - 1, axis=1)[:, :s]
    idx.sort(axis=1)
    return idx


def sparsify_dense(Xmixed: np.ndarray, s: int, rng: np.random.Generator) -> sp.csc_matrix:
    """randsample_fixedNumberEntries(X, s): p2 x n CSC with exactly s entries per column
    (fewer where a sampled value is exactly 0: MATLAB's sparse() drops zeros, :62), values
    X(ind)/(s/p2) -- a true division by SparsityLevel = s/p2 (:30-31,62)."""
    p2, n = Xmixed.shape
    rows = sample_rows(rng, p2, s, n)
    level = np.float64(s) / np.float64(p2)
    vals = Xmixed[rows, np.arange(n)[:, None]] / level
    indptr = np.arange(0, (n + 1) * s, s, dtype=np.int64)
    Y = sp.csc_matrix((vals.ravel(), rows.ravel().astype(np.int64), indptr), shape=(p2, n))
    Y.eliminate_zeros()
    Y.sort_indices()
    return Y


def sparsified_gmm_host(p: int, n: int, K: int, gamma: float, seed: int = 234, fwht=None):
    """Small-case pipeline on the host (numpy): returns dict with the CSC matrix and everything
    a Lloyd run needs.  ``fwht`` is the transform to use (callable m x n -> m x n); tests pass the
    oracle's or the HIP one."""
    X, centres, labels = gmm_dense(p, n, K, seed)
    rng = np.random.default_rng(seed + 1)
    p2 = 1 << int(np.ceil(np.log2(p))) if p > 1 else 2
    d = np.sign(rng.standard_normal(p2))
    d[d == 0] = 1.0
    Xs = X * (1.0 + 2.0 * EPS)
    Xp = np.zeros((p2, n))
    Xp[:p] = Xs
    Xp *= d[:, None]
    Xm = fwht(Xp) / np.sqrt(np.float64(p2))
    s = small_p_of(gamma, p2)
    Y = sparsify_dense(Xm, s, rng)
    gamma_used = s / p  # kmeans_sparsified.m:329 (divides by p, not p2)
    return dict(Y=Y, X=X, Xmixed=Xm, d=d, p=p, p2=p2, s=s, gamma=gamma_used, labels=labels, centres=centres)


# ---------------------------------------------------------------------------------------------
# device-side generator for bench-scale workloads (torch only for RNG / sort; the transform is ours)
# ---------------------------------------------------------------------------------------------
def sparsified_gmm_device(ctx, p: int, n_local: int, n_total: int, first: int, K: int, gamma: float,
                          seed: int = 234, chunk: int = 65536, noise: float = 0.1, order: str = "block",
                          layout: str = "csc"):
    """Generates points [first, first+n_local) of the n_total-point mixture directly into device
    CSC arrays (jc int64, ir int16/int32, x float64), chunk by chunk, never holding more than
    ``chunk`` dense columns.  Every rank generates the same global dataset: cluster means and the
    sign vector come from ``seed``; chunk c uses generator seed ^ (c+1) regardless of which rank
    owns it (chunks are aligned to global multiples of ``chunk``).
    order = "block": point i belongs to cluster floor(i*K/n) -- contiguous equal blocks, as in
    example_sparseKMeans.m:19-22; "shuffled": every point draws its cluster uniformly at random (data in
    arbitrary order: consecutive points have nothing to do with each other).
    layout = "records": the sparsifier writes the library's record layout directly (``rec`` uint8, for
    Shard.from_records) instead of the CSC arrays -- same samples, same values; the entries then exist once."""
    import torch

    from .engine import mix_sample_device, mix_sample_records_device, record_bytes

    dev = torch.device("cuda", ctx.device)
    p2 = 1 << int(np.ceil(np.log2(p))) if p > 1 else 2
    s = small_p_of(gamma, p2)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(seed)
    means = torch.randn((K, p), generator=g0, device=dev, dtype=torch.float64)
    sign = torch.sign(torch.randn(p2, generator=g0, device=dev, dtype=torch.float64))
    sign[sign == 0] = 1.0
    ir_dtype = torch.int16 if p2 <= 65536 else torch.int32   # int16 storage is read as uint16 row ids
    # 48 entries of slack: the fixed-stride kernels read (and ignore) up to 33 entries past a column
    records = layout == "records" and s <= 64      # (the record layout holds columns of at most 64 entries: CSC beyond, as StreamingSparsifier)
    R = record_bytes(s, 16 if p2 <= 65536 else 32) if records else 0
    if records:
        rec = torch.empty(n_local * R + 256, dtype=torch.uint8, device=dev)
        x = ir = None
    else:
        x = torch.zeros(n_local * s + 48, dtype=torch.float64, device=dev)
        ir = torch.zeros(n_local * s + 48, dtype=ir_dtype, device=dev)
    premul = float(1.0 + 2.0 * EPS)
    postdiv = float(np.sqrt(np.float64(p2)))
    last = first + n_local
    c = first // chunk
    while c * chunk < last:
        lo, hi = max(first, c * chunk), min(last, (c + 1) * chunk)
        gen = torch.Generator(device=dev)
        gen.manual_seed((seed * 1000003) ^ (c + 1))
        # generate the whole aligned chunk so the stream is rank-independent, then slice
        m = min((c + 1) * chunk, n_total) - c * chunk
        ids = torch.arange(c * chunk, c * chunk + m, device=dev)
        if order == "shuffled":
            lab = torch.randint(0, K, (m,), generator=gen, device=dev)
        else:
            lab = (ids * K) // n_total
        dense = means[lab] + noise * torch.randn((m, p), generator=gen, device=dev, dtype=torch.float32).double()
        a, b = lo - c * chunk, hi - c * chunk
        dense = dense[a:b].contiguous()
        o = (lo - first) * s
        # the product's own device sparsifier: mix -> sample (Philox keyed by the GLOBAL point index) -> CSC
        if records:
            mix_sample_records_device(ctx, dense, p2, sign, premul, postdiv, s, seed, lo, rec[(lo - first) * R:],
                                      16 if p2 <= 65536 else 32)
        else:
            mix_sample_device(ctx, dense, p2, sign, premul, postdiv, s, seed, lo, ir[o:], x[o:])
        c += 1
    if records:
        return dict(rec=rec, R=R, n=n_local, nnz=n_local * s, p2=p2, s=s, gamma=s / p, sign=sign, means=means,
                    ir_bits=16 if p2 <= 65536 else 32)
    jc = torch.arange(0, (n_local + 1) * s, s, dtype=torch.int64, device=dev)
    return dict(jc=jc, ir=ir, x=x, nnz=n_local * s, p2=p2, s=s, gamma=s / p, sign=sign, means=means)


def streamed_pixel_dataset(ctx, p: int, n_local: int, first: int, K: int, gamma: float, seed: int = 234,
                           chunk: int = 131072, pool_points: int = 1 << 20, sign=None, layout: str = "csc"):
    """Config-5-shaped ingest (BASELINE.json: "1e9 x 784 chunk-streamed from host DRAM, K=10"; shape of
    private/sampleAndMixFromLargeFile.m:79-129): 8-bit "pixel" points held in PINNED host memory are streamed chunk
    by chunk over PCIe through engine.StreamingSparsifier (copy stream + two staging buffers -> widen -> mix -> sample
    -> resident sparse shard).  The host pool holds ``pool_points`` mixture points (K cluster means in [48, 208],
    noise sigma 10, rounded and clipped to uint8) and is cycled: point i of the dataset is pool point i mod pool_points
    -- its SAMPLE depends on the global index i, so repeated pool points still give distinct sparse columns.
    Returns the dict of sparsified_gmm_device plus ``labels_pool`` (cluster of each pool point), ``pool`` (the pinned
    uint8 tensor) and ``ingest`` = dict(points, bytes, seconds, GBs)."""
    import time

    import torch

    from .engine import StreamingSparsifier

    dev = torch.device("cuda", ctx.device)
    p2 = 1 << int(np.ceil(np.log2(p))) if p > 1 else 2
    s = small_p_of(gamma, p2)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(seed)
    means = 48.0 + 160.0 * torch.rand((K, p), generator=g0, device=dev, dtype=torch.float64)
    if sign is None:
        sign = torch.sign(torch.randn(p2, generator=g0, device=dev, dtype=torch.float64))
        sign[sign == 0] = 1.0
    pool_points = int(min(pool_points, max(n_local, 1)))
    pool = torch.empty((pool_points, p), dtype=torch.uint8, pin_memory=True)
    labels_pool = torch.empty(pool_points, dtype=torch.int64)
    for c0 in range(0, pool_points, chunk):                      # generated on the device, parked in pinned host memory
        m = min(chunk, pool_points - c0)
        gen = torch.Generator(device=dev)
        gen.manual_seed((seed * 1000003) ^ (c0 + 1))
        lab = torch.randint(0, K, (m,), generator=gen, device=dev)
        px = means[lab] + 10.0 * torch.randn((m, p), generator=gen, device=dev, dtype=torch.float32).double()
        pool[c0:c0 + m].copy_(px.round().clamp_(0, 255).to(torch.uint8))
        labels_pool[c0:c0 + m].copy_(lab)
    torch.cuda.synchronize()
    sp_ = StreamingSparsifier(ctx, p, n_local, s, seed, sign, first=first, layout=layout)
    t0 = time.perf_counter()
    done = 0
    while done < n_local:
        o = (first + done) % pool_points                         # pool position of the next point
        m = min(chunk, n_local - done, pool_points - o)
        sp_.append(pool[o:o + m])                                # a pinned slice: sent from where it lies
        done += m
    shard_arrays = (sp_.ir, sp_.x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    common = dict(nnz=n_local * s, p2=p2, s=s, gamma=s / p, sign=sign, means=means, pool=pool, labels_pool=labels_pool,
                  pool_points=pool_points, ingest=dict(points=n_local, bytes=sp_.bytes_in, seconds=dt, GBs=sp_.bytes_in / dt / 1e9))
    if sp_.records:
        return dict(rec=sp_.rec, R=sp_.R, n=n_local, ir_bits=sp_.ir_bits, **common)
    jc = torch.arange(0, (n_local + 1) * s, s, dtype=torch.int64, device=dev)
    return dict(jc=jc, ir=shard_arrays[0], x=shard_arrays[1], **common)
