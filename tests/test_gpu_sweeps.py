"""-m gpu: seeded random-shape sweeps over the remaining entry points (the sweep over the Lloyd front half lives in
test_gpu_screen.py).  Shapes are drawn, not hand-picked: odd sizes, tails and boundaries of the launch
geometry get visited without anyone having thought of them."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from util import parts, random_csc, sample_rows_reference, set_switch

pytestmark = pytest.mark.gpu

# SPKM_SWEEP=n multiplies the number of seeds of the random-shape sweeps (bug hunting; default 1)
import os as _os
_SW = max(1, int(_os.environ.get("SPKM_SWEEP", "1")))


@pytest.mark.parametrize("seed", range(24 * _SW))
def test_assign_step_random_shapes(gpu_ctx, oracle, seed):
    """spkm_assign_dev (exact kernels): ragged or fixed columns, any K, gamma present or empty."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    rng = np.random.default_rng(2000 + seed)
    p = int(rng.choice([3, 17, 64, 100, 255, 256, 777, 1024, 1500, 3000]))
    K = int(rng.integers(1, 90))
    n = int(rng.integers(1, 3001))
    s = int(rng.integers(1, min(p, 70) + 1))
    ragged = bool(seed % 2)
    gamma = float(rng.choice([0.0, 0.03, 0.5]))
    X = random_csc(p, n, s, seed=seed, ragged=ragged, empty_cols=(0,) if ragged else ())
    Cm = rng.standard_normal((p, K)) * (gamma if gamma else 1.0)
    eng = LloydEngine(Shard.from_scipy(gpu_ctx, X), K, gamma if gamma else 1.0, unbiased=bool(gamma))
    eng.assign_step(torch.tensor(np.ascontiguousarray(Cm.T), device="cuda"))
    torch.cuda.synchronize()
    ra, rd = oracle.assign(p, n, *parts(X), Cm, gamma)
    assert np.array_equal(eng.assign.cpu().numpy(), ra)
    assert np.array_equal(eng.mind.cpu().numpy(), rd)
    eng.accumulate_step()
    torch.cuda.synchronize()
    S, Cnt, nk = oracle.accumulate(p, n, K, *parts(X), ra)
    red = eng.reduce.cpu().numpy()
    pk = p * K
    assert np.array_equal(red[pk:2 * pk].reshape(K, p).T, Cnt)
    assert np.array_equal(red[2 * pk:2 * pk + K], nk.astype(float))
    assert np.abs(red[:pk].reshape(K, p).T - S).max() <= 1e-12 * max(np.abs(S).max(), 1e-300)


@pytest.mark.parametrize("seed", range(12 * _SW))
def test_sparse_centres_random_shapes(gpu_ctx, oracle, seed):
    """spkm_assign_sparse_centers_dev (findClusterAssignments.m:63-75) against the oracle."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    rng = np.random.default_rng(3000 + seed)
    p = int(rng.choice([32, 100, 256, 512, 1024]))
    K = int(rng.integers(1, 40))
    n = int(rng.integers(1, 2001))
    s = int(rng.integers(1, min(p, 60) + 1))
    gamma = float(rng.choice([0.0, 0.05]))
    X = random_csc(p, n, s, seed=seed + 50, ragged=bool(seed % 2))
    Cs = random_csc(p, K, max(1, s), seed=seed + 99)             # sparse centres: columns of a sparse matrix
    eng = LloydEngine(Shard.from_scipy(gpu_ctx, X), K, gamma if gamma else 1.0, unbiased=bool(gamma))
    Cd = np.ascontiguousarray(Cs.toarray().T)
    M = np.ascontiguousarray((Cs != 0).toarray().T.astype(np.uint8))
    eng.assign_sparse_step(torch.tensor(Cd, device="cuda"), torch.tensor(M, device="cuda"))
    torch.cuda.synchronize()
    Cc = sp.csc_matrix(Cs)
    dist = oracle.dist_sparse_centers(p, n, *parts(X), *parts(Cc), K, gamma)
    d, a = oracle.min_cols(dist)
    assert np.array_equal(eng.assign.cpu().numpy(), a)
    assert np.array_equal(eng.mind.cpu().numpy(), d)


@pytest.mark.parametrize("seed", range(12 * _SW))
def test_mex_operators_random_shapes(gpu_ctx, oracle, seed):
    """The stand-alone operators through their host entry points, ragged inputs."""
    from sparsifiedkmeans_amd import ops
    rng = np.random.default_rng(4000 + seed)
    p = int(rng.choice([2, 9, 64, 300, 1024, 5000]))
    n = int(rng.integers(1, 1500))
    K = int(rng.choice([1, 2, 3, 4, 7, 20]))
    X = random_csc(p, n, int(rng.integers(1, min(p, 40) + 1)), seed=seed + 7, ragged=True)
    Cm = rng.standard_normal((p, K))
    assert np.array_equal(ops.SparseMatrixMinusCluster(X, Cm, ctx=gpu_ctx), oracle.dist_csc(p, n, *parts(X), Cm))
    c = rng.standard_normal(p)
    ip, nx2 = ops.SparseMatrixInnerProduct(X, c, ctx=gpu_ctx)
    rip, rnx = oracle.innerprod_csc(n, *parts(X), c)
    assert np.array_equal(ip, rip) and np.array_equal(nx2, rnx)
    assert np.array_equal(ops.SparseMatrixColumnNormSq(X, ctx=gpu_ctx), oracle.colnormsq_csc(n, X.indptr.astype(np.uint64), X.data))
    beta = float(rng.standard_normal())
    # |beta| > 1 can make the quadratic negative: sqrt gives NaN in both (the payload / sign of a NaN is not compared)
    assert np.array_equal(ops.SparseMatrixMinusCluster(X, Cm[:, :1], beta, ctx=gpu_ctx).ravel(),
                          oracle.dist_csc_beta(n, *parts(X), Cm[:, 0], beta), equal_nan=True)


@pytest.mark.parametrize("seed", range(12 * _SW))
def test_sparsifier_random_shapes(gpu_ctx, oracle, seed):
    """spkm_mix_sample_dev: any p (zero-padded to p2), any s <= p2, any column offset."""
    from sparsifiedkmeans_amd.engine import mix_sample_device
    rng = np.random.default_rng(5000 + seed)
    p = int(rng.integers(9, 3000))
    p2 = 1 << int(np.ceil(np.log2(p)))
    s = int(rng.integers(1, min(p2, 200) + 1))
    n = int(rng.integers(1, 700))
    col0 = int(rng.integers(0, 2**40))
    sd = int(rng.integers(0, 2**62))
    X = rng.standard_normal((p, n))
    d = np.sign(rng.standard_normal(p2))
    ir = torch.zeros(n * s + 16, dtype=torch.int16, device="cuda")
    xv = torch.zeros(n * s + 16, dtype=torch.float64, device="cuda")
    mix_sample_device(gpu_ctx, torch.tensor(np.ascontiguousarray(X.T), device="cuda"), p2, torch.tensor(d, device="cuda"),
                      1.0 + 2 * np.finfo(float).eps, float(np.sqrt(np.float64(p2))), s, sd, col0, ir, xv)
    torch.cuda.synchronize()
    rows = ir[: n * s].cpu().numpy().view(np.uint16).astype(np.int64).reshape(n, s)
    assert np.array_equal(rows, sample_rows_reference(sd, col0, n, p2, s))
    Xm = oracle.mix(X, d, p2)
    want = Xm[rows, np.arange(n)[:, None]] / (np.float64(s) / np.float64(p2))
    assert np.array_equal(xv[: n * s].cpu().numpy().reshape(n, s), want)


@pytest.mark.parametrize("seed", range(10 * _SW))
def test_fwht_random_shapes(gpu_ctx, oracle, seed):
    from sparsifiedkmeans_amd import ops
    rng = np.random.default_rng(6000 + seed)
    m = 1 << int(rng.integers(1, 16))
    n = int(rng.integers(1, max(2, 200000 // m)))
    x = rng.standard_normal((m, n))
    assert np.array_equal(ops.hadamard(x, ctx=gpu_ctx), oracle.fwht(x))


@pytest.mark.parametrize("seed", range(10 * _SW))
def test_dense_assign_random_shapes(gpu_ctx, seed):
    """spkm_dense_assign_dev / spkm_dense_accumulate_dev on drawn shapes (tails of the 64 x 128 x 64 MFMA tiling)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import numpy_ref
    from sparsifiedkmeans_amd.engine import dense_accumulate_device, dense_assign_device
    rng = np.random.default_rng(7000 + seed)
    p = int(rng.integers(1, 1300))
    n = int(rng.integers(1, 3000))
    K = int(rng.integers(1, 300))
    C = rng.standard_normal((p, K))
    X = C[:, rng.integers(0, K, n)] + 0.4 * rng.standard_normal((p, n))
    xd = torch.tensor(np.ascontiguousarray(X.T), device="cuda")
    a, d = dense_assign_device(gpu_ctx, xd, torch.tensor(np.ascontiguousarray(C.T), device="cuda"))
    a, d = a.cpu().numpy(), d.cpu().numpy()
    a0, d0, full = numpy_ref.dense_assign(X, C)
    scale = np.sum(X * X, axis=0) + np.max(np.sum(C * C, axis=0))
    tol2 = 64 * np.finfo(np.float64).eps * scale
    assert np.all(np.abs(d * d - d0 * d0) <= tol2 + 1e-300)
    chosen = full[a, np.arange(n)]
    assert np.all(chosen * chosen - d0 * d0 <= 4 * tol2)
    sums = torch.zeros((K, p), dtype=torch.float64, device="cuda")
    cnt = torch.zeros(K, dtype=torch.float64, device="cuda")
    dense_accumulate_device(gpu_ctx, xd, torch.tensor(a0.astype(np.int32), device="cuda"), sums, cnt)
    ref = np.zeros((K, p))
    np.add.at(ref, a0, X.T)
    assert np.array_equal(cnt.cpu().numpy(), np.bincount(a0, minlength=K).astype(np.float64))
    assert np.allclose(sums.cpu().numpy(), ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("seed", range(10 * _SW))
def test_driver_random_options(gpu_ctx, seed):
    """kmeans_sparsified with drawn options: the outputs are mutually consistent whatever the combination."""
    import warnings
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.kmeans import kmeans_sparsified
    rng = np.random.default_rng(8000 + seed)
    p = int(rng.choice([50, 64, 100, 128, 300]))
    K = int(rng.integers(2, 9))
    n = int(rng.integers(200, 3000))
    X, centres, labels = synth.gmm_dense(p, n, K, seed=seed)
    cs = bool(rng.integers(0, 2))
    opts = dict(Sparsify=True, SparsityLevel=float(rng.choice([0.05, 0.2, 0.5, 1.0])), SketchType="Hadamard",
                Start=str(rng.choice(["sample", "Arthur", "uniform"])), Replicates=int(rng.integers(1, 4)),
                EmptyAction=str(rng.choice(["singleton", "drop"])), ColumnSamples=cs, MaxIter=int(rng.integers(1, 40)),
                denseCenters=bool(rng.integers(0, 2)), unbiasedDistance=bool(rng.integers(0, 2)),
                unbiasedInitialization=bool(rng.integers(0, 2)), MB_limit=float(rng.choice([0.05, 1.0, 500.0])),
                rng=int(seed), nargout=int(rng.choice([5, 6, 7, 8, 9])))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = kmeans_sparsified(X if cs else X.T, K, **opts)
    IDX, C, SUMD, D, OUT = out[:5]
    Kb = C.shape[1] if cs else C.shape[0]
    assert 1 <= Kb <= K and (C.shape[0] if cs else C.shape[1]) == p
    assert np.all(np.isfinite(C)) and np.all(np.isfinite(D)) and np.all(D >= 0)
    if IDX.size:                                             # 'drop' in the last iteration leaves IDX empty (reference quirk)
        assert IDX.shape == (n,) and IDX.min() >= 1 and IDX.max() <= Kb
        if opts["Replicates"] == 1:
            assert np.allclose(SUMD, [np.sum(D[IDX == k + 1] ** 2) for k in range(Kb)], rtol=1e-12, atol=1e-300)
    assert OUT["objectives"].shape == (opts["Replicates"],) and np.all(OUT["iterations"] <= opts["MaxIter"])
    if len(out) > 5 and IDX.size:
        Cp = C if cs else C.T
        C2 = out[5] if cs else out[5].T
        assert C2.shape == (p, Kb)
        want = np.stack([X[:, IDX == k + 1].mean(axis=1) if np.any(IDX == k + 1) else np.zeros(p) for k in range(Kb)], axis=1)
        assert np.allclose(C2, want, rtol=1e-11, atol=1e-11)
        if len(out) > 7:
            IDX2, D2 = out[6], out[7]
            direct = np.sqrt(((X[:, None, :] - Cp[:, :, None]) ** 2).sum(axis=0))      # Kb x n
            assert np.allclose(D2, direct.min(axis=0), rtol=1e-6, atol=1e-6)
            assert np.allclose(direct[IDX2 - 1, np.arange(n)], direct.min(axis=0), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("seed", range(16 * _SW))
def test_screen_equals_exact_kernels_midsize(gpu_ctx, seed, monkeypatch):
    """1e5 .. 6e5 points (many chunks per workgroup, ragged last chunk, every tile / round variant by chance):
    the screen path against the all-exact kernels, every point, bit for bit.  No CPU oracle at this size."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    rng = np.random.default_rng(9000 + seed)
    p = int(rng.choice([64, 128, 256, 512, 1000, 1024]))
    s = int(rng.integers(1, min(64, p) + 1))
    K = int(rng.integers(2, 200))
    n = int(rng.integers(100_000, 600_001))
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    # fixed-stride CSC on the device: s distinct ascending rows per column
    keys = torch.rand((n, p), generator=g, device="cuda")
    rows = torch.topk(keys, s, dim=1, largest=False).indices.sort(dim=1).values.to(torch.int16)
    del keys
    lab = torch.randint(0, K, (n,), generator=g, device="cuda")
    C = torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
    vals = C[lab[:, None], rows.long()] * (p / s) + 0.5 * torch.randn((n, s), generator=g, device="cuda", dtype=torch.float64)
    pad = 48
    ir = torch.zeros(n * s + pad, dtype=torch.int16, device="cuda")
    xv = torch.zeros(n * s + pad, dtype=torch.float64, device="cuda")
    ir[: n * s] = rows.reshape(-1)
    xv[: n * s] = vals.reshape(-1)
    jc = torch.arange(0, (n + 1) * s, s, dtype=torch.int64, device="cuda")
    shard = Shard.from_device(gpu_ctx, p, jc, ir, xv, nnz=n * s)
    gamma = s / p
    centers = (C * gamma * (p / s)).contiguous()               # centres / gamma is comparable to the values
    e1 = LloydEngine(shard, K, gamma)
    e1.assign_accumulate_step(centers)
    torch.cuda.synchronize()
    assert e1.last_path_info()[0] == 1
    set_switch(monkeypatch, gpu_ctx, "SPKM_NO_SCREEN")
    e0 = LloydEngine(shard, K, gamma)
    e0.assign_accumulate_step(centers)
    torch.cuda.synchronize()
    assert e0.last_path_info()[0] == 0
    assert torch.equal(e1.assign, e0.assign)
    assert torch.equal(e1.mind, e0.mind)
    pk = p * K
    assert torch.equal(e1.reduce[pk:2 * pk + K], e0.reduce[pk:2 * pk + K])
    scale = float(e0.reduce[:pk].abs().max().item())
    assert float((e1.reduce[:pk] - e0.reduce[:pk]).abs().max().item()) <= 1e-11 * max(scale, 1e-300)


def test_adaptive_policy_soak(gpu_ctx, monkeypatch):
    """60 consecutive fused calls on one shard while the centres wander between converged, slightly perturbed and
    scrambled states: the library moves between the two-phase screen, the plain screen and the exact kernels on its
    own; every call's assignments and min-distances equal those of the all-exact kernels on the same centres."""
    import os
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device
    n, p, K = 300_000, 512, 60
    d = synth.sparsified_gmm_device(gpu_ctx, p, n, n, 0, K, 0.1, seed=77)
    shard = Shard.from_device(gpu_ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
    planted = mix_device(gpu_ctx, d["means"].contiguous(), d["p2"], d["sign"], 1.0, float(np.sqrt(np.float64(d["p2"]))))
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    eng = LloydEngine(shard, K, d["gamma"])
    # the all-exact reference runs on a second shard object over the same device buffers: an exact call on the
    # same shard would make the library forget the bounds it carries between screen calls
    twin = Shard.from_device(gpu_ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
    ref = LloydEngine(twin, K, d["gamma"])
    modes = set()
    skipped = 0
    for it in range(60):
        phase = (it // 6) % 4
        if phase in (0, 1):
            c = planted + 0.01 * torch.randn(planted.shape, generator=g, device="cuda", dtype=torch.float64)
        elif phase == 2:
            c = planted[torch.randint(0, K, (K,), generator=g, device="cuda")] + 0.05 * torch.randn(
                planted.shape, generator=g, device="cuda", dtype=torch.float64)       # duplicates, uncovered clusters
        else:
            c = torch.randn(planted.shape, generator=g, device="cuda", dtype=torch.float64) * 0.05   # everything ambiguous
        c = c.contiguous()
        set_switch(None, gpu_ctx, "SPKM_NO_SCREEN", False)
        eng.assign_accumulate_step(c)
        torch.cuda.synchronize()
        path, listed = eng.last_path_info()
        ra, rn = eng.last_screen_rounds()
        modes.add("exact" if path == 0 else ("two-phase" if ra < rn else "plain"))
        skipped += eng.last_screen_mode()[4]
        set_switch(None, gpu_ctx, "SPKM_NO_SCREEN")
        ref.assign_accumulate_step(c)
        torch.cuda.synchronize()
        set_switch(None, gpu_ctx, "SPKM_NO_SCREEN", False)
        assert torch.equal(eng.assign, ref.assign), f"call {it}"
        assert torch.equal(eng.mind, ref.mind), f"call {it}"
    assert "two-phase" in modes and "plain" in modes           # the policy really moved between modes
    assert skipped > 0                                          # and the carried bounds skipped steps on the way


@pytest.mark.parametrize("seed", range(12 * _SW))
def test_lloyd_runs_random_shapes_equal_oracle_every_iteration(gpu_ctx, oracle, seed, monkeypatch):
    """Short Lloyd runs on drawn shapes (n not a multiple of 16 or 64, K with narrow / carried remainders, 1..64
    entries per column): every iteration's assignments and min-distances equal the oracle's for the centres that went
    in, while the library moves through its forms (plain / two-phase / hinted screen, carried bounds) on its own."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard
    rng = np.random.default_rng(9000 + seed)
    p = int(rng.choice([64, 128, 256, 500, 1024]))
    s_ = int(rng.integers(1, min(p, 64) + 1))
    K = int(rng.choice([2, 3, 17, 33, 36, 40, 64, 68, 100]))
    n = int(rng.integers(200, 3000))
    X = random_csc(p, n, s_, seed=seed)
    # clustered values so that runs converge: shift every column by one of K centres' entries
    lab = (np.arange(n) * K) // n
    cen0 = rng.standard_normal((p, K)) * 2.0
    X = X.tocsc()
    for i in range(n):
        sl = slice(X.indptr[i], X.indptr[i + 1])
        X.data[sl] = 0.3 * X.data[sl] + cen0[X.indices[sl], lab[i]]
    gam = s_ / p
    C = gam * cen0 + 0.05 * rng.standard_normal((p, K))        # near the ML-scaled planted centres
    if seed % 3 == 0:
        C[:, 1 % K] = C[:, 0]                                   # a duplicate centre: ties
    eng = LloydEngine(Shard.from_scipy(gpu_ctx, X), K, gam)
    cd = torch.tensor(np.ascontiguousarray(C.T), device="cuda")
    forms = set()
    for it in range(7):
        cin = cd.cpu().numpy().T.copy()
        eng.iterate(cd)
        torch.cuda.synchronize()
        forms.add(eng.last_screen_mode()[0])
        ra, rd = oracle.assign(p, n, *parts(X), cin, gam)
        assert np.array_equal(eng.assign.cpu().numpy(), ra), f"iteration {it}"
        assert np.array_equal(eng.mind.cpu().numpy(), rd), f"iteration {it}"
