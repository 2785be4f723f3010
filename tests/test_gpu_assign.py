"""-m gpu: fused assignment (distance + sqrt + argmin) through the C ABI vs the oracle:
assignments bit-exact (the north-star bar) and min distances bit-exact."""
import numpy as np
import pytest
import torch

from util import parts, random_csc

pytestmark = pytest.mark.gpu


def run_assign(ctx, X, Cm, gamma):
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n = X.shape
    K = Cm.shape[1]
    eng = LloydEngine(Shard.from_scipy(ctx, X), K, gamma if gamma else 1.0, unbiased=bool(gamma))
    centers = torch.tensor(np.ascontiguousarray(Cm.T), device=f"cuda:{ctx.device}")
    eng.assign_step(centers)
    torch.cuda.synchronize()
    return eng.assign.cpu().numpy(), eng.mind.cpu().numpy(), eng.stats.cpu().numpy(), eng.nk.cpu().numpy()


CASES = [  # p, n, K, nnz/col, ragged, gamma   (tile geometry exercised)
    (512, 5000, 5, 26, False, 26 / 512),      # config 1 shape: KT=16, one tile
    (512, 10000, 3, 26, False, 26 / 512),     # config 1' shape
    (1024, 20000, 100, 51, False, 51 / 1024), # config 2/4 shape per point: KT=16, 7 tiles
    (1024, 6000, 10, 51, False, 51 / 784),    # config 3 shape (784 -> 1024)
    (1024, 3001, 100, 40, True, 0.05),        # ragged + empty columns, tails of every length
    (256, 4000, 64, 13, True, 0.0),           # KT=64, gamma empty
    (300, 4000, 40, 15, True, 0.1),           # KT=32/16 choice, p not a power of two
    (64, 1000, 1, 5, True, 0.2),              # K=1
    (64, 1000, 2, 5, True, 0.2),
    (2048, 700, 33, 60, True, 0.03),          # tile does not fit LDS: generic path
    (70000, 300, 4, 50, True, 0.001),         # 32-bit row ids
]


@pytest.mark.parametrize("p,n,K,s,ragged,gamma", CASES)
def test_assign_bit_exact(gpu_ctx, oracle, p, n, K, s, ragged, gamma):
    X = random_csc(p, n, s, seed=p * 7 + K, ragged=ragged, empty_cols=(1, n - 1) if ragged else ())
    Cm = np.random.default_rng(K + 1).standard_normal((p, K)) * 3.0 * (gamma if gamma else 1.0)
    a, d, stats, nk = run_assign(gpu_ctx, X, Cm, gamma)
    ra, rd = oracle.assign(p, n, *parts(X), Cm, gamma)
    assert np.array_equal(a, ra)
    assert np.array_equal(d, rd)
    assert np.array_equal(nk, np.bincount(ra, minlength=K))
    assert stats[1] == rd.max() and int(stats[2]) == int(np.argmax(rd))
    assert abs(stats[0] - np.sum(rd * rd)) <= 1e-12 * max(1.0, np.sum(rd * rd))


def test_fused_paths_equal_min_of_the_reference_s_own_distances(gpu_ctx, oracle):
    """Every assignment path of the engine against tests/golden/ref_dist_*.npz -- distances produced by the REFERENCE's
    own `switch (K)` (SparseMatrixMinusCluster.c:131-183, compiled from /root/reference by oracle/Makefile) -- followed by
    MATLAB's `min` (findClusterAssignments.m:169: first index on ties; the fixtures hold duplicate centroids).  The exact
    kernel (`assign_step`) and the fused certified-screen call (`assign_accumulate_step`; the p = 1024 fixtures have
    fixed-stride columns of 51 entries, the benchmark's point shape) must both give those indices and those minima."""
    import os

    import scipy.sparse as sp

    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    gold = os.path.join(os.path.dirname(__file__), "golden")
    files = sorted(f for f in os.listdir(gold) if f.startswith("ref_dist_"))
    assert len(files) == 8
    for f in files:
        z = np.load(os.path.join(gold, f))
        p, n, K = int(z["p"]), int(z["n"]), int(z["K"])
        X = sp.csc_matrix((z["x"], z["ir"].astype(np.int64), z["jc"].astype(np.int64)), shape=(p, n))
        want_d, want_a = oracle.min_cols(z["dist"])
        a, d, _, nk = run_assign(gpu_ctx, X, z["C"], 0.0)
        assert np.array_equal(a, want_a) and np.array_equal(d, want_d), f
        eng = LloydEngine(Shard.from_scipy(gpu_ctx, X), K, 1.0, unbiased=False)
        eng.assign_accumulate_step(torch.tensor(np.ascontiguousarray(z["C"].T), device="cuda:0"))
        assert np.array_equal(eng.assign.cpu().numpy(), want_a), f
        assert np.array_equal(eng.mind.cpu().numpy(), want_d), f
        assert np.array_equal(eng.nk.cpu().numpy(), np.bincount(want_a, minlength=K)), f


def test_ties_take_first_index(gpu_ctx, oracle):
    """Duplicate centroids give exactly equal distances: MATLAB's min keeps the first
    (findClusterAssignments.m:169); empty columns give all-zero distances -> index 0."""
    p, n, K = 512, 3000, 37
    X = random_csc(p, n, 20, seed=3, ragged=True, empty_cols=(0, 5, 77))
    Cm = np.random.default_rng(0).standard_normal((p, K))
    Cm[:, 20] = Cm[:, 3]     # same tile
    Cm[:, 36] = Cm[:, 3]     # different tile
    Cm[:, 17] = Cm[:, 16]
    a, d, _, _ = run_assign(gpu_ctx, X, Cm, 0.0)
    ra, rd = oracle.assign(p, n, *parts(X), Cm, 0.0)
    assert np.array_equal(a, ra) and np.array_equal(d, rd)
    assert not np.any(np.isin(a, [20, 36, 17]))
    assert a[0] == 0 and a[5] == 0 and d[77] == 0.0


def test_near_ties_sqrt_collapse(gpu_ctx, oracle):
    """Squared distances a few ulp apart can round to the same sqrt; the reference compares the
    sqrt values, so the lower index wins even when its squared distance is (slightly) larger."""
    p, K = 64, 16
    rng = np.random.default_rng(9)
    n = 4000
    X = random_csc(p, n, 1, seed=1)           # one entry per column: acc = (x - c)^2 exactly one term
    Cm = np.zeros((p, K))
    base = rng.standard_normal(p)
    for k in range(K):                         # centroids that differ by ~1 ulp per row
        Cm[:, k] = np.nextafter(base, base + (1 if k % 2 else -1), dtype=np.float64) if k else base
    for k in range(2, K):
        Cm[:, k] = Cm[:, k % 2] + (k // 2) * np.spacing(Cm[:, k % 2])
    a, d, _, _ = run_assign(gpu_ctx, X, Cm, 0.0)
    ra, rd = oracle.assign(p, n, *parts(X), Cm, 0.0)
    assert np.array_equal(a, ra) and np.array_equal(d, rd)


@pytest.mark.parametrize("p,n,K,s", [(64, 3000, 1000, 8), (256, 2000, 513, 13)])
def test_many_centroids(gpu_ctx, oracle, p, n, K, s):
    """K far above one tile: 63 exact tiles / 32 screen tiles on 256 workgroups."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    X = random_csc(p, n, s, seed=K)
    Cm = np.random.default_rng(K).standard_normal((p, K))
    a, d, _, nk = run_assign(gpu_ctx, X, Cm, s / p)
    ra, rd = oracle.assign(p, n, *parts(X), Cm, s / p)
    assert np.array_equal(a, ra) and np.array_equal(d, rd)
    eng = LloydEngine(Shard.from_scipy(gpu_ctx, X), K, s / p)
    eng.assign_accumulate_step(torch.tensor(np.ascontiguousarray(Cm.T), device="cuda:0"))
    assert np.array_equal(eng.assign.cpu().numpy(), ra) and np.array_equal(eng.mind.cpu().numpy(), rd)
