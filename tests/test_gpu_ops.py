"""-m gpu: the five mex-equivalent operators, through the C ABI, bit-exact against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from util import parts, random_csc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p,n,K", [(2, 1, 1), (64, 257, 2), (64, 257, 3), (512, 300, 4), (512, 4096, 7),
                                   (1024, 257, 10), (1024, 1000, 100), (3000, 50, 5)])
def test_minus_cluster_matches_oracle(gpu_ctx, oracle, p, n, K):
    import sparsifiedkmeans_amd as S

    X = random_csc(p, n, max(1, p // 20), seed=p + n + K, ragged=True, empty_cols=(0,) if n > 1 else ())
    Cm = np.random.default_rng(K).standard_normal((p, K))
    got = S.SparseMatrixMinusCluster(X, Cm, ctx=gpu_ctx)
    ref = oracle.dist_csc(p, n, *parts(X), Cm)
    assert got.shape == (K, n)
    assert np.array_equal(got, ref)  # bit-exact


def test_minus_cluster_beta(gpu_ctx, oracle):
    import sparsifiedkmeans_amd as S

    X = random_csc(128, 500, 9, seed=5, ragged=True)
    c = np.random.default_rng(1).standard_normal(128)
    got = S.SparseMatrixMinusCluster(X, c, beta=0.37, ctx=gpu_ctx)
    ref = oracle.dist_csc_beta(500, *parts(X), c, 0.37)
    assert np.array_equal(got[0], ref, equal_nan=True)  # sqrt of a slightly negative sum is NaN on both


def test_minus_cluster_errors(gpu_ctx):
    import sparsifiedkmeans_amd as S
    from sparsifiedkmeans_amd._lib import SpkmError

    X = random_csc(16, 4, 3, seed=0)
    with pytest.raises(SpkmError, match="did not have p rows"):        # SparseMatrixMinusCluster.c:104-107
        S.SparseMatrixMinusCluster(X, np.zeros((15, 2)), ctx=gpu_ctx)
    with pytest.raises(SpkmError, match="beta"):                       # :119-120
        S.SparseMatrixMinusCluster(X, np.zeros((16, 2)), beta=1.0, ctx=gpu_ctx)
    with pytest.raises(TypeError, match="sparse"):                     # :72-75
        S.SparseMatrixMinusCluster(np.zeros((16, 4)), np.zeros(16), ctx=gpu_ctx)


def test_inner_product_and_norms(gpu_ctx, oracle):
    import sparsifiedkmeans_amd as S

    X = random_csc(700, 3000, 35, seed=11, ragged=True, empty_cols=(3, 2999))
    c = np.random.default_rng(2).standard_normal(700)
    ip, nx2 = S.SparseMatrixInnerProduct(X, c, ctx=gpu_ctx)
    rip, rnx2 = oracle.innerprod_csc(3000, *parts(X), c)
    assert np.array_equal(ip, rip) and np.array_equal(nx2, rnx2)
    assert np.array_equal(S.SparseMatrixColumnNormSq(X, ctx=gpu_ctx), oracle.colnormsq_csc(3000, X.indptr, X.data))


def _golden(prefix):
    import os

    gold = os.path.join(os.path.dirname(__file__), "golden")
    files = sorted(f for f in os.listdir(gold) if f.startswith(prefix))
    return [(f, np.load(os.path.join(gold, f))) for f in files]


def _fixture_csc(z):
    return sp.csc_matrix((z["x"], z["ir"].astype(np.int64), z["jc"].astype(np.int64)), shape=(int(z["p"]), int(z["n"])))


def test_sparse_operators_equal_the_reference_s_own_outputs(gpu_ctx, oracle):
    """The three stand-alone HIP operators against vectors produced by the REFERENCE's own loops
    (tests/golden/make_ref_fixtures.py: SparseMatrixMinusCluster.c:121-129,131-183, SparseMatrixInnerProduct.c:86-100,
    SparseMatrixColumnNormSq.c:70-77 compiled from /root/reference in the build container) -- bit for bit, every branch
    of the reference's `switch (K)`.  Where the prebuilt oracle/_ref binary travelled with the snapshot, fresh random
    matrices go through it as well."""
    import sparsifiedkmeans_amd as S

    dist = _golden("ref_dist_")
    assert len(dist) == 8
    for f, z in dist:
        got = S.SparseMatrixMinusCluster(_fixture_csc(z), z["C"], ctx=gpu_ctx)
        assert np.array_equal(got, z["dist"]), f
    (f, z), = _golden("ref_beta_")
    for b in z["betas"]:
        key = "dist_beta_" + str(float(b)).replace(".", "p").replace("-", "m")
        got = S.SparseMatrixMinusCluster(_fixture_csc(z), z["c"], beta=float(b), ctx=gpu_ctx)
        assert np.array_equal(got[0], z[key], equal_nan=True), key
    (f, z), = _golden("ref_ip_")
    ip, nx2 = S.SparseMatrixInnerProduct(_fixture_csc(z), z["c"], ctx=gpu_ctx)
    assert np.array_equal(ip, z["ip"]) and np.array_equal(nx2, z["nx2"])
    assert np.array_equal(S.SparseMatrixColumnNormSq(_fixture_csc(z), ctx=gpu_ctx), z["nsq"])
    if oracle.ref_available("sparse"):
        for p, n, K, s in [(1024, 3000, 100, 51), (784, 2000, 10, 40), (300, 5000, 3, 15), (64, 9000, 1, 6)]:
            X = random_csc(p, n, s, seed=p + K, ragged=True, empty_cols=(0, n - 1))
            Cm = np.random.default_rng(K).standard_normal((p, K))
            jc, ir, x = parts(X)
            assert np.array_equal(S.SparseMatrixMinusCluster(X, Cm, ctx=gpu_ctx), oracle.ref_dist_csc(p, n, jc, ir, x, Cm))
            if K == 1:
                c = Cm[:, 0]
                assert np.array_equal(S.SparseMatrixMinusCluster(X, c, beta=0.3, ctx=gpu_ctx)[0],
                                      oracle.ref_dist_csc_beta(n, jc, ir, x, c, 0.3), equal_nan=True)
                ip, nx2 = S.SparseMatrixInnerProduct(X, c, ctx=gpu_ctx)
                rip, rnx2 = oracle.ref_innerprod_csc(n, jc, ir, x, c)
                assert np.array_equal(ip, rip) and np.array_equal(nx2, rnx2)
                assert np.array_equal(S.SparseMatrixColumnNormSq(X, ctx=gpu_ctx), oracle.ref_colnormsq_csc(n, jc, x))


@pytest.mark.parametrize("m", [2, 4, 8, 16, 32, 64, 256, 1024, 4096, 16384, 32768])
@pytest.mark.parametrize("n", [1, 3, 17])
def test_hadamard_bit_exact(gpu_ctx, oracle, m, n):
    import sparsifiedkmeans_amd as S

    x = np.random.default_rng(m + n).standard_normal((m, n))
    ref = oracle.fwht(x)
    assert np.array_equal(S.hadamard(x, ctx=gpu_ctx), ref)
    assert np.array_equal(S.hadamard_pthreads(x, ctx=gpu_ctx), ref)


def test_hadamard_equals_the_reference_s_own_outputs(gpu_ctx, oracle):
    """The HIP FWHT against vectors produced by the REFERENCE's code (tests/golden/make_ref_fixtures.py: hadamard.c:57-92
    and hadamard_pthreads.c:57-119 compiled from /root/reference in the build container) -- bit for bit.  Where the
    prebuilt oracle/_ref binary travelled with the snapshot, a fresh random matrix goes through it as well."""
    import os

    import sparsifiedkmeans_amd as S

    gold = os.path.join(os.path.dirname(__file__), "golden")
    files = sorted(f for f in os.listdir(gold) if f.startswith("ref_fwht_"))
    assert len(files) >= 5
    for f in files:
        z = np.load(os.path.join(gold, f))
        assert np.array_equal(S.hadamard(z["x"], ctx=gpu_ctx), z["out"]), f
        assert np.array_equal(S.hadamard_pthreads(z["x"], ctx=gpu_ctx), z["out"]), f
    if oracle.ref_available("portable"):
        x = np.random.default_rng(77).standard_normal((1024, 2500))
        assert np.array_equal(S.hadamard(x, ctx=gpu_ctx), oracle.ref_fwht(x, "portable"))
        if oracle.ref_available("pthreads"):
            assert np.array_equal(S.hadamard_pthreads(x, ctx=gpu_ctx), oracle.ref_fwht(x, "pthreads", 8))


def test_hadamard_many_columns_and_vector(gpu_ctx, oracle):
    import sparsifiedkmeans_amd as S

    x = np.random.default_rng(0).standard_normal((1024, 4099))
    assert np.array_equal(S.hadamard(x, ctx=gpu_ctx), oracle.fwht(x))
    v = np.random.default_rng(1).standard_normal(64)
    assert np.array_equal(S.hadamard(v, ctx=gpu_ctx), oracle.fwht(v)[:, 0])


def test_hadamard_errors(gpu_ctx):
    import sparsifiedkmeans_amd as S
    from sparsifiedkmeans_amd._lib import SpkmError

    with pytest.raises(SpkmError, match="power of 2"):          # hadamard.c:108-110
        S.hadamard(np.zeros((12, 2)), ctx=gpu_ctx)
    with pytest.raises(SpkmError, match="greater than 1"):      # hadamard.c:100-102
        S.hadamard(np.zeros((1, 2)), ctx=gpu_ctx)
    with pytest.raises(TypeError):                               # hadamard.c:137-140
        S.hadamard(sp.csc_matrix(np.eye(4)), ctx=gpu_ctx)
    with pytest.raises(TypeError):                               # hadamard.c:134-136
        S.hadamard(np.zeros((4, 1), complex), ctx=gpu_ctx)
