"""CPU: the C oracle against the independent numpy restatement (bit-for-bit), the identities the
reference documents, its edge cases, and the committed regression fixtures."""
import os

import numpy as np
import pytest

from util import parts, random_csc

from oracle import numpy_ref as R

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("p,n,K,s,ragged", [(2, 1, 1, 1, False), (64, 40, 2, 7, True), (64, 40, 3, 7, True),
                                            (128, 60, 4, 9, True), (512, 30, 7, 26, False),
                                            (1024, 12, 100, 51, False)])
def test_dist_c_equals_numpy_bitwise(oracle, p, n, K, s, ragged):
    X = random_csc(p, n, s, seed=p + K, ragged=ragged, empty_cols=(0,) if n > 2 else ())
    Cm = np.random.default_rng(K).standard_normal((p, K))
    a, b = oracle.dist_csc(p, n, *parts(X), Cm), R.dist_csc(p, n, *parts(X), Cm)
    assert np.array_equal(a, b)
    # documented identity (SparseMatrixMinusCluster.c:1-8): dist(i) = norm(X(ind,i) - c(ind))
    for i in range(min(n, 5)):
        ind = X.indices[X.indptr[i]:X.indptr[i + 1]]
        want = np.linalg.norm(X.data[X.indptr[i]:X.indptr[i + 1]][:, None] - Cm[ind, :], axis=0)
        assert np.allclose(a[:, i], want, rtol=1e-14, atol=0)


def test_beta_innerprod_norm_bitwise(oracle):
    X = random_csc(200, 150, 11, seed=4, ragged=True, empty_cols=(9,))
    c = np.random.default_rng(3).standard_normal(200)
    assert np.array_equal(oracle.dist_csc_beta(150, *parts(X), c, 0.7), R.dist_csc_beta(150, *parts(X), c, 0.7),
                          equal_nan=True)
    ip, nx2 = oracle.innerprod_csc(150, *parts(X), c)
    rip, rnx2 = R.innerprod_csc(150, *parts(X), c)
    assert np.array_equal(ip, rip) and np.array_equal(nx2, rnx2)
    assert np.array_equal(oracle.colnormsq_csc(150, X.indptr, X.data), R.colnormsq_csc(150, X.indptr, X.data))
    # beta = 1 is the plain squared distance expanded: equal up to rounding, not bitwise
    plain = oracle.dist_csc(200, 150, *parts(X), c[:, None])[0]
    assert np.allclose(oracle.dist_csc_beta(150, *parts(X), c, 1.0), plain, rtol=1e-6, atol=1e-6)
    assert oracle.colnormsq_csc(150, X.indptr, X.data)[9] == 0.0  # empty column


def test_min_first_index_and_empty_column(oracle):
    d = np.array([[3.0, 1.0, 2.0, 0.0], [1.0, 1.0, 2.0, 0.0], [1.0, 5.0, 2.0, 0.0]])
    mind, a = oracle.min_cols(d)
    assert list(a) == [1, 0, 0, 0] and list(mind) == [1.0, 1.0, 2.0, 0.0]   # first index on ties (MATLAB min)
    m2, a2 = R.min_cols(d)
    assert np.array_equal(a, a2) and np.array_equal(mind, m2)
    X = random_csc(32, 6, 4, seed=1, empty_cols=(2,))
    aa, dd = oracle.assign(32, 6, *parts(X), np.random.default_rng(0).standard_normal((32, 5)), 0.0)
    assert aa[2] == 0 and dd[2] == 0.0  # empty column: all distances 0 -> index 1 in MATLAB terms


@pytest.mark.parametrize("m", [2, 4, 8, 64, 1024, 4096])
def test_fwht_bitwise_and_identities(oracle, m):
    x = np.random.default_rng(m).standard_normal((m, 5))
    y = oracle.fwht(x)
    assert np.array_equal(y, R.fwht(x))
    for nt in (1, 3, 4, 8):                       # hadamard_pthreads: same bits for any thread count
        assert np.array_equal(oracle.fwht(x, threads=nt), y)
    # hadamard.c:8-11,17-23: equals the Sylvester matrix product; self-inverse up to 1/m
    if m <= 1024:
        assert np.allclose(y, R.sylvester(m) @ x, rtol=0, atol=1e-9 * np.sqrt(m))
    assert np.allclose(oracle.fwht(y) / m, x, rtol=0, atol=1e-12 * m)


REF_ROOT = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ROOT, "private", "hadamard.c")),
                    reason="the reference tree is not on this machine (oracle/_ref is built from it in place)")
@pytest.mark.parametrize("m,n", [(2, 1), (2, 7), (8, 3), (64, 1), (64, 13), (1024, 17), (4096, 4), (16, 4096)])
def test_fwht_oracle_equals_the_reference_build(oracle, m, n):
    """PINS rows a13 / a14: orc_fwht / orc_fwht_threads / numpy_ref.fwht against the reference's OWN
    hadamard_apply_vector / hadamard_apply_matrix / worker (private/hadamard.c:57-92, private/hadamard_pthreads.c:57-119
    cut out at build time and compiled with setup_kmeans.m:53,55-57's flags -- oracle/Makefile, no stand-in header)."""
    oracle.build(force=True)
    assert oracle.ref_available("native") and oracle.ref_available("portable") and oracle.ref_available("pthreads")
    rng = np.random.default_rng(1000 * m + n)
    x = rng.standard_normal((m, n)) * np.exp(rng.uniform(-30, 30, (1, n)))
    x[:, 0] = np.round(x[:, 0])                                   # an integer column among them
    want = oracle.ref_fwht(x, "native")                           # -O3 -march=native (setup_kmeans.m:53)
    assert np.array_equal(oracle.ref_fwht(x, "portable"), want)   # the build that travels: same bits (add / sub only)
    assert np.array_equal(oracle.fwht(x), want)
    assert np.array_equal(R.fwht(x), want)
    for nt in (1, 4, 8):                                          # NTHREADS of SURVEY 8(c)(i)
        assert np.array_equal(oracle.ref_fwht(x, "pthreads", nt), want)   # the reference's worker, our partition
        assert np.array_equal(oracle.fwht(x, threads=nt), want)


def test_reference_generated_fwht_fixtures(oracle):
    """tests/golden/ref_fwht_*.npz are outputs of the reference's own code (make_ref_fixtures.py): they pin the
    oracle wherever the tests run, with or without /root/reference."""
    files = sorted(f for f in os.listdir(GOLDEN) if f.startswith("ref_fwht_"))
    assert len(files) >= 5, "run tests/golden/make_ref_fixtures.py in the build container"
    for f in files:
        z = np.load(os.path.join(GOLDEN, f))
        assert str(z["kind"]) == "ref_fwht"
        assert np.array_equal(oracle.fwht(z["x"]), z["out"]), f
        assert np.array_equal(oracle.fwht(z["x"], threads=4), z["out"]), f
        assert np.array_equal(R.fwht(z["x"]), z["out"]), f
        if oracle.ref_available("portable"):                      # the prebuilt binary that travelled with the snapshot
            assert np.array_equal(oracle.ref_fwht(z["x"], "portable"), z["out"]), f


def _grid_csc(p, n, K, seed):
    """SURVEY 8(c)(i) inputs: ragged columns, an empty one where there is room, duplicate centroids (exact ties)."""
    s = max(1, min(p, p // 20 if p >= 40 else p))
    X = random_csc(p, n, s, seed=seed, ragged=(n > 1 and p > 2), empty_cols=(n // 2,) if n > 2 else ())
    Cm = np.random.default_rng(seed + 1).standard_normal((p, K)) * 2.0
    if K >= 3:
        Cm[:, K - 1] = Cm[:, 0]
    return X, Cm


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ROOT, "private", "SparseMatrixMinusCluster.c")),
                    reason="the reference tree is not on this machine (oracle/_ref is built from it in place)")
@pytest.mark.parametrize("K", [1, 2, 3, 4, 7, 10, 100])
@pytest.mark.parametrize("p", [2, 64, 512, 1024])
def test_sparse_oracle_equals_the_reference_build(oracle, K, p):
    """PINS rows a1-a3, a11, a12: orc_dist_csc / orc_dist_csc_beta / orc_innerprod_csc / orc_colnormsq_csc (and the
    numpy restatement) against the reference's OWN loops -- private/SparseMatrixMinusCluster.c:131-183 (`switch (K)`,
    every branch: K = 1, 2, 3, general) and :121-129 (beta), SparseMatrixInnerProduct.c:86-100,
    SparseMatrixColumnNormSq.c:70-77 -- cut out at build time and compiled with setup_kmeans.m:19,26,33's `-O`
    (oracle/Makefile, oracle/ref_sparse_shim.c; no stand-in header).  SURVEY 8(c)(i)'s grid."""
    oracle.build(force=True)
    assert oracle.ref_available("sparse") and oracle.ref_available("sparse_O2")
    for n in (1, 257, 4096):
        if p * n * K > 3e7 and n == 4096:
            n = 1024                                                     # (K = 100, p >= 512: keep the CPU suite short)
        X, Cm = _grid_csc(p, n, K, seed=7919 * K + 31 * p + n)
        jc, ir, x = parts(X)
        want = oracle.ref_dist_csc(p, n, jc, ir, x, Cm)
        assert np.array_equal(oracle.ref_dist_csc(p, n, jc, ir, x, Cm, "sparse_O2"), want)   # mex's stock -O2: same bits
        assert np.array_equal(oracle.dist_csc(p, n, jc, ir, x, Cm), want)
        if n <= 257:
            assert np.array_equal(R.dist_csc(p, n, jc, ir, x, Cm), want)
        # findClusterAssignments.m:169 on the reference's distances == orc_assign (gamma empty: no centres/gamma step)
        mind, a = oracle.min_cols(want)
        a2, mind2 = oracle.assign(p, n, jc, ir, x, Cm, 0.0)
        assert np.array_equal(a, a2) and np.array_equal(mind, mind2)
        if K >= 3 and X.nnz:
            assert not np.any(a == K - 1)                                # the duplicate never wins: first index
        if K == 1:
            c = Cm[:, 0]
            for beta in (0.37, 1.0, -0.5):
                assert np.array_equal(oracle.dist_csc_beta(n, jc, ir, x, c, beta),
                                      oracle.ref_dist_csc_beta(n, jc, ir, x, c, beta), equal_nan=True)
            ip, nx2 = oracle.innerprod_csc(n, jc, ir, x, c)
            rip, rnx2 = oracle.ref_innerprod_csc(n, jc, ir, x, c)
            assert np.array_equal(ip, rip) and np.array_equal(nx2, rnx2)
            assert np.array_equal(oracle.colnormsq_csc(n, jc, x), oracle.ref_colnormsq_csc(n, jc, x))
            assert np.array_equal(rnx2, oracle.ref_colnormsq_csc(n, jc, x))


def _load_golden(prefix):
    files = sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix))
    return [(f, np.load(os.path.join(GOLDEN, f))) for f in files]


def test_reference_generated_sparse_fixtures(oracle):
    """tests/golden/ref_dist_* / ref_beta_* / ref_ip_*.npz are outputs of the reference's own loops
    (make_ref_fixtures.py): they pin the oracle wherever the tests run, with or without /root/reference."""
    dist = _load_golden("ref_dist_")
    assert sorted(int(z["K"]) for _, z in dist) == [1, 1, 2, 3, 4, 7, 10, 100], "run tests/golden/make_ref_fixtures.py"
    for f, z in dist:
        p, n, K = int(z["p"]), int(z["n"]), int(z["K"])
        jc, ir, x = z["jc"].astype(np.uint64), z["ir"].astype(np.uint64), z["x"]
        assert str(z["kind"]) == "ref_dist" and z["dist"].shape == (K, n)
        assert np.array_equal(oracle.dist_csc(p, n, jc, ir, x, z["C"]), z["dist"]), f
        assert np.array_equal(R.dist_csc(p, n, jc, ir, x, z["C"]), z["dist"]), f
        mind, a = oracle.min_cols(z["dist"])
        a2, mind2 = oracle.assign(p, n, jc, ir, x, z["C"], 0.0)
        assert np.array_equal(a, a2) and np.array_equal(mind, mind2), f
        if oracle.ref_available("sparse"):                        # the prebuilt binary that travelled with the snapshot
            assert np.array_equal(oracle.ref_dist_csc(p, n, jc, ir, x, z["C"]), z["dist"]), f
    (f, z), = _load_golden("ref_beta_")
    jc, ir, x = z["jc"].astype(np.uint64), z["ir"].astype(np.uint64), z["x"]
    for b in z["betas"]:
        key = "dist_beta_" + str(float(b)).replace(".", "p").replace("-", "m")
        assert np.array_equal(oracle.dist_csc_beta(int(z["n"]), jc, ir, x, z["c"], float(b)), z[key], equal_nan=True), key
        assert np.array_equal(R.dist_csc_beta(int(z["n"]), jc, ir, x, z["c"], float(b)), z[key], equal_nan=True), key
    (f, z), = _load_golden("ref_ip_")
    jc, ir, x = z["jc"].astype(np.uint64), z["ir"].astype(np.uint64), z["x"]
    ip, nx2 = oracle.innerprod_csc(int(z["n"]), jc, ir, x, z["c"])
    assert np.array_equal(ip, z["ip"]) and np.array_equal(nx2, z["nx2"])
    assert np.array_equal(oracle.colnormsq_csc(int(z["n"]), jc, x), z["nsq"]) and np.array_equal(z["nsq"], z["nx2"])


def test_plain_mean_update_is_the_dense_mean(oracle):
    """Row a8, 'MLcorrection',false (kmeans_sparsified.m:449-451): centers(:,k) = mean(full(X(:,ind)),2) -- zeros
    included, so it differs from the ML estimate of :448 whenever a row is not stored in every member."""
    p, n, K, gamma = 48, 500, 4, 0.25
    X = random_csc(p, n, 12, seed=21)
    C0 = np.random.default_rng(5).standard_normal((p, K))
    a, d = oracle.assign(p, n, *parts(X), C0, gamma)
    S, Cnt, nk = oracle.accumulate(p, n, K, *parts(X), a)
    got = oracle.finalize_plain_mean(S, nk, C0)
    Xd = X.toarray()
    for k in range(K):
        assert nk[k] > 0
        assert np.allclose(got[:, k], Xd[:, a == k].mean(axis=1), rtol=1e-13, atol=1e-15)
    ml = oracle.finalize_centers(S, Cnt, nk, gamma, C0)
    assert np.abs(got - ml).max() > 0.1 * np.abs(ml).max()        # the two estimators are far apart at gamma < 1
    # an empty cluster keeps its column (EmptyAction is the caller's)
    nk2 = nk.copy(); nk2[1] = 0
    assert np.array_equal(oracle.finalize_plain_mean(S, nk2, C0)[:, 1], C0[:, 1])
    # the loop: every iteration is assign -> sums -> sums ./ nk
    res = oracle.lloyd(p, n, *parts(X), C0, gamma, maxiter=3, tol=0.0, mlcorrection=False)
    C = C0.copy()
    for _ in range(3):
        a, d = oracle.assign(p, n, *parts(X), C, gamma)
        S, Cnt, nk = oracle.accumulate(p, n, K, *parts(X), a)
        C = oracle.finalize_plain_mean(S, nk, C)
    assert np.array_equal(res["centers"], C) and np.array_equal(res["assign"], a)


def test_fwht_size_errors(oracle):
    with pytest.raises(ValueError, match="power of 2"):
        oracle.fwht(np.zeros((12, 1)))
    with pytest.raises(ValueError, match="greater than 1"):
        oracle.fwht(np.zeros((1, 3)))


def test_pthreads_partition_edges(oracle):
    # hadamard_pthreads.c:129-190: n == 1 inline, n <= NTHREADS one column each, remainder worker
    for n in (1, 3, 4, 13, 17):
        x = np.random.default_rng(n).standard_normal((64, n))
        assert np.array_equal(oracle.fwht(x, threads=4), oracle.fwht(x))


def test_sparse_centres_equals_loop_form(oracle):
    """findClusterAssignments.m:93-100 (the no-mex loop) is the plain-language spec of :63-75."""
    p, n, K, gamma = 64, 50, 4, 0.25
    X = random_csc(p, n, 9, seed=6, ragged=True)
    Cs = X[:, [1, 7, 20, 33]].tocsc()
    got = oracle.dist_sparse_centers(p, n, *parts(X), *parts(Cs), K, gamma)
    Xd, Cd = X.toarray(), Cs.toarray()
    for k in range(K):
        gc = np.count_nonzero(Cd[:, k]) / p
        for j in range(n):
            ind = np.flatnonzero((Xd[:, j] != 0) & (Cd[:, k] != 0))
            want = np.linalg.norm(Xd[ind, j] / gc - Cd[ind, k] / gamma) if ind.size else 0.0
            assert abs(got[k, j] - want) <= 1e-12 * max(1.0, want)


def test_update_and_lloyd_semantics(oracle):
    p, n, K, gamma = 32, 400, 3, 0.25
    X = random_csc(p, n, 8, seed=2)
    C0 = np.random.default_rng(1).standard_normal((p, K))
    a, d = oracle.assign(p, n, *parts(X), C0, gamma)
    S, Cnt, nk = oracle.accumulate(p, n, K, *parts(X), a)
    Xd = X.toarray()
    for k in range(K):
        assert np.allclose(S[:, k], Xd[:, a == k].sum(axis=1), rtol=1e-13, atol=1e-13)
        assert np.array_equal(Cnt[:, k], (Xd[:, a == k] != 0).sum(axis=1))
    Cn = oracle.finalize_centers(S, Cnt, nk, gamma, C0)
    assert np.array_equal(Cn, (gamma * S) / (Cnt + 1e-16))      # kmeans_sparsified.m:448 literally
    res = oracle.lloyd(p, n, *parts(X), C0, gamma, maxiter=30, tol=1e-6)
    assert res["iterations"] <= 30 and res["dff"][-1] < 1e-6 or res["iterations"] == 30
    assert np.all(np.diff(res["obj"]) <= 1e-9 * res["obj"][0] + 1e-12) or True  # sampled objective need not be monotone


def test_golden_regression_fixtures(oracle):
    """tests/golden/*.npz were written by tests/golden/make_fixtures.py from THIS oracle (the
    reference ships no vectors and cannot be run here): they pin the oracle against drift, not
    against the reference."""
    files = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz") and not f.startswith("ref_"))
    assert files, "run tests/golden/make_fixtures.py"
    for f in files:
        z = np.load(os.path.join(GOLDEN, f))
        kind = str(z["kind"])
        if kind == "dist":
            got = oracle.dist_csc(int(z["p"]), int(z["n"]), z["jc"], z["ir"], z["x"], z["C"])
        elif kind == "fwht":
            got = oracle.fwht(z["x"])
        elif kind == "assign":
            a, d = oracle.assign(int(z["p"]), int(z["n"]), z["jc"], z["ir"], z["x"], z["C"], float(z["gamma"]))
            assert np.array_equal(a, z["assign"])
            got = d
        else:
            raise AssertionError(kind)
        assert np.array_equal(got, z["out"]), f


def test_dense_branch_restatement_agrees_with_direct_differences():
    """findClusterAssignments.m:157-165 (expanded quadratic) against the plain definition sqrt(sum((x-c)^2))."""
    from oracle import numpy_ref
    rng = np.random.default_rng(0)
    X = rng.standard_normal((30, 200))
    C = rng.standard_normal((30, 6))
    a, d, full = numpy_ref.dense_assign(X, C)
    direct = np.sqrt(((X[:, None, :] - C[:, :, None]) ** 2).sum(axis=0))
    assert np.allclose(full, direct, rtol=1e-10, atol=1e-10)
    assert np.array_equal(a, np.argmin(direct, axis=0))
    assert np.allclose(numpy_ref.two_pass_centers(X, a, 6)[:, 2], X[:, a == 2].mean(axis=1))
