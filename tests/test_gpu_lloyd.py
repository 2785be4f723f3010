"""-m gpu: accumulate / finalize / full Lloyd iterations vs the oracle."""
import numpy as np
import pytest
import torch

from util import parts, random_csc

pytestmark = pytest.mark.gpu


def _engine(ctx, X, K, gamma):
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    return LloydEngine(Shard.from_scipy(ctx, X), K, gamma)


@pytest.mark.parametrize("p,n,K,s", [(512, 5000, 5, 26), (1024, 20000, 100, 51), (2048, 2000, 9, 30),
                                     (20000, 500, 3, 40)])
def test_accumulate_counts_exact_sums_close(gpu_ctx, oracle, p, n, K, s):
    X = random_csc(p, n, s, seed=n, ragged=True, empty_cols=(2,))
    Cm = np.random.default_rng(5).standard_normal((p, K))
    eng = _engine(gpu_ctx, X, K, 0.05)
    centers = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda:0")
    eng.assign_step(centers)
    eng.accumulate_step()
    red = eng.reduce.cpu().numpy()
    a = eng.assign.cpu().numpy()
    S, Cnt, nk = oracle.accumulate(p, n, K, *parts(X), a)
    pk = p * K
    got_S = red[:pk].reshape(K, p).T
    got_C = red[pk:2 * pk].reshape(K, p).T
    assert np.array_equal(got_C, Cnt)                           # integer counts: exact
    assert np.array_equal(red[2 * pk:2 * pk + K], nk.astype(float))
    scale = np.abs(S).max()
    assert np.abs(got_S - S).max() <= 1e-12 * scale             # order of summation differs only
    d = eng.mind.cpu().numpy()
    assert abs(red[-1] - np.sum(d * d)) <= 1e-12 * np.sum(d * d)


def test_lloyd_iterations_teacher_forced_and_end_to_end(gpu_ctx, oracle):
    """Per iteration: same (X, centres) in -> same assignments out (bit-exact) and centres within
    1e-6 relative (BASELINE.json north_star); then a free-running comparison."""
    from sparsifiedkmeans_amd import synth

    data = synth.sparsified_gmm_host(p=512, n=5000, K=5, gamma=0.05, seed=234, fwht=oracle.fwht)
    Y, p2, gamma, K = data["Y"], data["p2"], data["gamma"], 5
    rng = np.random.default_rng(1)
    C0 = oracle.mix(data["X"][:, rng.choice(5000, K, replace=False)], data["d"], p2)
    eng = _engine(gpu_ctx, Y, K, gamma)
    centers = torch.tensor(np.ascontiguousarray(C0.T), device="cuda:0")
    Cref = C0.copy()
    for it in range(6):
        # teacher forcing: both sides start the iteration from the oracle's centres
        centers.copy_(torch.tensor(np.ascontiguousarray(Cref.T)))
        out = eng.iterate(centers).cpu().numpy()
        ref = oracle.lloyd(p2, Y.shape[1], *parts(Y), Cref, gamma, maxiter=1, tol=0.0)
        assert np.array_equal(eng.assign.cpu().numpy(), ref["assign"]), f"iteration {it}"
        assert np.array_equal(eng.mind.cpu().numpy(), ref["mind"])
        got = centers.cpu().numpy().T
        assert np.abs(got - ref["centers"]).max() <= 1e-6 * np.abs(ref["centers"]).max()
        assert abs(np.sqrt(out[0]) - ref["dff"][0]) <= 1e-9 * max(ref["dff"][0], 1e-300) + 1e-12
        assert abs(np.sqrt(out[1]) - ref["obj"][0]) <= 1e-12 * ref["obj"][0]
        Cref = ref["centers"]
    # free-running: 20 iterations from C0 on both sides
    centers.copy_(torch.tensor(np.ascontiguousarray(C0.T)))
    for it in range(20):
        eng.iterate(centers)
    ref = oracle.lloyd(p2, Y.shape[1], *parts(Y), C0, gamma, maxiter=20, tol=0.0)
    eng.assign_step(centers)
    a_ref, _ = oracle.assign(p2, Y.shape[1], *parts(Y), ref["centers"], gamma)
    assert np.count_nonzero(eng.assign.cpu().numpy() != a_ref) == 0
    got = centers.cpu().numpy().T
    assert np.abs(got - ref["centers"]).max() <= 1e-6 * np.abs(ref["centers"]).max()


def test_empty_cluster_keeps_column_and_reports(gpu_ctx, oracle):
    p, n, K = 256, 2000, 6
    X = random_csc(p, n, 12, seed=4)
    Cm = np.random.default_rng(2).standard_normal((p, K))
    Cm[:, 4] = 1e6  # nobody is close to this centre
    eng = _engine(gpu_ctx, X, K, 0.05)
    centers = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda:0")
    eng.iterate(centers)
    assert eng.nk.cpu().numpy()[4] == 0
    assert np.array_equal(centers.cpu().numpy()[4], Cm[:, 4])  # untouched: host applies EmptyAction


def test_mix_fused_matches_oracle(gpu_ctx, oracle):
    from sparsifiedkmeans_amd.engine import mix_device

    p, p2, n = 784, 1024, 257
    rng = np.random.default_rng(0)
    X = rng.standard_normal((p, n))
    d = np.sign(rng.standard_normal(p2))
    y = mix_device(gpu_ctx, torch.tensor(np.ascontiguousarray(X.T), device="cuda:0"), p2,
                   torch.tensor(d, device="cuda:0"), 1.0 + 2 * np.finfo(float).eps, float(np.sqrt(p2)))
    assert np.array_equal(y.cpu().numpy().T, oracle.mix(X, d, p2))


@pytest.mark.parametrize("n", [1_048_581, 1_048_578, 2_000_003])
@pytest.mark.parametrize("fused", [False, True])
def test_counting_sort_with_n_not_a_multiple_of_four(gpu_ctx, oracle, n, fused):
    """k_scatter_by_cluster reads the assignment four points at a time and rounds every workgroup's span up to a multiple
    of four: from n = 1 048 577 on, trailing workgroups start PAST the end, and (round 2) their scalar tail placed the
    last n % 4 points a second time -- one cluster got nk + 1 entries in the permutation.  Counts
    exact and every point accumulated exactly once, on the exact path and on the fused one."""
    import scipy.sparse as sp
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, s, K, gamma = 32, 3, 5, 3 / 32
    rng = np.random.default_rng(n)
    rows = np.sort(np.argsort(rng.random((n, p)), axis=1)[:, :s], axis=1)          # s distinct ascending rows per point
    vals = rng.standard_normal((n, s)) * 2.0
    X = sp.csc_matrix((vals.ravel(), rows.ravel().astype(np.int64), np.arange(0, (n + 1) * s, s)), shape=(p, n))
    Cm = rng.standard_normal((p, K))
    eng = LloydEngine(Shard.from_scipy(gpu_ctx, X), K, gamma)
    assert eng.assign.data_ptr() % 16 == 0                                         # the 16-B vector path
    centers = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda:0")
    if fused:
        eng.assign_accumulate_step(centers)
    else:
        eng.assign_step(centers)
        eng.accumulate_step()
    a = eng.assign.cpu().numpy()
    a_ref, d_ref = oracle.assign(p, n, *parts(X), Cm, gamma)
    assert np.array_equal(a, a_ref) and np.array_equal(eng.mind.cpu().numpy(), d_ref)
    S, Cnt, nk = oracle.accumulate(p, n, K, *parts(X), a)
    red = eng.reduce.cpu().numpy()
    pk = p * K
    assert np.array_equal(red[pk:2 * pk].reshape(K, p).T, Cnt)
    assert np.array_equal(red[2 * pk:2 * pk + K], nk.astype(float))
    assert np.abs(red[:pk].reshape(K, p).T - S).max() <= 1e-11 * np.abs(S).max()
    assert abs(red[-1] - np.sum(d_ref * d_ref)) <= 1e-11 * np.sum(d_ref * d_ref)
