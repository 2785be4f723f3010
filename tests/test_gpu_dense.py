"""Two-pass outputs: dense assignment on the f64 matrix cores and dense per-cluster
means, against the numpy restatement of findClusterAssignments.m:157-171 / kmeans_sparsified.m:543-568.
The reference's own arithmetic here is a BLAS product (order undefined), so distances are compared to a
tolerance and assignments wherever the best / second-best gap exceeds it."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu


def _ref():
    from oracle import numpy_ref
    return numpy_ref


def _check_assign(X, C, a, d):
    a0, d0, full = _ref().dense_assign(X, C)
    scale = np.sum(X * X, axis=0) + np.max(np.sum(C * C, axis=0))
    # d^2 carries an absolute error ~ 1e-15 * (|x|^2 + |c|^2) on both sides (cancellation)
    tol2 = 64 * np.finfo(np.float64).eps * scale
    assert np.all(np.abs(d * d - d0 * d0) <= tol2 + 1e-300), float(np.max(np.abs(d * d - d0 * d0) / scale))
    srt = np.sort(full * full, axis=0)
    clear = (srt[1] - srt[0] > 4 * tol2) if full.shape[0] > 1 else np.ones(X.shape[1], bool)
    assert np.array_equal(a[clear], a0[clear])
    # everywhere: the chosen centre is (numerically) a minimiser
    chosen = full[a, np.arange(X.shape[1])]
    assert np.all(chosen * chosen - d0 * d0 <= 4 * tol2)


@pytest.mark.parametrize("p,n,K", [(64, 64, 16), (100, 257, 5), (256, 1000, 100), (33, 1, 3), (512, 300, 130),
                                   (1024, 129, 17), (8, 70, 1), (3, 5, 2)])
def test_dense_assign_matches_expanded_quadratic(p, n, K):
    import torch
    from sparsifiedkmeans_amd.engine import dense_assign_device, torch_context
    rng = np.random.default_rng(p * 7 + n + K)
    C = rng.standard_normal((p, K))
    lab = rng.integers(0, K, n)
    X = C[:, lab] + 0.3 * rng.standard_normal((p, n))
    ctx = torch_context()
    a, d = dense_assign_device(ctx, torch.tensor(np.ascontiguousarray(X.T), device="cuda"),
                               torch.tensor(np.ascontiguousarray(C.T), device="cuda"))
    _check_assign(X, C, a.cpu().numpy(), d.cpu().numpy())


def test_dense_assign_ties_take_first_index_and_exact_hits_are_zero():
    import torch
    from sparsifiedkmeans_amd.engine import dense_assign_device, torch_context
    p, K = 32, 20
    rng = np.random.default_rng(5)
    C = np.round(rng.standard_normal((p, K)) * 4) / 4      # exactly representable: products and sums are exact
    C[:, 7] = C[:, 3]
    C[:, 19] = C[:, 3]
    X = C[:, [3, 7, 19, 0, 5]].copy()
    a, d = dense_assign_device(torch_context(), torch.tensor(np.ascontiguousarray(X.T), device="cuda"),
                               torch.tensor(np.ascontiguousarray(C.T), device="cuda"))
    assert a.cpu().tolist() == [3, 3, 3, 0, 5]
    assert np.all(d.cpu().numpy() == 0.0)


def test_findClusterAssignments_dense_branch():
    from sparsifiedkmeans_amd.kmeans import findClusterAssignments
    rng = np.random.default_rng(11)
    X = rng.standard_normal((40, 333))
    C = rng.standard_normal((40, 7))
    a, d = findClusterAssignments(X, C)
    assert a.min() >= 1 and a.max() <= 7
    _check_assign(X, C, a - 1, d)
    with pytest.raises(ValueError):
        findClusterAssignments(X, C[:-1])


@pytest.mark.parametrize("p,n,K", [(64, 1000, 10), (100, 5000, 100), (1024, 700, 3), (7, 3, 5)])
def test_dense_accumulate_sums_and_counts(p, n, K):
    import torch
    from sparsifiedkmeans_amd.engine import dense_accumulate_device, torch_context
    rng = np.random.default_rng(n + K)
    X = rng.standard_normal((n, p))
    a = rng.integers(0, K, n).astype(np.int32)
    if K > 2:
        a[a == 1] = 0                                       # an empty cluster
    sums = torch.zeros((K, p), dtype=torch.float64, device="cuda")
    cnt = torch.zeros(K, dtype=torch.float64, device="cuda")
    ctx = torch_context()
    half = n // 2                                           # two chunks accumulate into the same tables
    for lo, hi in ((0, half), (half, n)):
        if hi > lo:
            dense_accumulate_device(ctx, torch.tensor(X[lo:hi], device="cuda"), torch.tensor(a[lo:hi], device="cuda"),
                                    sums, cnt)
    ref = np.zeros((K, p))
    np.add.at(ref, a, X)
    assert np.array_equal(cnt.cpu().numpy(), np.bincount(a, minlength=K).astype(np.float64))
    assert np.allclose(sums.cpu().numpy(), ref, rtol=1e-12, atol=1e-12)


def _gmm(p, n, K, seed):
    from sparsifiedkmeans_amd import synth
    X, centres, labels = synth.gmm_dense(p, n, K, seed)
    return X, centres, labels


@pytest.mark.parametrize("p,column_samples", [(64, False), (100, True)])
def test_driver_two_pass_outputs(p, column_samples):
    """[IDX,C,SUMD,D,OUTPUT,C2,IDX2,D2,SUMD2] = kmeans_sparsified(..., nargout 9) (kmeans_sparsified.m:525-568)."""
    from sparsifiedkmeans_amd.kmeans import kmeans_sparsified
    n, K = 3000, 4
    X, centres, labels = _gmm(p, n, K, 3)                   # p x n
    arg = X if column_samples else X.T
    start = centres + 0.05 * np.random.default_rng(1).standard_normal(centres.shape)   # no local-optimum lottery
    start = start if column_samples else start.T
    out = kmeans_sparsified(arg, K, Sparsify=True, SparsityLevel=0.25, SketchType="Hadamard", rng=7, nargout=9,
                            ColumnSamples=column_samples, MB_limit=0.5, Start=start)   # several chunks
    IDX, C, SUMD, D, OUTPUT, C2, IDX2, D2, SUMD2 = out
    Cc = C if column_samples else C.T                       # p x K
    C2c = C2 if column_samples else C2.T
    assert C2c.shape == (p, K) and IDX2.shape == (n,) and D2.shape == (n,) and SUMD2.shape == (K,)
    ref = _ref()
    assert np.allclose(C2c, ref.two_pass_centers(X, IDX - 1, K), rtol=1e-12, atol=1e-12)
    _check_assign(X, Cc, IDX2 - 1, D2)
    assert np.allclose(SUMD2, [np.sum(D[IDX2 == k + 1] ** 2) for k in range(K)], rtol=1e-12)
    # the second pass sees the unsampled data: its centres are the better estimate of the true means
    perm = [int(np.argmin(np.linalg.norm(centres - C2c[:, [k]], axis=0))) for k in range(K)]
    assert sorted(perm) == list(range(K))
    assert np.linalg.norm(C2c - centres[:, perm]) <= np.linalg.norm(Cc - centres[:, perm]) + 1e-9
    assert "TimeSecondPass_Centers" in OUTPUT
    # fewer outputs: nargout=6 stops after the centres
    out6 = kmeans_sparsified(arg, K, Sparsify=True, SparsityLevel=0.25, SketchType="Hadamard", rng=7, nargout=6,
                             ColumnSamples=column_samples, Start=start)
    assert len(out6) == 6 and np.allclose(out6[5], C2, rtol=1e-12, atol=1e-12)


def test_driver_two_pass_from_datafile(tmp_path):
    """'DataFile' path: recalculateAssignmentLargeFile (one streamed pass: means of the one-pass assignment and the
    dense re-assignment together)."""
    import warnings
    from sparsifiedkmeans_amd.kmeans import kmeans_sparsified
    p, n, K = 128, 2000, 3
    X, centres, labels = _gmm(p, n, K, 9)
    fn = str(tmp_path / "data.npy")
    np.save(fn, X.T)                                        # n x p on disk
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mem = kmeans_sparsified(X.T, K, Sparsify=True, SparsityLevel=0.2, SketchType="Hadamard", rng=21, nargout=9)
        dsk = kmeans_sparsified(fn, K, Sparsify=True, SparsityLevel=0.2, SketchType="Hadamard", rng=21, nargout=9,
                                MB_limit=0.25)
    assert np.array_equal(mem[0], dsk[0])
    assert np.allclose(mem[5], dsk[5], rtol=1e-12, atol=1e-12)
    assert np.array_equal(mem[6], dsk[6]) and np.allclose(mem[7], dsk[7], rtol=1e-9, atol=1e-12)
    assert np.allclose(mem[8], dsk[8], rtol=1e-9)
    assert "TimeSecondPass_Overall" in dsk[4] and "TimeSecondPass_JustRead" in dsk[4]
