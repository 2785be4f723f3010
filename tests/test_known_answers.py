"""Known-answer tests that do NOT route through oracle/: inputs are chosen so that every intermediate of the
reference's arithmetic is exactly representable, and the expected outputs are written down from the identities the
reference documents, in integer arithmetic:

  * hadamard(x) is the product with the Sylvester matrix, "same as hadamard(n)*x" (private/hadamard.c:8-11,17-23);
    applied twice it gives m*x;
  * SparseMatrixMinusCluster: dist(k,i) = norm( X(ind,i) - C(ind,k) ) over the stored rows ind of column i
    (private/SparseMatrixMinusCluster.c:1-8); the beta form expands the same square (:9-11,118-129);
  * SparseMatrixInnerProduct / SparseMatrixColumnNormSq: X(:,i)'*c and sum(X(:,i).^2)
    (private/SparseMatrixInnerProduct.c:1-9, private/SparseMatrixColumnNormSq.c:1-9);
  * [d,a] = min(dist,[],1): smallest value, FIRST index on ties (private/findClusterAssignments.m:169).

Each case runs against BOTH implementations: the CPU oracle (-m "not gpu") and the HIP path through the C ABI
(-m gpu).  This does not lift the oracle's "parity unpinned" status -- no output of the reference itself is involved
(the reference ships none and cannot be built here) -- but it is the one check in which neither side is the other's
yardstick.

Distance construction.  Rows come in adjacent pairs (2q, 2q+1) with weights (3, 4).  Point i stores m = g*g pairs;
on row r its value is base[r] + w_r*u_i and centroid k holds base[r] - w_r*t_k, with small integers / dyadic
fractions u_i, t_k.  Then x - c = w_r*(u_i + t_k) on every stored row, each pair contributes (9 + 16)*(u_i + t_k)^2
and the column sums to 25*m*(u_i + t_k)^2: every partial sum is an integer multiple of a power of two far below
2^53, and the square root is exactly 5*g*|u_i + t_k|.
"""
import numpy as np
import pytest
import scipy.sparse as sp


def sylvester(m):
    H = np.array([[1]], dtype=np.int64)
    while H.shape[0] < m:
        H = np.block([[H, H], [H, -H]])
    return H


class OracleBackend:
    name = "oracle"

    def __init__(self):
        from oracle import oracle as O

        O.build()
        self.O = O

    def hadamard(self, x):
        return self.O.fwht(x)

    def hadamard_pthreads(self, x):
        return self.O.fwht(x, threads=3)

    def dist(self, X, Cm, beta=None):
        jc, ir, xv = X.indptr.astype(np.uint64), X.indices.astype(np.uint64), X.data.astype(np.float64)
        if beta is not None:
            return self.O.dist_csc_beta(X.shape[1], jc, ir, xv, Cm.ravel(), beta)[None, :]
        return self.O.dist_csc(X.shape[0], X.shape[1], jc, ir, xv, Cm)

    def innerprod(self, X, c):
        return self.O.innerprod_csc(X.shape[1], X.indptr.astype(np.uint64), X.indices.astype(np.uint64), X.data, c)

    def colnormsq(self, X):
        return self.O.colnormsq_csc(X.shape[1], X.indptr.astype(np.uint64), X.data)

    def assign(self, X, Cm, gamma):
        a, d = self.O.assign(X.shape[0], X.shape[1], X.indptr.astype(np.uint64), X.indices.astype(np.uint64), X.data,
                             Cm, gamma or 0.0)
        return [(a, d)]


class HipBackend:
    name = "hip"

    def __init__(self):
        from sparsifiedkmeans_amd import ops
        from sparsifiedkmeans_amd.engine import torch_context

        self.ops, self.ctx = ops, torch_context(0)

    def hadamard(self, x):
        return self.ops.hadamard(x)

    def hadamard_pthreads(self, x):
        return self.ops.hadamard_pthreads(x)

    def dist(self, X, Cm, beta=None):
        return self.ops.SparseMatrixMinusCluster(X, Cm, beta)

    def innerprod(self, X, c):
        return self.ops.SparseMatrixInnerProduct(X, c)

    def colnormsq(self, X):
        return self.ops.SparseMatrixColumnNormSq(X)

    def assign(self, X, Cm, gamma):
        """every device route to an assignment: the exact kernels and the fused call (screen where the shard qualifies)"""
        import torch
        from sparsifiedkmeans_amd.engine import LloydEngine, Shard

        shard = Shard.from_scipy(self.ctx, X)
        K = Cm.shape[1]
        Ct = torch.tensor(np.ascontiguousarray(Cm.T), device="cuda")
        outs = []
        for fused in (False, True):
            eng = LloydEngine(shard, K, gamma if gamma else 1.0, unbiased=bool(gamma))
            if fused:
                eng.assign_accumulate_step(Ct)
            else:
                eng.assign_step(Ct)
            outs.append((eng.assign.cpu().numpy(), eng.mind.cpu().numpy()))
        return outs


@pytest.fixture(scope="module", params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def impl(request):
    return OracleBackend() if request.param == "oracle" else HipBackend()


@pytest.mark.parametrize("m", [2, 4, 8, 64, 1024, 4096])
def test_hadamard_is_the_integer_sylvester_product(impl, m):
    rng = np.random.default_rng(m)
    n = 7
    x = rng.integers(-1000, 1001, size=(m, n))
    want = sylvester(m) @ x                                     # int64, exact: |entries| <= 4096 * 1000
    for f in (impl.hadamard, impl.hadamard_pthreads):
        y = f(x.astype(np.float64))
        assert y.dtype == np.float64 and np.array_equal(y, want.astype(np.float64))
        assert np.array_equal(f(y), (m * x).astype(np.float64))  # H*H = m*I
    e = np.zeros((m, 1))
    e[m - 1, 0] = 1.0                                            # last column of H: the sign pattern itself
    assert np.array_equal(impl.hadamard(e)[:, 0], sylvester(m)[:, m - 1].astype(np.float64))


def _pythagorean_case(seed, p, n, K, g, dyadic):
    """returns X (p x n CSC), C (p x K), expected distances K x n, all exact (module docstring)"""
    rng = np.random.default_rng(seed)
    m = g * g
    assert 2 * m <= p
    scale = 0.125 if dyadic else 1.0
    u = rng.integers(-40, 41, size=n) * scale
    t = rng.integers(-40, 41, size=K) * scale
    if K >= 3:
        t[2] = t[0]                                              # duplicate centroid: an exact tie, first index must win
    base = rng.integers(-64, 65, size=p).astype(np.float64)
    w = np.where(np.arange(p) % 2 == 0, 3.0, 4.0)
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        q = np.sort(rng.choice(p // 2, m, replace=False))
        r = np.stack([2 * q, 2 * q + 1], axis=1).ravel()
        rows.append(r)
        vals.append(base[r] + w[r] * u[i])
        indptr.append(indptr[-1] + r.size)
    X = sp.csc_matrix((np.concatenate(vals), np.concatenate(rows).astype(np.int64), np.array(indptr, np.int64)), shape=(p, n))
    Cm = base[:, None] - w[:, None] * t[None, :]
    want = 5.0 * g * np.abs(u[None, :] + t[:, None])             # K x n, exact
    return X, Cm, want, u, t


@pytest.mark.parametrize("seed,p,n,K,g,dyadic", [(1, 64, 50, 1, 1, False), (2, 64, 300, 2, 2, False),
                                                 (3, 128, 300, 3, 3, True), (4, 256, 500, 7, 4, True),
                                                 (5, 1024, 400, 100, 5, True), (6, 1024, 257, 37, 1, False)])
def test_distances_on_pythagorean_columns(impl, seed, p, n, K, g, dyadic):
    X, Cm, want, _, _ = _pythagorean_case(seed, p, n, K, g, dyadic)
    got = impl.dist(X, Cm)
    assert got.shape == (K, n) and np.array_equal(got, want)


@pytest.mark.parametrize("beta", [1.0, 0.5, 2.0])
def test_beta_form_expands_the_same_square(impl, beta):
    """dist = sqrt( sum x^2 - 2*beta*x*c + c^2 ) over the stored rows (SparseMatrixMinusCluster.c:9-11,118-129);
    beta = 1 is the plain distance.  Evaluated here in small integers / dyadic fractions."""
    rng = np.random.default_rng(7)
    p, n = 64, 200
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        r = np.sort(rng.choice(p, 6, replace=False))
        rows.append(r)
        vals.append(rng.integers(-9, 10, size=6).astype(np.float64))
        indptr.append(indptr[-1] + 6)
    X = sp.csc_matrix((np.concatenate(vals), np.concatenate(rows).astype(np.int64), np.array(indptr, np.int64)), shape=(p, n))
    c = rng.integers(-9, 10, size=p).astype(np.float64)
    got = impl.dist(X, c[:, None], beta=beta)
    x = X.toarray()
    stored = np.zeros((p, n), bool)                              # a stored entry may be 0: it still contributes c^2
    for i in range(n):
        stored[X.indices[X.indptr[i]:X.indptr[i + 1]], i] = True
    tot = ((x * x - 2.0 * beta * x * c[:, None] + (c * c)[:, None]) * stored).sum(axis=0)   # small integers: exact in any order
    ok = tot >= 0                                                # (beta > 1 can make the "square" negative: NaN in the reference too)
    assert ok.sum() > n // 2
    # sqrt of an exactly known argument is correctly rounded on both sides
    assert np.array_equal(np.asarray(got).ravel()[ok], np.sqrt(tot[ok]))


def test_inner_product_and_norms_in_integers(impl):
    rng = np.random.default_rng(11)
    p, n = 512, 400
    X = sp.random(p, n, density=0.05, random_state=3, format="csc", data_rvs=lambda k: rng.integers(-50, 51, size=k).astype(np.float64))
    X.sort_indices()
    c = rng.integers(-50, 51, size=p).astype(np.float64)
    xi = X.toarray().astype(np.int64)
    ip, nx2 = impl.innerprod(X, c)
    assert np.array_equal(ip, (xi * c.astype(np.int64)[:, None]).sum(axis=0).astype(np.float64))
    assert np.array_equal(nx2, (xi * xi).sum(axis=0).astype(np.float64))
    assert np.array_equal(impl.colnormsq(X), (xi * xi).sum(axis=0).astype(np.float64))


@pytest.mark.parametrize("seed,p,n,K,g,gamma", [(21, 128, 2000, 3, 2, None), (22, 256, 3000, 10, 3, 0.25),
                                                (23, 1024, 3000, 100, 5, 0.0625), (24, 1024, 2000, 37, 4, 0.5),
                                                (25, 512, 1500, 20, 1, None)])
def test_argmin_with_first_index_ties(impl, seed, p, n, K, g, gamma):
    """findClusterAssignments(X, centers, [], gamma): distances to centers/gamma (:78), min with first index (:169).
    gamma is a power of two, so passing c*gamma makes centers/gamma the constructed c exactly."""
    X, Cm, want, u, t = _pythagorean_case(seed, p, n, K, g, True)
    a_want = np.argmin(want, axis=0)                             # numpy's argmin also returns the first minimum
    d_want = want[a_want, np.arange(n)]
    assert (want == d_want[None, :]).sum(axis=0).max() >= (2 if K >= 3 else 1)   # the case does contain exact ties
    for a, d in impl.assign(X, Cm * gamma if gamma else Cm, gamma):
        assert np.array_equal(a, a_want) and np.array_equal(d, d_want)
