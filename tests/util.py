"""Shared input builders for the parity tests (seeded; sizes the oracle finishes in seconds)."""
import numpy as np
import scipy.sparse as sp


def random_csc(p, n, nnz_per_col, seed, ragged=False, empty_cols=(), dtype=np.float64):
    """p x n CSC with ascending rows.  ragged: per-column count uniform in [0, 2*nnz_per_col]."""
    rng = np.random.default_rng(seed)
    indptr = [0]
    idx, val = [], []
    for i in range(n):
        if i in empty_cols:
            c = 0
        elif ragged:
            c = int(rng.integers(0, min(p, 2 * nnz_per_col) + 1))
        else:
            c = min(p, nnz_per_col)
        rows = np.sort(rng.choice(p, c, replace=False))
        idx.append(rows)
        val.append(rng.standard_normal(c) * 3.0)
        indptr.append(indptr[-1] + c)
    X = sp.csc_matrix((np.concatenate(val) if val else np.zeros(0), np.concatenate(idx).astype(np.int64)
                       if idx else np.zeros(0, np.int64), np.array(indptr, np.int64)), shape=(p, n))
    return X


def parts(X):
    return X.indptr.astype(np.uint64), X.indices.astype(np.uint64), X.data.astype(np.float64)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """numpy restatement of the counter-based generator of csrc/sample.hip (arrays of uint32)."""
    c0, c1, c2, c3 = (np.asarray(v, np.uint64) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M
        n1 = p1 & M
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M
        n3 = p0 & M
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return c0, c1, c2, c3


def sample_rows_reference(seed, col0, n, p2, s):
    """Selection sampling (Knuth Algorithm S) exactly as k_sample_rows draws it: [n, s] ascending rows."""
    cols = np.arange(col0, col0 + n, dtype=np.uint64)
    out = np.zeros((n, s), np.int64)
    taken = np.zeros(n, np.int64)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for r0 in range(0, p2, 4):
        u = philox4x32_10(cols & np.uint64(0xFFFFFFFF), cols >> np.uint64(32), np.full(n, r0 >> 2), np.zeros(n), k0, k1)
        for q in range(4):
            r = r0 + q
            if r >= p2:
                break
            t = (u[q] * np.uint64(p2 - r)) >> np.uint64(32)
            take = (t < (s - taken).astype(np.uint64)) & (taken < s)
            idx = np.flatnonzero(take)
            out[idx, taken[idx]] = r
            taken[idx] += 1
    assert np.all(taken == s)
    return out


def replay_driver_products(oracle, X, gamma_opt, seed, drop_zeros=True):
    """The random products kmeans_sparsified(X.T, K, Sparsify=True, SketchType='Hadamard', rng=seed) draws, replayed
    on the host with the oracle's transform: returns (Y scipy CSC p2 x n, sign vector d, s, p2, gamma_used).
    X is p x n (points as columns); p must make 16 <= p2 <= 16384 (the fused device sparsifier's range)."""
    from sparsifiedkmeans_amd import synth

    p, n = X.shape
    p2 = 1 << int(np.ceil(np.log2(p)))
    rng = np.random.default_rng(seed)
    d = np.sign(rng.standard_normal(p2))
    d[d == 0] = 1
    sample_seed = int(rng.integers(0, 2**63 - 1))
    Xm = oracle.mix(X, d, p2)
    s = synth.small_p_of(gamma_opt, p2)
    rows = sample_rows_reference(sample_seed, 0, n, p2, s)
    vals = Xm[rows, np.arange(n)[:, None]] / (np.float64(s) / np.float64(p2))
    Y = sp.csc_matrix((vals.ravel(), rows.ravel(), np.arange(0, (n + 1) * s, s)), shape=(p2, n))
    if drop_zeros:
        Y.eliminate_zeros()                                  # sparse() drops exact zeros (randsample_fixedNumberEntries.m:62)
    return Y, d, s, p2, s / p


def mix_start(oracle, S, d, p2):
    """centers = mix(start) (kmeans_sparsified.m:406): S is K x p in the original space -> p2 x K.  No (1+2eps)
    pre-scale here (that belongs to the data, :292)."""
    K, p = S.shape
    Z = np.zeros((p2, K))
    Z[:p] = S.T
    return oracle.fwht(Z * d[:, None]) / np.sqrt(np.float64(p2))


def set_switch(monkeypatch, ctx, name, on=True):
    """Toggle one of the library's SPKM_* A/B switches for the rest of the test: the environment variable AND the
    context's copy (the library reads its switches once, in spkm_ctx_create; spkm_ctx_reload_switches).  conftest's
    autouse fixture restores both after the test.  monkeypatch = None: plain os.environ (module-scoped fixtures)."""
    import os

    if monkeypatch is not None:
        if on:
            monkeypatch.setenv(name, "1")
        else:
            monkeypatch.delenv(name, raising=False)
    elif on:
        os.environ[name] = "1"
    else:
        os.environ.pop(name, None)
    ctx.reload_switches()


def mnist_like_pixels(n=60000, K=10, seed=3):
    """BASELINE.json config 3 by shape and value type (MNIST itself is not available offline): n x 784 uint8 "digit-like"
    images -- per class a stroke prototype on the 28 x 28 grid (three thick random strokes), per image a random gain, a
    +-1-pixel shift and pixel noise on the lit pixels; two thirds of the pixels are exactly 0, the classes overlap
    (K-means finds ~0.7 of the planted labels).  Returns (X uint8 [n, 784], labels [n])."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:28, 0:28]
    protos = np.zeros((K, 28, 28))
    for k in range(K):
        img = np.zeros((28, 28))
        for _ in range(3):
            a, b = rng.uniform(6, 22, 2), rng.uniform(6, 22, 2)
            for t in np.linspace(0, 1, 40):
                c = a + t * (b - a)
                img = np.maximum(img, np.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2) / (2 * 1.3 ** 2)))
        protos[k] = img
    labels = rng.integers(0, K, n)
    X = np.empty((n, 784), np.uint8)
    for c0 in range(0, n, 10000):
        lab = labels[c0:c0 + 10000]
        m = lab.size
        im = protos[lab]
        dy, dx = rng.integers(-1, 2, m), rng.integers(-1, 2, m)
        for sh in range(-1, 2):
            for sw in range(-1, 2):
                sel = (dy == sh) & (dx == sw)
                if sel.any():
                    im[sel] = np.roll(np.roll(im[sel], sh, axis=1), sw, axis=2)
        gain = rng.uniform(0.6, 1.0, (m, 1, 1))
        px = 255.0 * gain * im + 12.0 * rng.standard_normal((m, 28, 28)) * (im > 0.05)
        X[c0:c0 + m] = np.clip(np.round(px), 0, 255).astype(np.uint8).reshape(m, 784)
    return X, labels
