"""Experiment (CPU, numpy; not part of the product): how many (point, centroid) pairs would PER-CENTROID lower bounds
(Elkan's k-means bounds, each eroded by its own centroid's drift on the support) leave to evaluate per iteration, against
Hamerly's single bound (what csrc/screen.hip carries today)?  Benchmark generator shape (p = 1024, K = 100, s = 51) at
n = 20000, sample start with duplicates, noise 0.1 (headline) and 1.5 (the `overlap` regime).
    python tools/exp_elkan_cpu.py [noise] [n]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import synth
from oracle import oracle as O

noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
p, K, gam0 = 1024, 100, 0.05
rng = np.random.default_rng(234)
centres = rng.standard_normal((p, K))
labels = (np.arange(n) * K) // n
X = centres[:, labels] + noise * rng.standard_normal((p, n))
d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
Xm = O.fwht(X * d[:, None]) / np.sqrt(p)
s = synth.small_p_of(gam0, p)
Y = synth.sparsify_dense(Xm, s, rng)
gamma = s / p
R = Y.indices.reshape(n, s); V = Y.data.reshape(n, s)
lab0 = rng.integers(0, K, K)
C = O.fwht((centres[:, lab0] + 0.1 * rng.standard_normal((p, K))) * d[:, None]) / np.sqrt(p) * gamma

def dist(C):
    Cg = (C / gamma).T          # K x p
    out = np.empty((n, K))
    for k in range(K):
        out[:, k] = np.sqrt(((V - Cg[k][R]) ** 2).sum(1))
    return out

jc, ir, x = Y.indptr.astype(np.uint64), Y.indices.astype(np.uint64), Y.data
prevC = None
ar = np.arange(n)
for it in range(1, 26):
    D = dist(C)
    a = D.argmin(1)
    if prevC is not None:
        dr = np.sqrt(np.sort(((C - prevC) / gamma) ** 2, axis=0)[-s:].sum(0))        # drift on the support, per centroid
        lbk = lbk - dr[None, :]                                                       # Elkan: each bound by ITS centroid's drift
        ub_loose = ub + dr[a_prev]
        lbh = lbh - dr.max()
        ham = ub_loose < lbh
        # Elkan, step 1: one lower bound over the others from the per-centroid table
        l2 = lbk.copy(); l2[ar, a_prev] = np.inf
        pass1 = ub_loose < l2.min(1)
        # step 2: refresh ub (1 evaluation for the points that failed), then candidates = others with lbk < ub_fresh
        ub_fresh = D[ar, a_prev]
        cand = (l2 < ub_fresh[:, None]) & ~pass1[:, None]
        nc = cand.sum(1)
        evals = (~pass1).sum() + cand.sum()
        step16 = pass1.reshape(-1, 16).all(1).mean() if n % 16 == 0 else float("nan")
        print(f"it {it:2d}: moved {np.mean(a != a_prev):.4f}  Hamerly pass {ham.mean():.3f}  Elkan pass (no eval) {pass1.mean():.3f} (whole 16-steps {step16:.3f})  "
              f"candidates/point: mean {nc.mean():.2f} (of failing points {nc[~pass1].mean() if (~pass1).any() else 0:.2f}, max {nc.max()})  "
              f"evaluations {evals / (n * K):.4f} of n*K  max drift {dr.max():.3f} median {np.median(dr):.4f}", flush=True)
        # bounds after the call: evaluated pairs get fresh values
        ev = cand.copy(); ev[ar, a_prev] |= ~pass1
        lbk = np.where(ev, D, lbk)
        ub = np.where(~pass1, D[ar, a], ub_loose)
        lbk[ar, a] = np.minimum(lbk[ar, a], D[ar, a])  # (own entry unused)
        l3 = lbk.copy(); l3[ar, a] = np.inf
        lbh = np.where(ham, lbh, np.sort(D, 1)[:, 1])
    else:
        lbk = D.copy(); ub = D[ar, a]; lbh = np.sort(D, 1)[:, 1]
    a_prev = a
    S, Cnt, nk = O.accumulate(p, n, K, jc, ir, x, a.astype(np.int32))
    prevC = C
    C = O.finalize_centers(S, Cnt, nk, gamma, C)
