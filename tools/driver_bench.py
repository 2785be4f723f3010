"""End-to-end timing of the host driver on an in-memory dense dataset (the reference's own use case)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.kmeans import kmeans_sparsified

p, n, K = 1024, int(float(sys.argv[1])) if len(sys.argv) > 1 else 500000, 50
X, centres, labels = synth.gmm_dense(p, n, K, seed=1)
X = np.ascontiguousarray(X.T)                      # n x p, 4 GB at n = 5e5
kmeans_sparsified(X[:5000], K, Sparsify=True, SparsityLevel=0.05, rng=0)     # warm-up
for nargout in (5, 9):
    t0 = time.time()
    out = kmeans_sparsified(X, K, Sparsify=True, SparsityLevel=0.05, rng=1, nargout=nargout, MaxIter=30)
    dt = time.time() - t0
    O = out[4]
    print(f"n={n} p={p} K={K} nargout={nargout}: total {dt:.2f} s; sketch+sample {O['TimeToSketch']:.2f} s; "
          f"init {O['TimeInitialization']:.2f} s; Lloyd {O['TimeAlgo_wo_initialization']:.2f} s "
          f"({int(O['iterations'][0])} iterations); objective {O['objectives'][0]:.4e}")
