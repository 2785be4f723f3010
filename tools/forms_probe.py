import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from util import random_csc
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context
ctx = torch_context(0)
tot = {}
for seed in range(36):
    rng = np.random.default_rng(9000 + seed)
    p = int(rng.choice([64, 128, 256, 500, 1024])); s_ = int(rng.integers(1, min(p, 64) + 1))
    K = int(rng.choice([2, 3, 17, 33, 36, 40, 64, 68, 100])); n = int(rng.integers(200, 3000))
    X = random_csc(p, n, s_, seed=seed).tocsc()
    lab = (np.arange(n) * K) // n; cen0 = rng.standard_normal((p, K)) * 2.0
    for i in range(n):
        sl = slice(X.indptr[i], X.indptr[i + 1]); X.data[sl] = 0.3 * X.data[sl] + cen0[X.indices[sl], lab[i]]
    gam = s_ / p
    C = gam * cen0 + 0.05 * rng.standard_normal((p, K))
    eng = LloydEngine(Shard.from_scipy(ctx, X), K, gam)
    cd = torch.tensor(np.ascontiguousarray(C.T), device="cuda")
    row = []
    for it in range(7):
        eng.iterate(cd); torch.cuda.synchronize()
        m = eng.last_screen_mode(); row.append((m[0], m[4]))
    print(seed, p, s_, K, n, row)
