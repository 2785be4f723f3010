"""Dev aid: fused certified-screen call against the exact kernel on random fixed-stride shards; prints mismatches and the list length."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util import random_csc
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context
ctx = torch_context(0)
for (p, n, K, s) in ((1024, 20000, 100, 51), (1024, 20000, 10, 51), (1024, 20000, 32, 51), (1024, 20000, 7, 51), (256, 5000, 3, 13), (1024, 20000, 36, 51)):
    X = random_csc(p, n, s, seed=K)
    C = np.random.default_rng(K).standard_normal((p, K)) * 3.0
    sh = Shard.from_scipy(ctx, X)
    eng = LloydEngine(sh, K, 1.0, unbiased=False)
    c = torch.tensor(np.ascontiguousarray(C.T), device="cuda")
    eng.assign_step(c); torch.cuda.synchronize()
    a0 = eng.assign.cpu().numpy().copy(); d0 = eng.mind.cpu().numpy().copy()
    eng2 = LloydEngine(sh, K, 1.0, unbiased=False)
    eng2.assign_accumulate_step(c); torch.cuda.synchronize()
    a1 = eng2.assign.cpu().numpy(); d1 = eng2.mind.cpu().numpy()
    path, listed = eng2.last_path_info()
    bad = np.flatnonzero(a0 != a1)
    print((p, n, K, s), "path", path, "listed", listed, "assign mismatches", bad.size, "dist mismatches", int((d0 != d1).sum()), bad[:8], a0[bad[:8]], a1[bad[:8]])
