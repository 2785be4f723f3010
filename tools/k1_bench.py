import time, torch, numpy as np, sys
sys.path.insert(0, '/root/repo')
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context
ctx = torch_context(0)
n = 20_000_000
d = synth.sparsified_gmm_device(ctx, 1024, n, n, 0, 100, 0.05, seed=1)
sh = Shard.from_device(ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
for K in (1, 2, 3, 8, 16):
    eng = LloydEngine(sh, K, d["gamma"])
    c = torch.randn((K, 1024), dtype=torch.float64, device="cuda")
    for _ in range(2): eng.assign_step(c)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): eng.assign_step(c)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"K={K}: assign_step {dt*1e3:.2f} ms  -> {n*51*10/dt/1e12:.2f} TB/s of X")
