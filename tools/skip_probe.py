"""What share of the exact pass could be skipped in the converged regime if clusters whose membership did not change
(this call and the one before) were not recomputed?  Headline workload, 40 iterations from the sample start."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device, torch_context
order = sys.argv[1] if len(sys.argv) > 1 else "block"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
ctx = torch_context(0)
p, K = 1024, 100
data = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234, chunk=131072, order=order)
shard = Shard.from_device(ctx, data["p2"], data["jc"], data["ir"], data["x"], nnz=data["nnz"])
g = torch.Generator(device="cuda"); g.manual_seed(234 + 17)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = data["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
c = mix_device(ctx, start.contiguous(), data["p2"], data["sign"], 1.0, 32.0)
eng = LloydEngine(shard, K, data["gamma"])
prev = None; prev_touched = torch.ones(K, dtype=torch.bool, device="cuda")
for it in range(40):
    eng.iterate(c, want_mind=False)
    a = eng.assign
    if prev is not None:
        ch = a != prev
        touched = torch.zeros(K, dtype=torch.bool, device="cuda")
        touched[a[ch].long()] = True; touched[prev[ch].long()] = True
        nk = torch.bincount(a, minlength=K)
        skippable = ~touched & ~prev_touched
        print(it, "movers", int(ch.sum()), "touched clusters", int(touched.sum()), "skippable clusters", int(skippable.sum()),
              "share of points in them %.3f" % (float(nk[skippable].sum()) / n), flush=True)
        prev_touched = touched
    prev = a.clone()
