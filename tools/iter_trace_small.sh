python - <<'PY'
import os, sys, torch, numpy as np
sys.path.insert(0, ".")
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context, mix_device
ctx = torch_context(0); n = 12_500_000; K, p = 100, 1024
d = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234, chunk=131072)
g = torch.Generator(device="cuda"); g.manual_seed(251)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = d["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
c0 = mix_device(ctx, start.contiguous(), d["p2"], d["sign"], 1.0, 32.0)
for mode in ("sync",):
    sh = Shard.from_device(ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
    eng = LloydEngine(sh, K, d["gamma"]); c = c0.clone()
    for it in range(20):
        eng.iterate(c)
        if mode == "sync": torch.cuda.synchronize()
        if mode == "sync": print(mode, it, eng.last_screen_mode(), float(eng.out[1].item()) ** 0.5)
    torch.cuda.synchronize(); print(mode, "end", eng.last_screen_mode())
PY
