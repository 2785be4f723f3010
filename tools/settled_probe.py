"""GPU box: per-iteration counters of a lazy run on one shard (what the settled iterations' kernels actually work on):
listed points after the bounds test, points sent to the exact list, movers, with the wall time of each iteration.
    python tools/settled_probe.py [N] [iterations] [order]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_lib  # noqa: E402,F401  (SPKM_AB_LIB: another build of the library)
from sparsifiedkmeans_amd import synth                                   # noqa: E402
from sparsifiedkmeans_amd.engine import Context, LloydEngine, Shard, mix_device  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_500_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
order = sys.argv[3] if len(sys.argv) > 3 else "block"
p, K, gam0, s = 1024, 100, 0.05, 51
ctx = Context()
d = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, gam0, seed=234, order=order, layout="records")
p2, gamma = d["p2"], d["gamma"]
shard = Shard.from_records(ctx, p2, n, d["s"], d["rec"], d["ir_bits"])
shard.reset_policy()
shard.set_lazy_stats(True)
g = torch.Generator(device="cuda")
g.manual_seed(234 + 17)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = d["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
c = mix_device(ctx, start.contiguous(), p2, d["sign"], 1.0, float(np.sqrt(np.float64(p2))))
eng = LloydEngine(shard, K, gamma)
print("iter   ms     form listed ambig early skipped16 sums ptmode | streamed")
for it in range(iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c_before = c.clone() if os.environ.get("SPKM_PROBE_CHANGED") else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(c, want_mind=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    changed = int((c_before != c).any(dim=0 if c.shape[0] != K else 1).sum().item()) if c_before is not None else -1
    m = eng.last_screen_mode()
    print(f"{it:3d} {ms:8.3f} r{eng.last_screen_rounds()[0]:<2d} {m[0]:2d} {m[1]:8d} {m[2]:8d} {m[3]:8d} {m[4]:9d} {m[6]:2d} {m[7]:2d} | {eng.exact_pass_points()[1]}" + (f" changed centroids {changed}" if changed >= 0 else ""))
