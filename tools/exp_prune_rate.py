"""Experiment (not part of the product): how many (point, centroid-group) pairs would a POINT-granular two-phase screen
finish after A of the 13 rounds?  Runs the headline generator at a reduced n, iterates the engine, and for a sample of
points compares partial masked distances (first 4A entries) with the hint the library would carry (previous exact distance
to the previous centroid + that centroid's drift on the support)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import synth  # noqa: E402
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device, torch_context  # noqa: E402

n = int(float(os.environ.get("N", "4e6")))
order = os.environ.get("ORDER", "block")
K, p = 100, 1024
ctx = torch_context(0)
data = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234, chunk=131072, order=order, noise=0.1)
shard = Shard.from_device(ctx, data["p2"], data["jc"], data["ir"], data["x"], nnz=data["nnz"])
p2, s, gamma = data["p2"], data["s"], data["gamma"]
g = torch.Generator(device="cuda")
g.manual_seed(234 + 17)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = data["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
centers = mix_device(ctx, start.contiguous(), p2, data["sign"], 1.0, float(np.sqrt(np.float64(p2)))).clone()
eng = LloydEngine(shard, K, gamma)
m = 40000
sel = torch.arange(0, n, n // m, device="cuda")[:m]
X = data["x"][: n * s].view(n, s)[sel].float()
R = data["ir"][: n * s].view(n, s)[sel].long() & 0xFFFF
print("centers", tuple(centers.shape), "s", s, "gamma", gamma)
Kdim = 0 if centers.shape[0] == K else 1


def dist2(C, upto):
    # C: K x p2 (row k = centroid k); masked squared distance over the first `upto` entries
    Cm = (C if Kdim == 0 else C.t()) / gamma
    out = torch.empty((m, K), device="cuda")
    for k0 in range(0, K, 10):
        G = Cm[k0:k0 + 10].float()[:, R[:, :upto]]          # 10 x m x upto
        out[:, k0:k0 + 10] = ((X[None, :, :upto] - G) ** 2).sum(-1).t()
    return out


prev_c = None
prev_a = None
prev_ub = None
for it in range(1, 11):
    cur = centers.clone()
    full = dist2(cur, s)
    if it == 1:
        eng.iterate(centers)
        dm = eng.mind[sel].float()
        mine = full.min(1).values.sqrt()
        print("formula check: max rel diff of min distance", float(((dm - mine).abs() / (dm + 1e-30)).max()))
    else:
        eng.iterate(centers)
    a = eng.assign[sel].long()
    if prev_a is not None:
        Cm = cur if Kdim == 0 else cur.t()
        Pm = prev_c if Kdim == 0 else prev_c.t()
        d = (Cm - Pm) / gamma
        top = d.pow(2).topk(s, dim=1).values.sum(1).sqrt()              # drift on the support (per centroid)
        hint_cons = (prev_ub + top[prev_a].float())                     # carried bound + rigorous drift
        full_d2 = d.pow(2).sum(1)
        hint_lib = (prev_ub ** 2 + (2.0 * s / p2) * full_d2[prev_a].float()).sqrt()   # what k_bounds_steps writes
        hint_best = full[torch.arange(m), prev_a].sqrt()                # the tightest a hint could be
        moved = float((a != prev_a).float().mean())
        line = [f"it {it}: moved {moved:.3f}"]
        for A in (3, 5, 7):
            part = dist2(cur, 4 * A)
            part[torch.arange(m), prev_a] = float("inf")                # the hinted centroid itself is always evaluated
            for name, h, fac in (("cons x1.0", hint_cons, 1.0), ("lib x1.5", hint_lib, 1.5), ("best x1.5", hint_best, 1.5),
                                 ("best x1.0", hint_best, 1.0)):
                alive = part <= (h * h)[:, None] * fac                  # pairs that survive phase A (kernel: m2 >= 1.5 h^2)
                t32 = torch.stack([alive[:, 0:32].any(1), alive[:, 32:64].any(1), alive[:, 64:100].any(1)], 1)
                own_tile = torch.clamp(prev_a // 32, max=2)
                other = torch.ones_like(t32)
                other[torch.arange(m), own_tile] = False
                q_other = float((t32 & other).float().sum() / other.float().sum())
                g8 = torch.stack([alive[:, j:j + 8].any(1) for j in range(0, 96, 8)], 1).float().mean()
                # 16 consecutive sample points ~ a step's worth of neighbours only in block order; report per point
                line.append(f"A={A} {name}: pair {float(alive.float().mean()):.3f} grp8 {float(g8):.3f} tile(other) {q_other:.3f}")
        print("\n   ".join(line), flush=True)
    prev_c, prev_a = cur, a
    prev_ub = full[torch.arange(m), a].sqrt()
