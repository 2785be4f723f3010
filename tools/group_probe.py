"""Experiment (GPU box): how much of the screen's (point, centroid) work in the COLD iterations of the headline run
could bounds with finer granularity than one lower bound per point remove?  Compared, per iteration of a run from the
bench's sample start, on exact masked distances evaluated in torch at a reduced N:

  hamerly   one lower bound per point, eroded by the largest drift of any centroid (what the library carries today)
  tiles3    one lower bound per point and screen tile (centroids 0-31 / 32-63 / 64-99), eroded by the tile's largest drift
  groupsG   G groups of centroids formed once after iteration 1 by the size of the drift seen then (movers together)
  elkan     one lower bound per (point, centroid): the floor of what carried bounds can do

For each: the share of (point, centroid) evaluations still needed, for points and for 16-point steps (a step needs a
group if any of its points does).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import synth  # noqa: E402
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device, torch_context  # noqa: E402

ctx = torch_context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000
order = sys.argv[2] if len(sys.argv) > 2 else "block"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
K, p = 100, 1024
d = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234, order=order)
p2, s, gamma = d["p2"], d["s"], d["gamma"]
sh = Shard.from_device(ctx, p2, d["jc"], d["ir"], d["x"], nnz=d["nnz"])
g = torch.Generator(device="cuda")
g.manual_seed(234 + 17)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = d["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
c = mix_device(ctx, start.contiguous(), p2, d["sign"], 1.0, float(np.sqrt(np.float64(p2))))
eng = LloydEngine(sh, K, gamma)
ir = d["ir"][: n * s].view(n, s).long() & 0xffff
x = d["x"][: n * s].view(n, s)
ar = torch.arange(n, device="cuda")


def all_dists(cent):
    cg = cent / gamma
    dd = torch.empty((n, K), device="cuda", dtype=torch.float64)
    for i0 in range(0, n, 20000):
        i1 = min(n, i0 + 20000)
        cc = cg[:, ir[i0:i1]]
        dd[i0:i1] = ((x[i0:i1][None] - cc) ** 2).sum(-1).sqrt().T
    return dd


def group_min(dd_other, groups):
    return torch.stack([dd_other[:, gi].min(dim=1).values for gi in groups], 1)          # n x G


def steps_share(need_pt_group, sizes):
    """need: n x G bool; sizes: G.  share of (point, centroid) work needed at point and at 16-point-step granularity"""
    w = sizes.double() / sizes.sum()
    pt = (need_pt_group.double() * w).sum(1).mean().item()
    m = n // 16 * 16
    st = (need_pt_group[:m].view(-1, 16, need_pt_group.shape[1]).any(dim=1).double() * w).sum(1).mean().item()
    return pt, st


schemes = {}          # name -> dict(groups=[idx tensors], lb=n x G)
prev_c = prev_a = ub = None
elk = None
for it in range(1, iters + 1):
    dd = all_dists(c)
    v, a = dd.min(dim=1)
    dd_o = dd.clone()
    dd_o[ar, a] = float("inf")
    if it == 1:
        schemes["hamerly"] = dict(groups=[torch.arange(K, device="cuda")])
        schemes["tiles3"] = dict(groups=[torch.arange(0, 32, device="cuda"), torch.arange(32, 64, device="cuda"),
                                         torch.arange(64, 100, device="cuda")])
        for sc in schemes.values():
            sc["lb"] = group_min(dd_o, sc["groups"])
        elk = dd.clone()
        print(f"iter 1: full evaluation (no bounds); obj {v.pow(2).sum().sqrt():.6g}")
    else:
        delta = ((c - prev_c) / gamma).norm(dim=1)
        U = ub + delta[prev_a]
        moved = (a != prev_a).double().mean().item()
        srt = torch.sort(delta, descending=True).values
        print(f"iter {it}: reassigned {moved:.4f}; drift top8 {[round(t, 2) for t in srt[:8].tolist()]} "
              f"median {srt[K // 2]:.3g} p90 {srt[K // 10]:.3g}; mean d1 {v.mean():.3g} mean gap {(dd_o.min(1).values - v).mean():.3g}")
        if it == 2:
            # groups by the drift seen in the first update: the G-1 smallest groups hold the biggest movers
            orderk = torch.argsort(delta, descending=True)
            for G, cuts in (("groups4", [0, 4, 12, 36, 100]), ("groups8", [0, 2, 4, 8, 16, 32, 48, 72, 100]),
                            ("groups12", [0, 1, 2, 4, 6, 8, 12, 16, 24, 36, 52, 72, 100])):
                gs = [orderk[cuts[j]:cuts[j + 1]] for j in range(len(cuts) - 1)]
                schemes[G] = dict(groups=gs, lb=None)
        for name, sc in schemes.items():
            gs = sc["groups"]
            sizes = torch.tensor([len(gi) for gi in gs], device="cuda")
            if sc.get("lb") is None:                       # formed this iteration: everything is evaluated once
                sc["lb"] = group_min(dd_o, gs)
                print(f"    {name:9s} formed (full evaluation)")
                continue
            dg = torch.stack([delta[gi].max() for gi in gs])
            L = sc["lb"] - dg[None, :]
            need = ~(U[:, None] < L)                       # n x G
            # Hamerly's first refinement: tighten U to the exact own distance when anything is needed (one evaluation)
            pt, st = steps_share(need, sizes)
            fresh = group_min(dd_o, gs)
            # points re-evaluated for a group get that group's fresh bound; a point whose assignment changed gets all fresh
            chg = (a != prev_a)[:, None]
            sc["lb"] = torch.where(need | chg, fresh, L)
            print(f"    {name:9s} needed work: points {pt:.4f}  steps {st:.4f}   (groups' drift {[round(t, 2) for t in dg.tolist()]})")
        Le = elk - delta[None, :]
        need = ~(U[:, None] < Le)
        need[ar, prev_a] = False
        pt = need.double().mean().item()
        m = n // 16 * 16
        st = need[:m].view(-1, 16, K).any(dim=1).double().mean().item()
        print(f"    {'elkan':9s} needed work: points {pt:.4f}  steps {st:.4f}")
        elk = torch.where(need, dd, Le)
        elk[ar, a] = dd[ar, a]
    ub = v.clone()
    prev_c, prev_a = c.clone(), a.clone()
    eng.iterate(c)
    torch.cuda.synchronize()
