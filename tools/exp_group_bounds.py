"""Experiment (not part of the product): how much of the cold iterations could PER-TILE lower bounds (Yinyang-style groups of
32 centroids, carried one iteration) skip?  Headline generator at n = 4e6; per iteration: share of (point, tile) pairs whose
tile bound clears the point's upper bound, the same per 16 consecutive sample points, Hamerly's single bound, and the drift
(on the support) of the centroids -- maximum per tile against the median.  DESIGN.md 4.2f quotes the output."""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/sparsifiedkmeans_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device, torch_context
n = int(4e6); K, p = 100, 1024
ctx = torch_context(0)
data = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234, chunk=131072, order="block", noise=0.1)
shard = Shard.from_device(ctx, data["p2"], data["jc"], data["ir"], data["x"], nnz=data["nnz"])
p2, s, gamma = data["p2"], data["s"], data["gamma"]
g = torch.Generator(device="cuda"); g.manual_seed(251)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = data["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
centers = mix_device(ctx, start.contiguous(), p2, data["sign"], 1.0, float(np.sqrt(np.float64(p2)))).clone()
eng = LloydEngine(shard, K, gamma)
m = 40000
sel = torch.arange(0, n, n // m, device="cuda")[:m]
X = data["x"][: n * s].view(n, s)[sel].float()
R = data["ir"][: n * s].view(n, s)[sel].long() & 0xFFFF
def dist(C):
    Cm = C / gamma
    out = torch.empty((m, K), device="cuda")
    for k0 in range(0, K, 10):
        G = Cm[k0:k0 + 10].float()[:, R]
        out[:, k0:k0 + 10] = ((X[None] - G) ** 2).sum(-1).t()
    return out.sqrt()
tiles = [(0, 32), (32, 64), (64, 100)]
prev_c = prev_d = prev_a = None
prev_dr = None
ar = torch.arange(m, device="cuda")
for it in range(1, 11):
    cur = centers.clone()
    d = dist(cur)
    eng.iterate(centers)
    a = eng.assign[sel].long()
    if prev_c is not None:
        dr = ((cur - prev_c) / gamma).pow(2).topk(s, dim=1).values.sum(1).sqrt().float()   # per-centroid drift on the support
        ub = prev_d[ar, prev_a] + dr[prev_a]
        pd = prev_d.clone(); pd[ar, prev_a] = float("inf")
        skip = []
        for (lo, hi) in tiles:
            lbg = pd[:, lo:hi].min(1).values - dr[lo:hi].max()
            skip.append(lbg > ub)
        sk = torch.stack(skip, 1)
        # steps of 16 consecutive sample points stand in for steps (block order: neighbours share a cluster)
        st = sk.view(-1, 16, 3).all(1).float().mean().item()
        # the same test with the centroids REGROUPED by the drift of the previous iteration (32 largest movers in one group)
        if prev_dr is not None:
            order = torch.argsort(prev_dr, descending=True)
            grp = [order[0:32], order[32:64], order[64:100]]
            sk2 = torch.stack([(pd[:, gi].min(1).values - dr[gi].max()) > ub for gi in grp], 1)
            st2 = sk2.view(-1, 16, 3).all(1).float().mean().item()
            regrouped = f"regrouped by last drift: per (point, group) {sk2.float().mean().item():.3f} per (16-sample, group) {st2:.3f} max drift per group {[round(dr[gi].max().item(), 2) for gi in grp]}"
        else:
            regrouped = ""
        # Hamerly (one group): everything
        lball = pd.min(1).values - dr.max()
        print(f"it {it}: moved {(a != prev_a).float().mean().item():.3f}  per (point, tile) skippable {sk.float().mean().item():.3f}  per (16-sample, tile) {st:.3f}  Hamerly point skip {(lball > ub).float().mean().item():.3f}  max drift per tile {[round(dr[lo:hi].max().item(), 2) for lo, hi in tiles]} median drift {dr.median().item():.3f}  {regrouped}", flush=True)
        prev_dr = dr
    prev_c, prev_d, prev_a = cur, d, a
