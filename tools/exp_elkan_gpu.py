"""Experiment (GPU box; not part of the product): how many (point, centroid) pairs would PER-CENTROID lower bounds (Elkan's
k-means bounds, each eroded by its own centroid's drift on the support) leave to evaluate per iteration, against Hamerly's
single bound (what csrc/screen.hip carries today)?  Headline generator at n = 4e6 (the run follows the library's own
iterations), bounds tracked on a 40000-point sample.     python tools/exp_elkan_gpu.py [noise] [order] [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/sparsifiedkmeans_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device, torch_context
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
order = sys.argv[2] if len(sys.argv) > 2 else "block"
n = int(float(sys.argv[3])) if len(sys.argv) > 3 else int(4e6)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 14
K, p = 100, 1024
ctx = torch_context(0)
data = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234, chunk=131072, order=order, noise=noise)
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
ar = torch.arange(m, device="cuda")
prev_c = None
INF = float("inf")
for it in range(1, iters + 1):
    cur = centers.clone()
    D = dist(cur)
    eng.iterate(centers)
    a = D.argmin(1)
    if prev_c is not None:
        dr = ((cur - prev_c) / gamma).pow(2).topk(s, dim=1).values.sum(1).sqrt().float()   # per-centroid drift on the support
        lbk = lbk - dr[None, :]
        ub_loose = ub + dr[a_prev]
        lbh = lbh - dr.max()
        ham = ub_loose < lbh
        l2 = lbk.clone(); l2[ar, a_prev] = INF
        pass1 = ub_loose < l2.min(1).values
        ub_fresh = D[ar, a_prev]
        cand = (l2 < ub_fresh[:, None]) & ~pass1[:, None]
        nc = cand.sum(1).float()
        evals = (~pass1).sum().item() + cand.sum().item()
        # wave view: 16 consecutive sample points share a wave; its loop length is the largest candidate count among them
        wmax = nc.view(-1, 16).max(1).values
        fail = ~pass1
        print(f"it {it:2d}: moved {(a != a_prev).float().mean().item():.4f}  Hamerly pass {ham.float().mean().item():.3f}  Elkan pass(no eval) {pass1.float().mean().item():.3f} "
              f"(whole 16-groups {pass1.view(-1, 16).all(1).float().mean().item():.3f})  candidates/point mean {nc.mean().item():.2f} (failing points {nc[fail].mean().item() if fail.any() else 0:.2f}, max {int(nc.max().item())}, "
              f"mean of 16-group max {wmax.mean().item():.2f})  evaluations {evals / (m * K):.4f} of m*K  drift max {dr.max().item():.3f} median {dr.median().item():.4f}  "
              f"ub median {ub_fresh.median().item():.2f} 2nd-best median {l2.min(1).values.median().item():.2f}", flush=True)
        ev = cand.clone(); ev[ar, a_prev] |= fail
        lbk = torch.where(ev, D, lbk)
        ub = torch.where(fail, D[ar, a], ub_loose)
        lbh = torch.where(ham, lbh, D.topk(2, dim=1, largest=False).values[:, 1])
    else:
        lbk = D.clone(); ub = D[ar, a]; lbh = D.topk(2, dim=1, largest=False).values[:, 1]
    a_prev = a
    prev_c = cur
