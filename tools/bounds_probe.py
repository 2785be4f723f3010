"""Experiment: how many points (and 16-point steps) would triangle-inequality bounds carried across Lloyd iterations
(Hamerly-style: upper bound to the assigned centre, lower bound to all others, both moved by the centres' drift)
certify without any distance evaluation?  Bench workload at a reduced N; exact masked distances in torch."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context, mix_device
ctx = torch_context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000
K, p = 100, 1024
d = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234)
p2, s, gamma = d["p2"], d["s"], d["gamma"]
sh = Shard.from_device(ctx, p2, d["jc"], d["ir"], d["x"], nnz=d["nnz"])
g = torch.Generator(device="cuda"); g.manual_seed(234 + 17)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = d["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
c = mix_device(ctx, start.contiguous(), p2, d["sign"], 1.0, float(np.sqrt(np.float64(p2))))
eng = LloydEngine(sh, K, gamma)
ir = d["ir"][: n * s].view(n, s).long() & 0xffff
x = d["x"][: n * s].view(n, s)

TILES = [(0, 32), (32, 64), (64, 100)]
tl = [torch.empty(n, device='cuda', dtype=torch.float64) for _ in TILES]

JSET = None
minJ = torch.empty(n, device='cuda', dtype=torch.float64)

def dists(cent):
    cg = cent / gamma                                   # K x p2
    out1 = torch.empty(n, device="cuda", dtype=torch.float64); out2 = torch.empty_like(out1); a = torch.empty(n, device="cuda", dtype=torch.long)
    for i0 in range(0, n, 20000):
        i1 = min(n, i0 + 20000)
        cc = cg[:, ir[i0:i1]]                           # K x m x s
        dd = ((x[i0:i1][None] - cc) ** 2).sum(-1).sqrt().T   # m x K
        v, idx = torch.topk(dd, 2, dim=1, largest=False)
        out1[i0:i1], out2[i0:i1], a[i0:i1] = v[:, 0], v[:, 1], idx[:, 0]
        # per-tile lower bounds (tiles of 32 centroids; the carried remainder rides in tile 2): min over the tile's
        # centroids other than the assigned one
        dd2 = dd.clone(); dd2[torch.arange(i1 - i0), idx[:, 0]] = float("inf")
        for t, (lo, hi) in enumerate(TILES):
            tl[t][i0:i1] = dd2[:, lo:hi].min(dim=1).values
        if JSET is not None:
            minJ[i0:i1] = dd2[:, JSET].min(dim=1).values
    return out1, out2, a

prev = None
lbj = None
for it in range(14):
    if prev is not None:
        delta_ = ((c - prev[0]) / gamma).norm(dim=1)
        JSET = torch.topk(delta_, 8).indices
    d1, d2, a = dists(c)
    if prev is not None:
        pc, ub, lb, pa, ptl = prev
        delta = ((c - pc) / gamma).norm(dim=1)
        dmax = delta.max()
        U = ub + delta[pa]
        Lb = lb - dmax
        ok = U < Lb
        steps_ok = ok.view(-1, 16).all(dim=1).float().mean().item() if n % 16 == 0 else float("nan")
        changed = (a != pa).float().mean().item()
        print(f"iter {it}: drift max {dmax:.3g} mean {delta.mean():.3g}  d1 mean {d1.mean():.3g} gap mean {(d2-d1).mean():.3g}  points certified {ok.float().mean():.4f}  steps certified {steps_ok:.4f}  reassigned {changed:.5f}")
        # carried bounds (Hamerly): certified points keep the moved lower bound, the others get fresh ones
        lb = torch.where(ok, Lb, d2)
        # jumper version: the 8 largest movers bounded explicitly (exact distance here), the others by their own max drift
        rest = delta.clone(); rest[JSET] = 0
        okj = (U < lbj - rest.max()) & (U < minJ) & ~torch.isin(pa, JSET)
        print(f"          jumpers: top-8 drifts {[round(v, 1) for v in delta[JSET].tolist()]}, others' max {rest.max():.3g}; points {okj.float().mean():.4f}  steps {okj.view(-1, 16).all(dim=1).float().mean().item():.4f}")
        lbj = torch.where(okj, torch.minimum(lbj - rest.max(), minJ), d2)
        # per-tile version: a (step, tile) pair is skipped when all 16 points pass for that tile and the tile is not
        # the own tile of any of them -- unless every tile passes (then the whole step is skipped)
        passes = []
        for t, (lo, hi) in enumerate(TILES):
            passes.append(U < ptl[t] - delta[lo:hi].max())
        P = torch.stack(passes, 1)                                  # n x T
        own = torch.stack([(pa >= lo) & (pa < hi) for lo, hi in TILES], 1)
        allp = P.all(dim=1)
        skip_pt = (P & ~own) | allp[:, None]
        st = skip_pt.view(-1, 16, len(TILES)).all(dim=1).float().mean().item()
        print(f"          per-tile bounds: (step, tile) pairs skippable {st:.4f}; tile drifts {[round(delta[lo:hi].max().item(), 1) for lo, hi in TILES]}")
        newtl = [torch.where(P[:, t], ptl[t] - delta[lo:hi].max(), tl[t]) for t, (lo, hi) in enumerate(TILES)]
    else:
        lb = d2
        lbj = d2.clone()
        newtl = [x_.clone() for x_ in tl]
    prev = (c.clone(), d1.clone(), lb, a.clone(), newtl)
    eng.iterate(c)
    torch.cuda.synchronize()
