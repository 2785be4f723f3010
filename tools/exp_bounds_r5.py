"""Experiment (GPU box; not part of the product): sizes two ways of doing less arithmetic in the cold iterations of a run.

(1) GROUP BOUNDS at 16-point-step granularity: lower bounds per (point, centroid group) -- one group (Hamerly, what the
    library carries today), the screen's 32-centroid tiles, 16-centroid half tiles, 4-centroid pieces, single centroids
    (Elkan) -- each eroded by the largest drift inside its group; a step evaluates a group when any of its 16 points needs
    it (plus the group of the point's own centroid).  Static groups against groups re-formed by drift (migrating centroids
    together).  Printed: share of the (entry, centroid) work that is left.
(2) ENTRY ORDER of the screen copy: partial sums over a column's first 4 A entries are what the hinted two-phase screen
    compares with its hints; with the entries ordered by |x| descending the partial sums grow faster.  Printed: share of
    the (step, tile) pairs whose 16 points all clear 1.5 x hint^2 after A rounds, storage order against |x| order, for the
    library-style hint sqrt(ub^2 + (2 s / p) |dc|^2) and for the tightest possible one (the current distance).

    python tools/exp_bounds_r5.py [noise] [order] [n] [iters]
"""
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
g = torch.Generator(device="cuda"); g.manual_seed(234 + 17)          # bench.py's start
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = data["means"][lab] + noise * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
centers = mix_device(ctx, start.contiguous(), p2, data["sign"], 1.0, float(np.sqrt(np.float64(p2)))).clone()
eng = LloydEngine(shard, K, gamma)

NST = 2560                                                            # sampled 16-point steps (consecutive points each)
nsteps = n // 16
st0 = (torch.arange(NST, device="cuda") * (nsteps // NST)) * 16
sel = (st0[:, None] + torch.arange(16, device="cuda")[None, :]).reshape(-1)
m = sel.numel()
X = data["x"][: n * s].view(n, s)[sel].float()
R = data["ir"][: n * s].view(n, s)[sel].long() & 0xFFFF
ordr = torch.argsort(-X.abs(), dim=1)                                 # |x| descending
ROUNDS = (1, 2, 3, 4, 7)

def evaluate(C):
    """full squared distances [m, K] and partial sums after 4 A entries in storage order / |x| order: dict A -> [m, K]"""
    Cm = (C / gamma).float()
    D2 = torch.empty((m, K), device="cuda")
    Pst = {A: torch.empty((m, K), device="cuda") for A in ROUNDS}
    Psr = {A: torch.empty((m, K), device="cuda") for A in ROUNDS}
    P16 = {A: torch.empty((m, K), device="cuda") for A in ROUNDS}     # round 6: |x| order, x and c rounded to f16, f32 sums
    X16 = X.half().float()
    for k0 in range(0, K, 10):
        G = Cm[k0:k0 + 10][:, R]                                      # [10, m, s]
        T = (X[None] - G) ** 2
        D2[:, k0:k0 + 10] = T.sum(-1).t()
        Ts = torch.gather(T, 2, ordr[None].expand(T.shape[0], -1, -1))
        T16 = torch.gather((X16[None] - G.half().float()) ** 2, 2, ordr[None].expand(T.shape[0], -1, -1))
        for A in ROUNDS:
            Pst[A][:, k0:k0 + 10] = T[:, :, : 4 * A].sum(-1).t()
            Psr[A][:, k0:k0 + 10] = Ts[:, :, : 4 * A].sum(-1).t()
            P16[A][:, k0:k0 + 10] = T16[:, :, : 4 * A].sum(-1).t()
    return D2, Pst, Psr, P16

TILES = [torch.arange(0, 32, device="cuda"), torch.arange(32, 64, device="cuda"), torch.arange(64, 100, device="cuda")]
def groups_of(size, perm=None):
    idx = torch.arange(K, device="cuda") if perm is None else perm
    if size == 32:
        return [idx[0:32], idx[32:64], idx[64:100]]
    return [idx[a:a + size] for a in range(0, K, size)]

class Scheme:
    """per-centroid lower bounds updated at GROUP granularity (see the module comment)"""
    def __init__(self, name, size, regroup):
        self.name, self.size, self.regroup = name, size, regroup    # regroup: "static" | "drift" (re-formed every iteration by drift)
        self.lbk = None
    def start(self, D, a):
        self.lbk = D.clone(); self.ub = D[ar, a].clone(); self.a = a.clone()
    def step(self, D, dr):
        perm = torch.argsort(dr, descending=True) if self.regroup == "drift" else None
        grps = [torch.arange(K, device="cuda")] if self.size >= K else groups_of(self.size, perm)
        ub = self.ub + dr[self.a]
        own = torch.zeros((m, K), dtype=torch.bool, device="cuda"); own[ar, self.a] = True
        work = 0.0
        need_any = torch.zeros(m, dtype=torch.bool, device="cuda")
        needs = []
        for gi in grps:
            l = self.lbk[:, gi] - dr[gi].max()
            l = torch.where(own[:, gi], torch.full_like(l, float("inf")), l)
            nd = l.min(1).values <= ub
            needs.append(nd); need_any |= nd
        newlb = self.lbk.clone()
        for gi, nd in zip(grps, needs):
            ownin = own[:, gi].any(1)
            nd_pt = nd | (need_any & ownin)                         # the group of the point's own centroid rides along
            nd_st = nd_pt.view(-1, 16).any(1)                       # a step evaluates a group when any of its points needs it
            work += nd_st.float().sum().item() * gi.numel()
            ev = nd_st[:, None].expand(-1, 16).reshape(-1)
            newlb[:, gi] = torch.where(ev[:, None], D[:, gi], self.lbk[:, gi] - dr[gi].max())
        evp = need_any.view(-1, 16).any(1)[:, None].expand(-1, 16).reshape(-1)
        # evaluated points: exact argmin over what was evaluated (proved to contain the winner); others keep a, ub eroded
        Dm = torch.where(newlb == D, D, torch.full_like(D, float("inf")))     # (evaluated entries)
        a_new = torch.where(evp, Dm.argmin(1), self.a)
        self.ub = torch.where(evp, D[ar, a_new], ub)
        self.a = a_new
        self.lbk = newlb
        return work / (NST * K), (~need_any).float().mean().item(), (~need_any).view(-1, 16).all(1).float().mean().item()

ar = torch.arange(m, device="cuda")
schemes = [Scheme("hamerly", K, "static"), Scheme("tile32", 32, "static"), Scheme("tile32/drift", 32, "drift"),
           Scheme("half16", 16, "static"), Scheme("half16/drift", 16, "drift"), Scheme("piece4", 4, "static"),
           Scheme("piece4/drift", 4, "drift"), Scheme("elkan", 1, "static")]
prev_c = None
for it in range(1, iters + 1):
    cur = centers.clone()
    D2, Pst, Psr, P16 = evaluate(cur)
    D = D2.sqrt()
    eng.iterate(centers)
    a_true = D.argmin(1)
    if prev_c is None:
        for sc in schemes: sc.start(D, a_true)
        print(f"it {it}: full evaluation", flush=True)
    else:
        diff = (cur - prev_c) / gamma
        dr = diff.pow(2).topk(s, dim=1).values.sum(1).sqrt().float() * (1 + 1e-6)
        full2 = diff.pow(2).sum(1).float()
        line = [f"it {it:2d}: moved {(a_true != a_prev).float().mean().item():.4f} drift max {dr.max().item():.1f} med {dr.median().item():.2f} | work left:"]
        for sc in schemes:
            w, ppass, spass = sc.step(D, dr)
            bad = int((sc.a != a_true).sum().item())                # every scheme must reproduce the exact assignment
            line.append(f"{sc.name} {w:.3f}" + (f" (!{bad} wrong)" if bad else ""))
        print(" ".join(line), flush=True)
        # (2) hinted early finish per (step, tile): second smallest partial sum of the tile against 1.5 hint^2
        ub_prev = D_prev[ar, a_prev]
        hint_lib2 = ub_prev ** 2 + (2.0 * s / p2) * full2[a_prev]
        hint_best2 = D2[ar, a_prev]
        xnorm = X.pow(2).sum(1).sqrt()
        cmax = float((cur / gamma).abs().max().item())
        out = []
        for A in ROUNDS:
            row = [f"A={A}:"]
            for nm, P in (("storage", Pst[A]), ("sorted", Psr[A])):
                for hn, h2 in (("lib", hint_lib2), ("best", hint_best2)):
                    fin = []
                    for gi in TILES:
                        m2 = P[:, gi].topk(2, dim=1, largest=False).values[:, 1]
                        fin.append((m2 >= 1.5 * h2).view(-1, 16).all(1))
                    row.append(f"{nm}/{hn} {torch.stack(fin, 1).float().mean().item():.3f}")
            # round 6: the same question asked of an f16 first phase -- x and c rounded to f16 (f32 sums), the
            # input rounding folded into the bound: sqrt(P_true) >= sqrt(P16) - 2^-11 (|x|_A + |c|_A) with |x|_A <= the point's
            # norm and |c|_A <= sqrt(4 A) max|c| (what the kernel would have at hand)
            e16 = 2.0 ** -11 * (xnorm + float(np.sqrt(4.0 * A)) * cmax)
            for hn, h2 in (("lib", hint_lib2), ("best", hint_best2)):
                fin = []
                for gi in TILES:
                    m2 = P16[A][:, gi].topk(2, dim=1, largest=False).values[:, 1]
                    lb = (m2.sqrt() - e16).clamp_min(0.0) ** 2
                    fin.append((lb >= 1.5 * h2).view(-1, 16).all(1))
                row.append(f"f16/{hn} {torch.stack(fin, 1).float().mean().item():.3f}")
            out.append(" ".join(row))
        print("      early-finished (step, tile) pairs  " + "  |  ".join(out), flush=True)
        # (3) what a partial sum is worth as a carried LOWER bound: sqrt(min over the other centroids) -- median / 1 % quantile over
        # the points, and the median over 16-point steps of the step's smallest (a block summary keeps the minimum of 1024)
        out = []
        own = torch.zeros((m, K), dtype=torch.bool, device="cuda"); own[ar, a_true] = True
        for A in ROUNDS:
            for nm, P in (("storage", Pst[A]), ("sorted", Psr[A])):
                lbp = torch.where(own, torch.full_like(P, float("inf")), P).min(1).values.sqrt()
                out.append(f"A={A} {nm} {lbp.median().item():.1f}/{lbp.quantile(0.01).item():.1f}/{lbp.view(-1, 16).min(1).values.median().item():.1f}")
        print(f"      partial sums as lower bounds (median / 1 pct / median of step minima; ub median {D[ar, a_true].median().item():.1f}): " + "  ".join(out), flush=True)
    D_prev, a_prev, prev_c = D, a_true, cur
