import os, sys
import numpy as np, torch
ROOT = "/root/repo" if os.path.exists("/root/repo") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context
from util import parts, replay_driver_products, mix_start
ctx = torch_context(0)
K, p, gopt, n = 33, 512, 0.05, 6000
X, centres, labels = synth.gmm_dense(p, n, K, seed=5)
S = X[:, np.random.default_rng(1).choice(n, K, replace=False)].T
Y, d, s, p2, g = replay_driver_products(O, X, gopt, 3)
C0 = mix_start(O, S, d, p2)
shard = Shard.from_scipy(ctx, Y)
for lazy in (False, True):
    shard.reset_policy(); shard.set_lazy_stats(lazy)
    eng = LloydEngine(shard, K, g)
    c = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
    print("lazy", lazy)
    prev = None
    for it in range(20):
        used = c.cpu().numpy().T.copy()
        out = eng.iterate(c, want_mind=False).cpu().numpy()
        a = eng.assign.cpu().numpy()
        ra, rd = O.assign(p2, n, *parts(Y), used, g)
        Sm, Cnt, nk = O.accumulate(p2, n, K, *parts(Y), ra)
        red = eng.reduce.cpu().numpy(); pk = p2 * K
        es = np.abs(red[:pk].reshape(K, p2).T - Sm).max() / np.abs(Sm).max()
        ec = np.abs(red[pk:2*pk].reshape(K, p2).T - Cnt).max()
        mv = -1 if prev is None else int((a != prev).sum())
        print(it, "dff", float(np.sqrt(out[0])), "obj2", out[1], "assign_err", int((a != ra).sum()), "sum_relerr", es, "cnt_err", ec, "nk_err", int(np.abs(eng.nk.cpu().numpy()-nk).max()), "movers", mv, "mode", eng.last_screen_mode())
        prev = a
