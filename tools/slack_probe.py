"""GPU box: distribution of the carried bounds' slack (lb - ub; lb as a true bound, accumulated drift subtracted) at chosen
iterations of a lazy block-order run -- what decides how many blocks of 1024 points pass on their summary alone.
    python tools/slack_probe.py [N] [iterations,comma-separated]          (SPKM_AB_LIB: another build of the library)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_lib  # noqa
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import Context, LloydEngine, Shard, mix_device

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
marks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "9,17,40,90").split(",")]
p, K, gam0 = 1024, 100, 0.05
ctx = Context()
d = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, gam0, seed=234, order="block", layout="records")
shard = Shard.from_records(ctx, d["p2"], n, d["s"], d["rec"], d["ir_bits"])
shard.reset_policy(); shard.set_lazy_stats(True)
g = torch.Generator(device="cuda"); g.manual_seed(234 + 17)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = d["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
c = mix_device(ctx, start.contiguous(), d["p2"], d["sign"], 1.0, float(np.sqrt(np.float64(d["p2"]))))
eng = LloydEngine(shard, K, d["gamma"])
for it in range(max(marks) + 1):
    eng.iterate(c, want_mind=False)
    if it in marks:
        torch.cuda.synchronize()
        ub, lb, a = shard.debug_bounds()
        sl = lb - ub.astype(np.float64)
        nb = n // 1024
        bmin = sl[: nb * 1024].reshape(nb, 1024).min(1)
        q = np.quantile(sl, [0.0001, 0.001, 0.01, 0.1, 0.5])
        qb = np.quantile(bmin, [0.01, 0.1, 0.5, 0.9])
        m = eng.last_screen_mode()
        print(f"it {it:3d} rounds {eng.last_screen_rounds()} skipped16 {m[4]} | ub med {np.median(ub):.1f} lb med {np.median(lb):.1f} | "
              f"slack quantiles 1e-4..0.5: {np.round(q, 1).tolist()} | share of points with slack < 0 / 5 / 20: "
              f"{(sl < 0).mean():.5f} {(sl < 5).mean():.5f} {(sl < 20).mean():.5f} | block-min slack quantiles .01 .1 .5 .9: {np.round(qb, 1).tolist()} "
              f"blocks with min slack < 0 / 5: {(bmin < 0).mean():.4f} {(bmin < 5).mean():.4f}", flush=True)
