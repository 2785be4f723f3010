"""GPU box experiment: does the (VALU/LDS-bound) screen tolerate an HBM-bound accumulation pass running beside it on
another stream?  Two contexts (two streams), two shards; A loops fused lazy calls whose work is the plain screen over all
points (SPKM_NO_BOUNDS, no movers), B loops spkm_accumulate_dev.  Alone, then together.
    python tools/exp_overlap.py [N per shard]"""
import os
import sys
import threading
import time

os.environ["SPKM_NO_BOUNDS"] = "1"
os.environ["SPKM_NO_PRUNE"] = "1"
os.environ["SPKM_NO_HINT"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsifiedkmeans_amd import synth                                   # noqa: E402
from sparsifiedkmeans_amd.engine import Context, LloydEngine, Shard, mix_device  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000
p, K, gam0 = 1024, 100, 0.05


def make(ctx, layout, seed):
    d = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, gam0, seed=seed, order="block", layout=layout)
    if layout == "records":
        sh = Shard.from_records(ctx, d["p2"], n, d["s"], d["rec"], d["ir_bits"])
    else:
        sh = Shard.from_device(ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
    g = torch.Generator(device="cuda"); g.manual_seed(seed + 17)
    lab = torch.randint(0, K, (K,), generator=g, device="cuda")
    start = d["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
    c = mix_device(ctx, start.contiguous(), d["p2"], d["sign"], 1.0, float(np.sqrt(np.float64(d["p2"]))))
    return d, sh, c


ctxA, ctxB = Context(), Context()
dA, shA, cA = make(ctxA, "records", 234)
dB, shB, cB = make(ctxB, "csc", 235)
shA.set_lazy_stats(True)
eA = LloydEngine(shA, K, dA["gamma"])
eB = LloydEngine(shB, K, dB["gamma"])
eB.assign_accumulate_step(cB)                 # an assignment for B's accumulation
for _ in range(3):
    eA.assign_accumulate_step(cA, want_mind=False)
torch.cuda.synchronize()


def loop(fn, secs, out):
    t_end = time.perf_counter() + secs
    k = 0
    t0 = time.perf_counter()
    while time.perf_counter() < t_end:
        for _ in range(4):
            fn()
        k += 4
        torch.cuda.synchronize()   # (both streams: a coarse fence every 4 calls keeps the queues short)
    out.append((time.perf_counter() - t0) / k * 1e3)


fa = lambda: eA.assign_accumulate_step(cA, want_mind=False)
fb = lambda: eB.accumulate_step()
ra, rb = [], []
loop(fa, 1.0, ra); loop(fb, 1.0, rb)
print(f"alone:    screen call {ra[0]:.2f} ms   accumulate call {rb[0]:.2f} ms   (N = {n} each)   form {eA.last_screen_mode()}")
ra2, rb2 = [], []
ta = threading.Thread(target=loop, args=(fa, 2.0, ra2)); tb = threading.Thread(target=loop, args=(fb, 2.0, rb2))
ta.start(); tb.start(); ta.join(); tb.join()
print(f"together: screen call {ra2[0]:.2f} ms   accumulate call {rb2[0]:.2f} ms")
print(f"serial cost of one of each {ra[0] + rb[0]:.2f} ms; together, per pair, about {max(ra2[0], rb2[0]):.2f} ms if the rates matched")
