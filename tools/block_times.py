#!/usr/bin/env python3
"""Developer aid: per-workgroup start/end times of the tiled assignment kernel at bench scale."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparsifiedkmeans_amd import _lib, synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device, torch_context
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = torch_context(0)
d = synth.sparsified_gmm_device(ctx, 1024, n, n, 0, K, 0.05, chunk=131072)
shard = Shard.from_device(ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
eng = LloydEngine(shard, K, d["gamma"])
centers = mix_device(ctx, (d["means"][:K] + 0.0).contiguous(), d["p2"], d["sign"], 1.0, 32.0)
L = _lib.lib()
eng.assign_step(centers); torch.cuda.synchronize()
_lib.check(L.spkm_debug_block_times(ctx.handle, 1, None, 0, None))
eng.assign_step(centers); torch.cuda.synchronize()
buf = (C.c_int64 * 1024)(); nb = C.c_int()
_lib.check(L.spkm_debug_block_times(ctx.handle, 0, buf, 512, C.byref(nb)))
t = np.array(buf[: 2 * nb.value]).reshape(-1, 2).astype(float)
t0 = t[:, 0][t[:, 0] > 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # microseconds
print("kernel ms", eng.last_assign_kernel_ms())
dur = en - st
act = dur > 1
print(f"blocks={nb.value} active={act.sum()} start us: min {st[act].min():.1f} max {st[act].max():.1f}; end us: min {en[act].min():.1f} p50 {np.median(en[act]):.1f} max {en[act].max():.1f}")
print("duration us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (dur[act].min(), *np.percentile(dur[act], [10, 50, 90]), dur[act].max()))
for x in range(8):
    m = act & (np.arange(nb.value) % 8 == x)
    print(f"  XCD {x}: n={m.sum()} end p50 {np.median(en[m]):.0f} max {en[m].max():.0f}")
late = np.argsort(-en)[:12]
print("latest blocks (id, xcd, local idx, end):", [(int(b), int(b % 8), int(b // 8), round(float(en[b]))) for b in late])
