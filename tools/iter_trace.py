"""Per-iteration view of the bench run (sample start): screen / accumulation kernel times, form, skipped steps."""
import os, sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import _lib, synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context, mix_device
ctx = torch_context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
K, p = 100, 1024
d = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234)
sh = Shard.from_device(ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
g = torch.Generator(device="cuda"); g.manual_seed(234 + 17)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = d["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
c = mix_device(ctx, start.contiguous(), d["p2"], d["sign"], 1.0, float(np.sqrt(np.float64(d["p2"]))))
eng = LloydEngine(sh, K, d["gamma"])
L = _lib.lib()
for it in range(14):
    _lib.check(L.spkm_timing_log(ctx.handle, 2))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.iterate(c)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    buf = (C.c_double * 4)(); cnt = C.c_int()
    _lib.check(L.spkm_timing_read(ctx.handle, buf, 4, C.byref(cnt)))
    m = eng.last_screen_mode()
    print(f"iter {it}: {dt*1e3:6.2f} ms  screen {buf[0]:6.2f}  acc {buf[1]:6.2f}  form {m[0]} listed {m[1]} early {m[3]} skipped {m[4]}")
