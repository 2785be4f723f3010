"""Times the first fused call of a lazy run (the one-pass form when K <= 16) on a synthetic shard: n K [repeats]."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import _lib, synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context, mix_device
ctx = torch_context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
p = 1024
d = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234, order="shuffled")
sh = Shard.from_device(ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
sh.set_lazy_stats(True)
g = torch.Generator(device="cuda"); g.manual_seed(251)
start = d["means"][torch.randint(0, K, (K,), generator=g, device="cuda")] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
c0 = mix_device(ctx, start.contiguous(), d["p2"], d["sign"], 1.0, float(np.sqrt(np.float64(d["p2"]))))
eng = LloydEngine(sh, K, d["gamma"])
L = _lib.lib()
for it in range(reps):
    sh.reset_policy()
    c = c0.clone()
    _lib.check(L.spkm_timing_log(ctx.handle, 2))
    eng.iterate(c, want_mind=False)
    torch.cuda.synchronize()
    buf = (C.c_double * 4)(); cnt = C.c_int()
    _lib.check(L.spkm_timing_read(ctx.handle, buf, 4, C.byref(cnt)))
    m = eng.last_screen_mode()
    print(f"call {it}: kernel {buf[0]:7.3f} ms  second slot {buf[1]:6.3f}  onepass {m[6]} listed {m[1]}  ({n * 510 / buf[0] / 1e6:.0f} GB/s of records)")
