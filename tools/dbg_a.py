"""Timing aid: the two-phase screen (SPKM_SCREEN_A rounds for all centroids, the rest only for each tile's leader)
against the full screen, from a converged start (distinct centres = the planted means)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import _lib, synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context, mix_device
import ctypes as C
ctx = torch_context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
d = synth.sparsified_gmm_device(ctx, 1024, n, n, 0, 100, 0.05, seed=234)
sh = Shard.from_device(ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
g = torch.Generator(device="cuda"); g.manual_seed(251)
start = d["means"] + 0.01 * torch.randn((100, 1024), generator=g, device="cuda", dtype=torch.float64)
c0 = mix_device(ctx, start.contiguous(), d["p2"], d["sign"], 1.0, 32.0)
L = _lib.lib()
for A in (13, 6, 4, 3, 2):
    os.environ["SPKM_SCREEN_A"] = str(A)
    eng = LloydEngine(sh, 100, d["gamma"])
    c = c0.clone()
    for _ in range(3): eng.iterate(c)
    _lib.check(L.spkm_timing_log(ctx.handle, 1))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(6): eng.iterate(c)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 6
    buf = (C.c_double * 6)(); cnt = C.c_int()
    _lib.check(L.spkm_timing_read(ctx.handle, buf, 6, C.byref(cnt))); _lib.check(L.spkm_timing_log(ctx.handle, 0))
    print(f"A={A}: {dt*1e3:.2f} ms/iter, screen kernel {np.mean(buf[:cnt.value]):.2f} ms, path/listed {eng.last_path_info()}")
