"""GPU box: where does the driver's ingest of pageable host memory spend its time?  (host copy into pinned staging,
H2D, widen + mix + sample) -- each alone, on 1 GB float32 chunks."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd import engine

p, m = 1024, 262144
src = torch.randn((4 * m, p), dtype=torch.float32)            # pageable, 4 GB
pin = torch.empty((m, p), dtype=torch.float32, pin_memory=True)
dev = torch.empty((m, p), dtype=torch.float32, device="cuda")
for thr in (1, 4, 8, 16, 32, 64):
    
    t0 = time.perf_counter()
    for c in range(4):
        engine._parallel_host_copy(pin, src[c * m:(c + 1) * m], threads=thr)
    dt = time.perf_counter() - t0
    print(f"host copy pageable->pinned, {thr:2d} threads: {4 * m * p * 4 / dt / 1e9:6.1f} GB/s")
t0 = time.perf_counter()
for c in range(4):
    pin.copy_(src[c * m:(c + 1) * m])
print(f"torch copy_ alone: {4 * m * p * 4 / (time.perf_counter() - t0) / 1e9:6.1f} GB/s  (torch threads {torch.get_num_threads()})")
a = src.numpy(); b = pin.numpy()
t0 = time.perf_counter()
for c in range(4):
    np.copyto(b, a[c * m:(c + 1) * m])
print(f"numpy copyto alone: {4 * m * p * 4 / (time.perf_counter() - t0) / 1e9:6.1f} GB/s")
torch.cuda.synchronize()
t0 = time.perf_counter()
for c in range(4):
    dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D from pinned: {4 * m * p * 4 / (time.perf_counter() - t0) / 1e9:6.1f} GB/s")
t0 = time.perf_counter()
for c in range(4):
    dev.copy_(src[c * m:(c + 1) * m])
torch.cuda.synchronize()
print(f"H2D from pageable (torch): {4 * m * p * 4 / (time.perf_counter() - t0) / 1e9:6.1f} GB/s")
