"""Driver-level timing: kmeans_sparsified() itself -- the entry point the reference's users call -- on an in-memory
dataset of N x 1024 float32 values (default N = 1e7: 41 GB of host memory), K = 100, 5 % sparsification, 'sample' start.
The dense mixture is generated on the GPU chunk by chunk and parked in host memory; everything from there on is the
driver: streamed ingest (PCIe -> widen -> mix -> sample -> resident shard), start, Lloyd loop (the fused call, one small
D2H per iteration), distances once at the end, unmix.   python tools/driver_bench.py [N] [K] [MaxIter]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd.kmeans import kmeans_sparsified

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
maxiter = int(sys.argv[3]) if len(sys.argv) > 3 else 100
p = 1024
g = torch.Generator(device="cuda")
g.manual_seed(234)
means = torch.randn((K, p), generator=g, device="cuda", dtype=torch.float32)
X = torch.empty((n, p), dtype=torch.float32, pin_memory=False)
t0 = time.time()
for c0 in range(0, n, 262144):
    m = min(262144, n - c0)
    lab = torch.randint(0, K, (m,), generator=g, device="cuda")                       # arbitrary point order
    X[c0:c0 + m].copy_(means[lab] + 0.1 * torch.randn((m, p), generator=g, device="cuda", dtype=torch.float32))
torch.cuda.synchronize()
t_gen = time.time() - t0
Xn = X.numpy()
kmeans_sparsified(Xn[:20000], K, Sparsify=True, SparsityLevel=0.05, rng=0, MaxIter=3)  # warm-up (library load, allocations)
res = {}
for start in ("sample", "Arthur"):
    t0 = time.time()
    IDX, C, SUMD, D, O = kmeans_sparsified(Xn, K, Sparsify=True, SparsityLevel=0.05, rng=1, MaxIter=maxiter, Start=start)
    dt = time.time() - t0
    its = int(O["iterations"][0])
    lloyd = float(O["TimeAlgo_wo_initialization"])
    res[start] = dict(total_s=dt, ingest_s=float(O["TimeToSketch"]), ingest_GBs=n * p * 4 / float(O["TimeToSketch"]) / 1e9,
                      init_s=float(O["TimeInitialization"]), lloyd_s=lloyd, iterations=its,
                      ms_per_iteration=1e3 * lloyd / its, fused_iterations=int(O["fusedIterations"][0]),
                      last_path=int(O["lastPath"][0]), objective=float(O["objectives"][0]),
                      stopping_diff=float(O["stoppingDiff"][0]), clusters_found=int(len(np.unique(IDX))))
    print(f"kmeans_sparsified N={n} d={p} K={K} Start={start}: total {dt:.2f} s = ingest {res[start]['ingest_s']:.2f} s "
          f"({res[start]['ingest_GBs']:.1f} GB/s of float32 from pageable host memory) + start {res[start]['init_s']:.2f} s + "
          f"{its} Lloyd iterations {lloyd:.3f} s ({res[start]['ms_per_iteration']:.2f} ms each, {res[start]['fused_iterations']} "
          f"through the fused call, last path {res[start]['last_path']})", flush=True)
print(json.dumps(dict(workload=f"kmeans_sparsified(X, {K}, Sparsify, 0.05) N={n} d={p} float32 in host memory, shuffled order",
                      datagen_s=t_gen, runs=res)))
