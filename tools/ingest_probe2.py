"""GPU box: H2D rate out of a pinned buffer depending on who wrote it last."""
import os, sys, time
import numpy as np, torch
p, m = 1024, 262144
src = torch.randn((m, p), dtype=torch.float32)
pin = torch.empty((m, p), dtype=torch.float32, pin_memory=True)
dev = torch.empty((m, p), dtype=torch.float32, device="cuda")
gsrc = torch.randn((m, p), dtype=torch.float32, device="cuda")
def h2d(tag, reps=3):
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
        print(f"  {tag} H2D #{r}: {m * p * 4 / (time.perf_counter() - t0) / 1e9:6.1f} GB/s")
print("affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
pin.copy_(gsrc); torch.cuda.synchronize(); h2d("after D2H fill")
pin.copy_(src); h2d("after CPU copy_ (torch, parallel)")
torch.set_num_threads(1); pin.copy_(src); h2d("after CPU copy_ (1 thread)")
torch.set_num_threads(16); t0 = time.perf_counter(); pin.copy_(src); print("  cpu copy 16 thr", m*p*4/(time.perf_counter()-t0)/1e9); h2d("after CPU copy_ (16 threads)")
# registered pageable memory instead of a staging copy
big = torch.randn((m, p), dtype=torch.float32)
t0 = time.perf_counter()
rc = torch.cuda.cudart().cudaHostRegister(big.data_ptr(), big.numel() * 4, 0)
print("hostRegister 1 GB:", rc, f"{time.perf_counter() - t0:.3f} s")
for r in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev.copy_(big, non_blocking=True); torch.cuda.synchronize()
    print(f"  registered pageable H2D #{r}: {m * p * 4 / (time.perf_counter() - t0) / 1e9:6.1f} GB/s")
t0 = time.perf_counter(); torch.cuda.cudart().cudaHostUnregister(big.data_ptr()); print(f"unregister {time.perf_counter() - t0:.3f} s")
