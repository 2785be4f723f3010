"""Practical HBM read bandwidth on this box: torch reductions / copies over a 40 GB tensor (reference point for
the roofline fractions of the streaming kernels)."""
import torch, time
x = torch.empty(5_000_000_000, dtype=torch.float64, device="cuda").fill_(1.0)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
dt = t(lambda: x.sum()); print(f"sum f64 40 GB: {40/dt/1e3:.2f} TB/s")
xi = x.view(torch.int64)
dt = t(lambda: xi.max()); print(f"max i64 40 GB: {40/dt/1e3:.2f} TB/s")
y = torch.empty(2_500_000_000, dtype=torch.float64, device="cuda")
dt = t(lambda: y.copy_(x[:2_500_000_000])); print(f"copy 20 GB -> 20 GB: read+write {40/dt/1e3:.2f} TB/s")
