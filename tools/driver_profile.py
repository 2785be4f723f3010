"""GPU box: cProfile of one kmeans_sparsified() call on N x 1024 float32 host data (where do the seconds outside ingest and
the Lloyd loop go?)"""
import cProfile, pstats, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsifiedkmeans_amd.kmeans import kmeans_sparsified
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
K, p = 100, 1024
g = torch.Generator(device="cuda"); g.manual_seed(234)
means = torch.randn((K, p), generator=g, device="cuda", dtype=torch.float32)
X = torch.empty((n, p), dtype=torch.float32)
for c0 in range(0, n, 262144):
    m = min(262144, n - c0)
    lab = torch.randint(0, K, (m,), generator=g, device="cuda")
    X[c0:c0 + m].copy_(means[lab] + 0.1 * torch.randn((m, p), generator=g, device="cuda", dtype=torch.float32))
Xn = X.numpy()
kmeans_sparsified(Xn[:20000], K, Sparsify=True, SparsityLevel=0.05, rng=0, MaxIter=3)
if os.environ.get("DRIVER_CPROFILE"):
    pr = cProfile.Profile(); t0 = time.time(); pr.enable()
    kmeans_sparsified(Xn, K, Sparsify=True, SparsityLevel=0.05, rng=1, MaxIter=20, Start="sample")
    pr.disable(); print("total", time.time() - t0)
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# per-line wall time inside kmeans_sparsified's own frame (operators on arrays are not function calls for cProfile)
code = kmeans_sparsified.__code__
acc, last = {}, [None, 0.0]
def tracer(frame, event, arg):
    if frame.f_code is not code:
        return None
    def local(frame, event, arg):
        now = time.perf_counter()
        if last[0] is not None:
            acc[last[0]] = acc.get(last[0], 0.0) + now - last[1]
        last[0], last[1] = (frame.f_lineno if event != "return" else None), now
        return local
    return local
sys.settrace(tracer); t0 = time.time()
kmeans_sparsified(Xn, K, Sparsify=True, SparsityLevel=0.05, rng=1, MaxIter=20, Start="sample")
sys.settrace(None); print("total (traced)", time.time() - t0)
import linecache
for ln, t in sorted(acc.items(), key=lambda kv: -kv[1])[:14]:
    print(f"{t:8.3f} s  line {ln}: {linecache.getline(code.co_filename, ln).strip()[:110]}")
