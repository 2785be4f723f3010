"""Dev aid (GPU box): the scenario of tests/test_gpu_screen.py::test_point_granular_bounds_list_on_data_in_arbitrary_order,
printing every call's form / counters; SPKM_AB_LIB=<path to another libspkm.so> runs it on that build (A/B across rounds)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_lib  # noqa: F401
from oracle import oracle as O
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context
O.build()
ctx = torch_context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4099
p, K, gopt = 256, (24 if n > 1000 else 3), 0.1
X, centres, labels = synth.gmm_dense(p, n, K, seed=31, noise=0.3)
X = X[:, np.random.default_rng(0).permutation(n)]
rng = np.random.default_rng(2)
d = np.sign(rng.standard_normal(p))
Y = synth.sparsify_dense(O.mix(X, d, p), synth.small_p_of(gopt, p), rng)
gam = synth.small_p_of(gopt, p) / p
shard = Shard.from_scipy(ctx, Y)
C0 = O.mix(X[:, rng.choice(n, K, replace=False)], d, p)
eng = LloydEngine(shard, K, gam)
c = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
for it in range(16):
    eng.iterate(c)
    torch.cuda.synchronize()
    print(it, "mode", eng.last_screen_mode(), "rounds", eng.last_screen_rounds(), "path", eng.last_path_info())
