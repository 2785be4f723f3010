"""GPU box: the headline workload in SHUFFLED point order at full size -- after the fused path has reached the point-list
regime (screen on listed points read from the record layout, results by list slot), its assignment and distances for the
last call are compared, for every one of the N points, with the all-exact f64 kernels (SPKM_NO_SCREEN=1) on the same
centres.      python tools/fullsize_shuffled_check.py [N] [iterations]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device, torch_context

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 16
K, p = 100, 1024
ctx = torch_context(0)
d = synth.sparsified_gmm_device(ctx, p, n, n, 0, K, 0.05, seed=234, order="shuffled")
p2, gamma = d["p2"], d["gamma"]
shard = Shard.from_device(ctx, p2, d["jc"], d["ir"], d["x"], nnz=d["nnz"])
g = torch.Generator(device="cuda"); g.manual_seed(234 + 17)
lab = torch.randint(0, K, (K,), generator=g, device="cuda")
start = d["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
c = mix_device(ctx, start.contiguous(), p2, d["sign"], 1.0, float(np.sqrt(np.float64(p2))))
eng = LloydEngine(shard, K, gamma)
modes = []
for it in range(iters):
    used = c.clone()
    eng.iterate(c)
    torch.cuda.synchronize()
    modes.append(eng.last_screen_mode()[7])
print("list modes per call (2 = point list):", modes)
a_fused, d_fused = eng.assign.clone(), eng.mind.clone()
os.environ["SPKM_NO_SCREEN"] = "1"
ctx.reload_switches()
ex = LloydEngine(shard, K, gamma)
ex.assign_accumulate_step(used)
torch.cuda.synchronize()
os.environ.pop("SPKM_NO_SCREEN")
ctx.reload_switches()
assert ex.last_path_info()[0] == 0
na = int((a_fused != ex.assign).sum().item()); nd = int((d_fused != ex.mind).sum().item())
print(f"N = {n}: assignments differing from the all-exact kernels: {na}; distances differing: {nd}")
sys.exit(1 if (na or nd or 2 not in modes) else 0)
