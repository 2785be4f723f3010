"""GPU box: randomized parity stress of the fused iteration (screen + carried bounds + point lists + cluster shortcut +
in-place assignment copy) against the CPU oracle: random shapes, block-ordered and shuffled data, 'sample' starts, every
iteration's assignment compared bit for bit, distances on demand and centroids at the end.
    python tools/stress_parity.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from sparsifiedkmeans_amd import synth
from sparsifiedkmeans_amd.engine import LloydEngine, Shard, torch_context
from util import parts

O.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = torch_context(0)
t_end = time.time() + budget
cases = fails = 0
while time.time() < t_end:
    p = int(rng.choice([64, 128, 256, 512, 1024]))
    gopt = float(rng.choice([0.05, 0.1, 0.2]))
    s = synth.small_p_of(gopt, p)
    n = int(rng.integers(50, 40000))
    K = int(rng.integers(2, min(130, n // 4)))
    if rng.random() < 0.25:
        K = int(rng.integers(2, min(17, n // 4)))             # few centroids: single-tile screen
    noise = float(rng.choice([0.1, 0.3, 0.6]))
    shuffled = bool(rng.integers(0, 2))
    iters = int(rng.integers(6, 26))
    # the library's A/B switches (none may change an output): each on in about one case of six
    switches = ["SPKM_NO_REC", "SPKM_NO_CLUSTER_SKIP", "SPKM_NO_POINT_LIST", "SPKM_NO_LATE_SPLIT", "SPKM_NO_INCREMENTAL",
                "SPKM_NO_HINT", "SPKM_NO_PRUNE", "SPKM_NO_BOUNDS", "SPKM_CHECK_ASSIGN", "SPKM_NO_REGROUP",
                "SPKM_NO_SUPPORT_DRIFT", "SPKM_NO_TEAMS", "SPKM_NO_DUAL", "SPKM_NO_SUMS_ONLY", "SPKM_NO_BLOCK_SKIP",
                "SPKM_NO_DIRECT_EVENTS", "SPKM_NO_PAIR_EVENTS", "SPKM_FORCE_PAIR_EVENTS", "SPKM_FORCE_PAIR_EVENTS"]
    on = [w for w in switches if rng.random() < 1 / 6]
    for w in switches:
        os.environ.pop(w, None)
    for w in on:
        os.environ[w] = "1"
    ctx.reload_switches()                                    # (the library reads its switches once per context)
    X, centres, labels = synth.gmm_dense(p, n, K, seed=int(rng.integers(1 << 30)), noise=noise)
    if shuffled:
        X = X[:, rng.permutation(n)]
    d = np.sign(rng.standard_normal(p)); d[d == 0] = 1
    Y = synth.sparsify_dense(O.mix(X, d, p), s, rng)
    gam = s / p
    Yones = Y.copy(); Yones.data[:] = 1.0                     # spones(X) (kmeans_sparsified.m:352-355)
    shard = Shard.from_scipy(ctx, Y)
    C0 = O.mix(X[:, rng.choice(n, K, replace=True)], d, p) + 1e-3 * rng.standard_normal((p, K))
    eng = LloydEngine(shard, K, gam)
    c = torch.tensor(np.ascontiguousarray(C0.T), device="cuda")
    want_mind = bool(rng.integers(0, 2))
    lazy = bool(rng.integers(0, 2))                          # lazy statistics: incremental sums once few points move
    shard.set_lazy_stats(lazy)
    on = on + (["lazy"] if lazy else [])
    ok = True
    for it in range(iters):
        used = c.cpu().numpy().T.copy()
        eng.iterate(c, want_mind=want_mind)
        torch.cuda.synchronize()
        ra, rd = O.assign(p, n, *parts(Y), used, gam)
        if not np.array_equal(eng.assign.cpu().numpy(), ra):
            ok = False; print("ASSIGN MISMATCH", dict(p=p, n=n, K=K, s=s, noise=noise, shuffled=shuffled, it=it, on=on)); break
        if want_mind and not np.array_equal(eng.mind.cpu().numpy(), rd):
            ok = False; print("MIND MISMATCH", dict(p=p, n=n, K=K, s=s, it=it, on=on)); break
        # centres after the update vs the oracle's update from the same assignment
        # (kmeans_sparsified.m:447-448 in numpy for the clusters that have members; an empty cluster keeps its column)
        got = c.cpu().numpy().T
        ind = sp.csr_matrix((np.ones(n), (ra, np.arange(n))), shape=(K, n))
        S = (Y @ ind.T).toarray()
        Cnt = (Yones @ ind.T).toarray()
        refc = np.where(np.bincount(ra, minlength=K)[None, :] > 0, gam * S / (Cnt + 1e-16), used)
        scale = max(1e-300, np.abs(refc).max())
        # objective of the call: the oracle's, or NaN when a lazy call left it out
        o2 = float(eng.out[1].item())
        if not (np.isnan(o2) and lazy and not want_mind) and abs(o2 - np.sum(rd * rd)) > 1e-9 * np.sum(rd * rd):
            ok = False; print("OBJECTIVE MISMATCH", dict(p=p, n=n, K=K, s=s, it=it, on=on, got=o2, want=float(np.sum(rd * rd)))); break
        if np.abs(got - refc).max() > 1e-9 * scale:
            ok = False; print("CENTRE MISMATCH", dict(p=p, n=n, K=K, s=s, it=it, on=on, err=np.abs(got - refc).max() / scale)); break
    if ok and not want_mind:
        eng.distances(torch.tensor(np.ascontiguousarray(used.T), device="cuda"))
        if not np.array_equal(eng.mind.cpu().numpy(), rd):
            ok = False; print("DISTANCES-ON-DEMAND MISMATCH", dict(p=p, n=n, K=K, s=s, on=on))
        st = eng.stats.cpu().numpy()
        if abs(st[0] - np.sum(rd * rd)) > 1e-9 * np.sum(rd * rd) or st[1] != rd.max() or int(st[2]) != int(np.argmax(rd)):
            ok = False; print("STATS-ON-DEMAND MISMATCH", dict(p=p, n=n, K=K, s=s, on=on, st=st.tolist()))
    cases += 1; fails += (not ok)
    del shard, eng
print(f"{cases} random cases, {fails} failures")
sys.exit(1 if fails else 0)
