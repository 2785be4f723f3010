"""-m gpu: per-point distances on demand (d_mind = NULL in the fused call + spkm_distances_dev) are the distances the
fused call would have stored, bit for bit -- and those are the oracle's."""
import numpy as np
import pytest
import torch

from util import parts, random_csc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p,n,K,s,ragged", [(256, 30000, 20, 13, False), (1024, 20000, 100, 51, False),
                                            (128, 9000, 7, 9, True), (512, 12000, 3, 64, False)])
def test_distances_on_demand(gpu_ctx, oracle, p, n, K, s, ragged):
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    X = random_csc(p, n, s, seed=p + K, ragged=ragged, empty_cols=(3,) if ragged else ())
    shard = Shard.from_scipy(gpu_ctx, X)
    C0 = np.random.default_rng(K).standard_normal((K, p))
    gam = s / p
    ref_eng = LloydEngine(shard, K, gam)
    eng = LloydEngine(shard, K, gam)
    c_ref, c = torch.tensor(C0, device="cuda"), torch.tensor(C0, device="cuda")
    for it in range(4):
        used = c.clone()
        ref_eng.assign_accumulate_step(c_ref)                       # stores the distances
        eng.mind.fill_(-1.0)
        eng.assign_accumulate_step(c, want_mind=False)              # does not
        assert np.array_equal(eng.assign.cpu().numpy(), ref_eng.assign.cpu().numpy())
        assert np.all(eng.mind.cpu().numpy() == -1.0)               # untouched
        st, st_ref = eng.stats.cpu().numpy(), ref_eng.stats.cpu().numpy()             # obj2, max distance, its index
        assert np.array_equal(st[1:], st_ref[1:]) and abs(st[0] - st_ref[0]) <= 1e-12 * st_ref[0]   # (obj2: order of summation)
        got = eng.distances(used).cpu().numpy()                     # right after the fused call: the kept-sort pass
        assert np.array_equal(got, ref_eng.mind.cpu().numpy())
        ra, rd = oracle.assign(p, n, *parts(X), used.cpu().numpy().T, gam)
        assert np.array_equal(got, rd) and np.array_equal(eng.assign.cpu().numpy(), ra)
        # an assignment the library's kept sort does not describe: generic kernel, distance to the GIVEN centroid
        a2 = torch.tensor(np.random.default_rng(it).integers(0, K, n).astype(np.int32), device="cuda")
        eng2 = LloydEngine(shard, K, gam)
        eng2.assign.copy_(a2)
        d2 = eng2.distances(used).cpu().numpy()
        full = oracle.dist_csc(p, n, *parts(X), used.cpu().numpy().T / gam)           # K x n
        assert np.array_equal(d2, full[a2.cpu().numpy(), np.arange(n)])
        for e, cc in ((ref_eng, c_ref), (eng, c)):
            e.allreduce_step()
            e.finalize_step(cc)
        c.copy_(c_ref)                                              # (sums are atomics: keep both loops on identical centres)
