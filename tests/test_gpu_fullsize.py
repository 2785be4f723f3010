"""-m gpu: the benchmark workload AT FULL SIZE (BASELINE.json config: N = 1e8, d = 1024, K = 100, s = 51), checked
through size-independent properties -- no CPU oracle can visit 5.1e9 stored entries in a test:

 * the certified-screen path and the all-exact f64 kernels (SPKM_NO_SCREEN=1; themselves bit-exact against the
   oracle at small sizes, tests/test_gpu_assign.py) give the SAME assignment and the SAME min-distance for every
   one of the 1e8 points, bit for bit, and the same counts; sums agree to summation-order noise;
 * counts add up to N; the objective equals sum(mind^2);
 * idempotence: after convergence one more iteration leaves every assignment and the centres' fixed point
   unchanged.

Needs ~110 GB of HBM; set SPKM_FULLSIZE_N to run a smaller instance."""
import os

import numpy as np
import pytest
import torch

from util import set_switch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workload(gpu_ctx):
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import Shard

    n = int(float(os.environ.get("SPKM_FULLSIZE_N", "1e8")))
    free, total = torch.cuda.mem_get_info()
    need = n * (51 * 10 + 51 * 6 + 16 + 8 * 12 + 24)           # shard + screen copy + norms + exact partials + outputs
    if free < need * 1.1:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of free HBM, {free / 2**30:.0f} available")
    p, K, gam = 1024, 100, 0.05
    d = synth.sparsified_gmm_device(gpu_ctx, p, n, n, 0, K, gam, seed=234)
    shard = Shard.from_device(gpu_ctx, d["p2"], d["jc"], d["ir"], d["x"], nnz=d["nnz"])
    g = torch.Generator(device="cuda")
    g.manual_seed(235)
    c0 = d["means"] + 0.5 * torch.randn(d["means"].shape, generator=g, device="cuda", dtype=torch.float64)
    sq = float(np.sqrt(np.float64(d["p2"])))
    centers0 = torch.zeros((K, d["p2"]), dtype=torch.float64, device="cuda")
    centers0[:, :p] = c0
    from sparsifiedkmeans_amd.engine import mix_device
    centers0 = mix_device(gpu_ctx, centers0, d["p2"], d["sign"], 1.0, sq)
    return dict(n=n, K=K, p2=d["p2"], gamma=d["gamma"], shard=shard, centers0=centers0, keep=d, ctx=gpu_ctx)


def _one_pass(w, centers, no_screen):
    from sparsifiedkmeans_amd.engine import LloydEngine
    if no_screen:
        set_switch(None, w["ctx"], "SPKM_NO_SCREEN")
    try:
        eng = LloydEngine(w["shard"], w["K"], w["gamma"])
        eng.assign_accumulate_step(centers)
        torch.cuda.synchronize()
        path, listed = eng.last_path_info()
    finally:
        set_switch(None, w["ctx"], "SPKM_NO_SCREEN", False)
    return eng, path, listed


def test_screen_and_exact_kernels_agree_on_every_point(workload):
    w = workload
    n, K, p2 = w["n"], w["K"], w["p2"]
    es, path_s, listed = _one_pass(w, w["centers0"], False)
    a_s, d_s, r_s = es.assign.clone(), es.mind.clone(), es.reduce.clone()
    del es
    ee, path_e, _ = _one_pass(w, w["centers0"], True)
    assert path_s == 1 and path_e == 0
    assert torch.equal(a_s, ee.assign)                          # 1e8 assignments, bit for bit
    assert torch.equal(d_s, ee.mind)                            # and the min-distances (f64 bit patterns)
    pk = p2 * K
    assert torch.equal(r_s[pk:2 * pk + K], ee.reduce[pk:2 * pk + K])        # per-row counts and cluster sizes
    assert float(r_s[2 * pk:2 * pk + K].sum().item()) == float(n)
    scale = float(ee.reduce[:pk].abs().max().item())
    assert float((r_s[:pk] - ee.reduce[:pk]).abs().max().item()) <= 1e-11 * scale     # sums: atomics order only
    obj = float((d_s * d_s).sum().item())
    assert abs(float(r_s[-1].item()) - obj) <= 1e-10 * obj
    assert listed < 0.001 * n                                   # almost every point is certified by the screen


def test_fixed_point_is_idempotent(workload):
    from sparsifiedkmeans_amd.engine import LloydEngine
    w = workload
    eng = LloydEngine(w["shard"], w["K"], w["gamma"])
    c = w["centers0"].clone()
    prev = None
    for it in range(40):
        eng.iterate(c)
        a = eng.assign.clone()
        if prev is not None and torch.equal(a, prev):
            break
        prev = a
    else:
        pytest.fail("no assignment fixed point in 40 iterations")
    c_fix = c.clone()
    eng.iterate(c)
    assert torch.equal(eng.assign, prev)
    assert torch.allclose(c, c_fix, rtol=1e-12, atol=1e-14)     # same points per cluster -> same means (atomics order)


def test_lloyd_run_with_every_layer_equals_exact_kernels_each_iteration(workload):
    """The benchmark's regime at full size: a Lloyd run from sampled (duplicate) centres.  The default path goes
    through the plain screen, the hinted two-phase screen and, once the centres settle, the carried bounds that skip
    whole steps; after EVERY iteration its assignments and min-distances equal, bit for bit, what the all-exact f64
    kernels give for the same centres (a second shard object over the same device buffers, so that the exact calls
    do not disturb the state the layers live on)."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device
    w = workload
    d, K, p2 = w["keep"], w["K"], w["p2"]
    g = torch.Generator(device="cuda")
    g.manual_seed(234 + 17)
    lab = torch.randint(0, K, (K,), generator=g, device="cuda")
    start = d["means"][lab] + 0.1 * torch.randn((K, 1024), generator=g, device="cuda", dtype=torch.float64)
    c = mix_device(w["shard"].ctx, start.contiguous(), p2, d["sign"], 1.0, float(np.sqrt(np.float64(p2))))
    twin = Shard.from_device(w["shard"].ctx, p2, d["jc"], d["ir"], d["x"], nnz=d["nnz"])
    w["shard"].reset_policy()                                   # the other tests of this module used the shard
    eng = LloydEngine(w["shard"], K, w["gamma"])
    exact = LloydEngine(twin, K, w["gamma"])
    forms, skipped = [], []
    for it in range(14):
        c_in = c.clone()
        eng.iterate(c)                                          # updates c in place
        torch.cuda.synchronize()
        m = eng.last_screen_mode()
        forms.append(m[0])
        skipped.append(m[4])
        set_switch(None, w["ctx"], "SPKM_NO_SCREEN")
        try:
            exact.assign_accumulate_step(c_in)
            torch.cuda.synchronize()
        finally:
            set_switch(None, w["ctx"], "SPKM_NO_SCREEN", False)
        assert exact.last_path_info()[0] == 0
        assert torch.equal(eng.assign, exact.assign), f"iteration {it}"
        assert torch.equal(eng.mind, exact.mind), f"iteration {it}"
    assert forms[0] == 0 and 2 in forms, forms                  # plain first, hinted later
    assert skipped[-1] > 0.5 * (w["n"] // 16), skipped          # the run has settled: most steps are skipped
