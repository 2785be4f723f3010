"""-m gpu: the CPU ORACLE at the benchmark's own size (BASELINE.json: N = 1e8, d = 1024, K = 100, s = 51), in the regime
the driver's bench line runs -- lazy statistics on (spkm_shard_set_lazy_stats: sums-only first pass, then incremental
sums by events, no distances), the CSC arrays released (the kernels read the record layout and the f32 screen copy) --
in cluster-contiguous AND shuffled point order.

Assignment is independent per point (private/SparseMatrixMinusCluster.c:169-182 + findClusterAssignments.m:169), so a
random sample of the 1e8 points is a fair witness: after EVERY one of the run's iterations the columns of 1e5 sampled
points -- fetched back from the library's own layout with spkm_shard_get_column_host -- go through orc_assign under the
centres the call was given, and the fused path's assignment of those points must be the oracle's, bit for bit.  Per-cluster
sums and counts (kmeans_sparsified.m:430-431,447-448) are checked for three clusters per iteration by gathering ALL their
members (about 1e6 points each) and running orc_accumulate over them: counts exact, sums to 1e-9 relative (north-star bar
1e-6; only the order of summation differs).  At the end the run's distances on demand equal the oracle's on the sample.

tests/test_gpu_fullsize.py compares HIP with HIP on all 1e8 points; this file compares HIP with the oracle on a sample,
at the same size.  Set SPKM_FULLSIZE_N to run a smaller instance."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SAMPLE = 100_000
ITERS = 14


def _members_csc(x, ir, s, idx):
    """CSC pieces (jc, ir, x as the oracle wants them) of the columns ``idx`` (device int64) of a fixed-stride dataset"""
    ent = (idx[:, None] * s + torch.arange(s, device=idx.device)[None, :]).reshape(-1)
    xv = x[ent].cpu().numpy()
    rv = (ir[ent].to(torch.int32) & 0xFFFF).cpu().numpy().astype(np.uint64)
    jc = np.arange(0, (idx.numel() + 1) * s, s, dtype=np.uint64)
    return jc, rv, xv


@pytest.mark.parametrize("order", ["block", "shuffled"])
def test_lazy_run_equals_the_oracle_on_a_sample_at_full_size(gpu_ctx, oracle, order):
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device

    n = int(float(os.environ.get("SPKM_FULLSIZE_N", "1e8")))
    p, K, gam0, s = 1024, 100, 0.05, 51
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    need = n * (s * 10 * 2 + s * 6 + 16 + 8 * 12 + 24 + 24)     # dataset (kept here for the gathers) + records + screen copy + state
    if free < need * 1.1:
        pytest.skip(f"needs {need / 2**30:.0f} GiB of free HBM, {free / 2**30:.0f} available")
    d = synth.sparsified_gmm_device(gpu_ctx, p, n, n, 0, K, gam0, seed=234, order=order)
    p2, gamma = d["p2"], d["gamma"]
    assert d["nnz"] == n * s
    x_t, ir_t = d["x"], d["ir"]                                 # this test's own handles on the dataset (member gathers)
    shard = Shard.from_device(gpu_ctx, p2, d["jc"], d["ir"], d["x"], nnz=d["nnz"])
    shard.reset_policy()
    shard.set_lazy_stats(True)
    g = torch.Generator(device="cuda")
    g.manual_seed(234 + 17)
    lab = torch.randint(0, K, (K,), generator=g, device="cuda")  # sampled (duplicate) start centres: the bench's start
    start = d["means"][lab] + 0.1 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
    c = mix_device(gpu_ctx, start.contiguous(), p2, d["sign"], 1.0, float(np.sqrt(np.float64(p2))))
    eng = LloydEngine(shard, K, gamma)

    rng = np.random.default_rng(20260929 + (order == "shuffled"))
    samp = np.sort(rng.choice(n, min(SAMPLE, n), replace=False))
    samp_t = torch.tensor(samp, device="cuda")
    sjc = sir = sx = None
    forms, released = [], False
    pk = p2 * K
    for it in range(ITERS):
        c_in = c.clone()
        eng.iterate(c, want_mind=False)                         # updates c in place
        torch.cuda.synchronize()
        forms.append(eng.last_screen_mode()[6])
        assert eng.last_path_info()[0] == 1
        if it == 0:
            released = shard.release_csc()                      # from here on the records are the library's only copy
            assert released
            cols = [shard.column(int(i)) for i in samp]         # ... and the sample is read back from THEM
            assert all(len(r) == s for r, _ in cols)
            sir = np.concatenate([r for r, _ in cols]).astype(np.uint64)
            sx = np.concatenate([v for _, v in cols])
            sjc = np.arange(0, (len(samp) + 1) * s, s, dtype=np.uint64)
            # the records hold the dataset's entries
            j2, r2, x2 = _members_csc(x_t, ir_t, s, samp_t[:1000])
            assert np.array_equal(r2, sir[: 1000 * s]) and np.array_equal(x2, sx[: 1000 * s])
        Cm = c_in.cpu().numpy().T.copy()                        # p2 x K, the centres the call was given
        ra, rd = oracle.assign(p2, len(samp), sjc, sir, sx, Cm, gamma)
        a_dev = eng.assign
        got = a_dev[samp_t].cpu().numpy()
        assert np.array_equal(got, ra), f"{order}, iteration {it}: {np.count_nonzero(got != ra)} of {len(samp)} sampled points differ"
        # sums / counts of three clusters over ALL their members
        red = eng.reduce
        nk_dev = eng.nk.cpu().numpy()
        assert int(nk_dev.sum()) == n
        for k in {it % K, (7 * it + 31) % K, K - 1}:
            idx = torch.nonzero(a_dev == k).reshape(-1)
            assert idx.numel() == nk_dev[k]
            if idx.numel() == 0:
                continue
            mjc, mir, mx = _members_csc(x_t, ir_t, s, idx)
            S, Cnt, nk1 = oracle.accumulate(p2, idx.numel(), 1, mjc, mir, mx, np.zeros(idx.numel(), np.int32))
            got_s = red[k * p2:(k + 1) * p2].cpu().numpy()
            got_c = red[pk + k * p2: pk + (k + 1) * p2].cpu().numpy()
            assert np.array_equal(got_c, Cnt[:, 0]), f"{order}, iteration {it}, cluster {k}: counts"
            assert np.abs(got_s - S[:, 0]).max() <= 1e-9 * np.abs(S[:, 0]).max(), f"{order}, iteration {it}, cluster {k}: sums"
            del idx
    # distances on demand for the last call, on the sample
    eng.distances(c_in)
    assert np.array_equal(eng.mind[samp_t].cpu().numpy(), rd)
    shard.set_lazy_stats(False)
    assert forms[0] == 3 and 2 in forms, forms                  # sums-only first pass, then incremental calls
    del eng, shard, d, x_t, ir_t
    torch.cuda.empty_cache()
