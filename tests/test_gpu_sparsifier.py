"""-m gpu: the device sparsifier (mix -> sample -> CSC), SURVEY section 8(f) #1."""
import numpy as np
import pytest
import torch

from util import sample_rows_reference

pytestmark = pytest.mark.gpu


def _run(ctx, X, d, p2, s, seed, col0):
    from sparsifiedkmeans_amd.engine import mix_sample_device

    n = X.shape[1]
    ir = torch.zeros(n * s + 16, dtype=torch.int16, device="cuda:0")
    xv = torch.zeros(n * s + 16, dtype=torch.float64, device="cuda:0")
    mix_sample_device(ctx, torch.tensor(np.ascontiguousarray(X.T), device="cuda:0"), p2,
                      torch.tensor(d, device="cuda:0"), 1.0 + 2 * np.finfo(float).eps, float(np.sqrt(np.float64(p2))),
                      s, seed, col0, ir, xv)
    torch.cuda.synchronize()
    rows = ir[: n * s].cpu().numpy().view(np.uint16).astype(np.int64).reshape(n, s)
    return rows, xv[: n * s].cpu().numpy().reshape(n, s)


@pytest.mark.parametrize("p,s", [(784, 51), (1024, 51), (512, 26), (100, 13), (64, 64), (2048, 1)])
def test_mix_sample_matches_reference_pipeline(gpu_ctx, oracle, p, s):
    p2 = 1 << int(np.ceil(np.log2(p)))
    n, seed, col0 = 777, 0x1234_5678_9ABC, 10_000_000_000
    rng = np.random.default_rng(p)
    X = rng.standard_normal((p, n))
    d = np.sign(rng.standard_normal(p2))
    rows, vals = _run(gpu_ctx, X, d, p2, s, seed, col0)
    # sampler: exactly s distinct ascending rows per column, identical to the numpy restatement of the generator
    assert np.all(np.diff(rows, axis=1) > 0) and rows.min() >= 0 and rows.max() < p2
    assert np.array_equal(rows, sample_rows_reference(seed, col0, n, p2, s))
    # values: mix(X)(rows) / (s/p2), the reference's two divisions, bit for bit against the oracle's mix
    Xm = oracle.mix(X, d, p2)
    want = Xm[rows, np.arange(n)[:, None]] / (np.float64(s) / np.float64(p2))
    assert np.array_equal(vals, want)


def test_sample_is_independent_of_chunking_and_uniform(gpu_ctx, oracle):
    p, p2, s, seed = 256, 256, 16, 99
    X = np.random.default_rng(0).standard_normal((p, 6000))
    d = np.ones(p2)
    whole, _ = _run(gpu_ctx, X, d, p2, s, seed, 0)
    part, _ = _run(gpu_ctx, X[:, 2500:4000], d, p2, s, seed, 2500)
    assert np.array_equal(part, whole[2500:4000])          # a column's sample depends on (seed, global index) only
    other, _ = _run(gpu_ctx, X, d, p2, s, seed + 1, 0)
    assert not np.array_equal(other, whole)
    cnt = np.bincount(whole.ravel(), minlength=p2)           # every row equally likely: 6000*16/256 = 375 expected
    assert abs(cnt.mean() - 375) < 1e-9 and cnt.min() > 290 and cnt.max() < 460
    # pairs of adjacent rows are not correlated beyond chance (selection sampling is exact)
    both = np.mean([(np.isin(7, r) and np.isin(8, r)) for r in whole])
    assert abs(both - (16 / 256) * (15 / 255)) < 0.004


def test_device_pipeline_feeds_the_engine(gpu_ctx, oracle):
    """End to end on device: dense chunk -> mix+sample -> adopted shard -> assignment equals the oracle on the
    same CSC matrix."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, p2, s, n, K = 512, 512, 26, 4000, 7
    rng = np.random.default_rng(3)
    X = rng.standard_normal((p, n))
    d = np.sign(rng.standard_normal(p2))
    rows, vals = _run(gpu_ctx, X, d, p2, s, 5, 0)
    ir = torch.tensor(rows.astype(np.uint16).view(np.int16).ravel(), device="cuda:0")
    ir = torch.cat([ir, torch.zeros(16, dtype=torch.int16, device="cuda:0")])
    xv = torch.cat([torch.tensor(vals.ravel(), device="cuda:0"), torch.zeros(16, dtype=torch.float64, device="cuda:0")])
    jc = torch.arange(0, (n + 1) * s, s, dtype=torch.int64, device="cuda:0")
    eng = LloydEngine(Shard.from_device(gpu_ctx, p2, jc, ir, xv, nnz=n * s), K, s / p)
    Cm = rng.standard_normal((p2, K))
    eng.assign_step(torch.tensor(np.ascontiguousarray(Cm.T), device="cuda:0"))
    ra, rd = oracle.assign(p2, n, jc.cpu().numpy().astype(np.uint64), rows.ravel().astype(np.uint64), vals.ravel(), Cm, s / p)
    assert np.array_equal(eng.assign.cpu().numpy(), ra) and np.array_equal(eng.mind.cpu().numpy(), rd)


@pytest.mark.parametrize("p,s,n", [(1024, 51, 3001), (784, 51, 1000), (512, 26, 4097), (64, 5, 333), (256, 64, 700)])
def test_records_written_by_the_sparsifier_feed_the_engine_like_csc(gpu_ctx, oracle, p, s, n):
    """spkm_mix_sample_rec_dev + spkm_shard_create_rec_dev: the sparsifier writes the library's record layout directly (a
    point's s values, then its s row ids, in spkm_record_bytes(s) bytes) and the shard adopts it -- the separate CSC arrays
    never exist.  The records hold bit for bit what the CSC form of the same call holds; a Lloyd run on them (fused calls:
    screen copy built FROM the records; the exact kernels: CSC arrays re-materialised from them; columns read back;
    distances on demand) gives the oracle's assignments and distances."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_sample_records_device, record_bytes

    p2 = 1 << int(np.ceil(np.log2(p)))
    K, seed, col0 = 9, 77, 123456
    rng = np.random.default_rng(p + s)
    X = rng.standard_normal((p, n))
    d = np.sign(rng.standard_normal(p2)); d[d == 0] = 1
    rows, vals = _run(gpu_ctx, X, d, p2, s, seed, col0)                     # the CSC form
    R = record_bytes(s, 16)
    assert R % 16 == 0 and R >= s * 10 and R < s * 10 + 16
    rec = torch.zeros(n * R + 256, dtype=torch.uint8, device="cuda:0")
    mix_sample_records_device(gpu_ctx, torch.tensor(np.ascontiguousarray(X.T), device="cuda:0"), p2, torch.tensor(d, device="cuda:0"),
                              1.0 + 2 * np.finfo(float).eps, float(np.sqrt(np.float64(p2))), s, seed, col0, rec)
    torch.cuda.synchronize()
    r = rec[: n * R].cpu().numpy().reshape(n, R)
    assert np.array_equal(np.ascontiguousarray(r[:, : s * 8]).view(np.float64), vals)
    assert np.array_equal(np.ascontiguousarray(r[:, s * 8: s * 10]).view(np.uint16).astype(np.int64), rows)
    shard = Shard.from_records(gpu_ctx, p2, n, s, rec)
    assert (shard.p, shard.n, shard.nnz) == (p2, n, n * s)
    ir0, x0 = shard.column(n - 1)
    assert np.array_equal(ir0, rows[n - 1]) and np.array_equal(x0, vals[n - 1])
    gam = s / p
    jc = np.arange(0, (n + 1) * s, s, dtype=np.uint64)
    irf, xf = rows.ravel().astype(np.uint64), vals.ravel()
    eng = LloydEngine(shard, K, gam)
    c = torch.tensor(np.ascontiguousarray((rng.standard_normal((p2, K)) * 0.3).T), device="cuda:0")
    for it in range(4):
        used = c.cpu().numpy().T.copy()
        eng.iterate(c)                                                     # fused call (the screen where the shape qualifies)
        ra, rd = oracle.assign(p2, n, jc, irf, xf, used, gam)
        assert np.array_equal(eng.assign.cpu().numpy(), ra) and np.array_equal(eng.mind.cpu().numpy(), rd), it
    S, Cnt, nk = oracle.accumulate(p2, n, K, jc, irf, xf, ra)
    want = oracle.finalize_centers(S, Cnt, nk, gam, used)
    assert np.abs(c.cpu().numpy().T - want).max() <= 1e-9 * np.abs(want).max()
    used = c.cpu().numpy().T.copy()
    eng.assign_step(c)                                                     # the all-exact kernels: CSC arrays come back from the records
    ra, rd = oracle.assign(p2, n, jc, irf, xf, used, gam)
    assert np.array_equal(eng.assign.cpu().numpy(), ra) and np.array_equal(eng.mind.cpu().numpy(), rd)
    eng.distances(c)
    assert np.array_equal(eng.mind.cpu().numpy(), rd)
    assert shard.release_csc()                                             # ... and go again
    eng.iterate(c)
    assert np.array_equal(eng.assign.cpu().numpy(), ra)


def test_record_shard_refuses_a_buffer_without_the_slack(gpu_ctx):
    """spkm.h: the allocation behind d_rec must extend 256 bytes past the last record (the record kernels fetch ahead of their
    bounds check).  No size crosses the C boundary, so the entry point asks the runtime for the allocation the pointer lies
    in: a raw hipMalloc of exactly n * R bytes is refused with SPKM_ERR_BAD_VALUE, one with the slack is taken."""
    import ctypes as C
    import os

    from sparsifiedkmeans_amd import _lib
    from sparsifiedkmeans_amd.engine import record_bytes

    L = _lib.lib()
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=C.RTLD_GLOBAL)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    p, n, s = 1024, 8192, 51
    R = record_bytes(s, 16)
    for extra, ok in ((0, False), (255, False), (256, True)):
        buf = C.c_void_p()
        assert hip.hipMalloc(C.byref(buf), n * R + extra) == 0
        try:
            h = C.c_void_p()
            rc = L.spkm_shard_create_rec_dev(gpu_ctx.handle, p, n, s, 16, buf, C.byref(h))
            if ok:
                assert rc == 0 and h.value
                L.spkm_shard_destroy(h)
            else:
                assert rc == _lib.ERR_BAD_VALUE, (extra, rc)
                assert b"slack" in L.spkm_ctx_last_error(gpu_ctx.handle)
        finally:
            torch.cuda.synchronize()
            hip.hipFree(buf)
