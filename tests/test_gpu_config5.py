"""-m gpu: BASELINE.json config 5 at ONE GPU's shard size -- 1.25e8 points x 784 (-> 1024), K = 10, streamed from
pinned host memory in chunks through the device sparsifier (the shape of private/sampleAndMixFromLargeFile.m:79-129:
chunk -> X*(1+2eps) -> mix -> sample -> append), then Lloyd iterations on the resident sparse shard.
Streamed chunks are checked against the ORACLE's transform and the numpy restatement of the sampler (not against the
in-memory HIP run)."""
import numpy as np
import pytest
import torch

from util import sample_rows_reference

pytestmark = pytest.mark.gpu

N_SHARD = 125_000_000          # 1e9 points over 8 GPUs


def test_config5_shard_streamed_ingest_and_lloyd(gpu_ctx, oracle, capsys):
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard, mix_device

    torch.cuda.empty_cache()                                       # (blocks cached by earlier tests are not "used")
    free, _ = torch.cuda.mem_get_info()
    if free < 185e9:
        pytest.skip("needs ~175 GB of free HBM at its peak (CSC arrays 64 GB + record layout 64 GB + screen copy 38 GB, "
                    "while the layouts are built from the arrays); ~115 GB resident once the arrays are released")
    p, K, seed = 784, 10, 77
    data = synth.streamed_pixel_dataset(gpu_ctx, p, N_SHARD, 0, K, 0.05, seed=seed, chunk=131072)
    p2, s, P = data["p2"], data["s"], data["pool_points"]
    assert (p2, s) == (1024, 51) and data["ingest"]["points"] == N_SHARD
    assert data["ingest"]["bytes"] == N_SHARD * p                  # one byte per value crossed PCIe
    gbs = data["ingest"]["GBs"]
    with capsys.disabled():
        print(f"\n[config 5] streamed {N_SHARD} points ({data['ingest']['bytes'] / 1e9:.1f} GB of uint8) in "
              f"{data['ingest']['seconds']:.2f} s = {gbs:.1f} GB/s over PCIe (Gen5 x16 spec 63 GB/s)")
    assert gbs > 8.0
    # ---- parity of streamed columns with the oracle: first chunk, a chunk boundary, a pool wrap-around, the very end ----
    d = data["sign"].cpu().numpy()
    pool = data["pool"].numpy()
    x_all, ir_all = data["x"], data["ir"]
    for lo in (0, 131072 - 8, P - 8, 3 * P + 12345, N_SHARD - 16):
        cnt = 16
        dense = pool[(lo + np.arange(cnt)) % P].astype(np.float64).T          # p x cnt
        Xm = oracle.mix(dense, d, p2)                                          # (1+2eps) pre-scale, sign, pad, FWHT, /sqrt(p2)
        rows = sample_rows_reference(seed, lo, cnt, p2, s)                     # Philox keyed by (seed, global index)
        want = Xm[rows, np.arange(cnt)[:, None]] / (np.float64(s) / np.float64(p2))
        got_x = x_all[lo * s:(lo + cnt) * s].cpu().numpy().reshape(cnt, s)
        got_r = ir_all[lo * s:(lo + cnt) * s].cpu().numpy().view(np.uint16).astype(np.int64).reshape(cnt, s)
        assert np.array_equal(got_r, rows), f"sampled rows differ at point {lo}"
        assert np.array_equal(got_x, want), f"sampled values differ at point {lo}"
    # ---- K = 10 Lloyd iterations on the resident shard (the fused call), from the planted means ----
    shard = Shard.from_device(gpu_ctx, p2, data["jc"], data["ir"], data["x"], nnz=data["nnz"])
    eng = LloydEngine(shard, K, data["gamma"])
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    start = data["means"] + 10.0 * torch.randn((K, p), generator=g, device="cuda", dtype=torch.float64)
    centers = mix_device(gpu_ctx, start.contiguous(), p2, data["sign"], 1.0, 32.0)
    # host copies of the first points for the spot check at the end: the device arrays are about to go
    n_chk = 4096
    irh = ir_all[: n_chk * s].cpu().numpy().view(np.uint16).astype(np.uint64)
    xh = x_all[: n_chk * s].cpu().numpy()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    for it in range(5):
        out = eng.iterate(centers)
        ev[it + 1].record()
        if it == 0:
            # the first fused call has built the record layout and the screen copy: the CSC arrays can go
            # (spkm_shard_release_csc) -- 64 GB of the shard's ~180 GB
            torch.cuda.synchronize()
            assert shard.release_csc()
            del x_all, ir_all
            data.pop("x"); data.pop("ir")
            torch.cuda.empty_cache()
            free_now, total_now = torch.cuda.mem_get_info()
            resident = (total_now - free_now) / 1e9
            with capsys.disabled():
                print(f"[config 5] resident after releasing the CSC arrays: {resident:.1f} GB")
            assert resident < 130.0, resident
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    assert eng.last_path_info()[0] == 1                                # screen + exact confirmation, as the benchmark
    with capsys.disabled():
        print(f"[config 5] Lloyd iterations on the shard (K=10): {[round(v, 1) for v in ms]} ms")
    # the clusters are far apart: every one of the first P points (one pass over the pool) sits with its planted cluster
    a = eng.assign[:P].cpu().numpy()
    lab = data["labels_pool"].numpy()
    m = np.zeros((K, K), np.int64)
    np.add.at(m, (a, lab), 1)
    assert (m > 0).sum() == K and m.max(axis=1).sum() == P
    nk = eng.global_nk().cpu().numpy()
    assert nk.sum() == N_SHARD and nk.min() > 0.08 * N_SHARD
    # spot check of the exact outputs against the oracle on the first 4096 points with the centres of the last call
    # (teacher-forced: same X, same centres in -> same assignments / distances out)
    c_before = centers.clone()
    eng.iterate(centers)
    jc = np.arange(0, (n_chk + 1) * s, s, dtype=np.uint64)
    ra, rd = oracle.assign(p2, n_chk, jc, irh, xh, c_before.cpu().numpy().T, data["gamma"])
    assert np.array_equal(eng.assign[:n_chk].cpu().numpy(), ra)
    assert np.array_equal(eng.mind[:n_chk].cpu().numpy(), rd)


def test_streaming_sparsifier_accepts_narrow_dtypes_and_pageable_sources(gpu_ctx, oracle):
    """uint8 / int16 / float32 / float64 chunks, numpy (pageable) and pinned torch sources, ragged chunk sizes: the
    resident shard equals oracle.mix + the numpy sampler, whatever the route."""
    from sparsifiedkmeans_amd.engine import StreamingSparsifier

    p, n, s, seed = 200, 3000, 26, 5
    p2 = 256
    rng = np.random.default_rng(1)
    d = np.sign(rng.standard_normal(p2))
    base = rng.integers(0, 256, size=(n, p))
    sign = torch.tensor(d, device="cuda")
    rows = sample_rows_reference(seed, 40, n, p2, s)
    for dtype in (np.uint8, np.int16, np.float32, np.float64):
        src = base.astype(dtype)
        Xm = oracle.mix(src.astype(np.float64).T, d, p2)
        want = Xm[rows, np.arange(n)[:, None]] / (np.float64(s) / np.float64(p2))
        for pinned in (False, True):
            sp_ = StreamingSparsifier(gpu_ctx, p, n, s, seed, sign, first=40)
            c0 = 0
            for m in (700, 1, 1299, 1000):
                blk = src[c0:c0 + m]
                sp_.append(torch.from_numpy(np.ascontiguousarray(blk)).pin_memory() if pinned else blk)
                c0 += m
            shard = sp_.finish()
            torch.cuda.synchronize()
            got_x = sp_.x[: n * s].cpu().numpy().reshape(n, s)
            got_r = sp_.ir[: n * s].cpu().numpy().view(np.uint16).astype(np.int64).reshape(n, s)
            assert np.array_equal(got_r, rows) and np.array_equal(got_x, want), (dtype, pinned)
            assert shard.n == n


def test_streaming_sparsifier_can_append_records(gpu_ctx, oracle):
    """layout="records": the streamed chunks land in the library's record layout (spkm_mix_sample_rec_dev) and the shard
    adopts them (spkm_shard_create_rec_dev): same rows and values as the CSC form of the same stream, column by column
    through the library's own read-back, and the same assignments from a fused call."""
    from sparsifiedkmeans_amd.engine import LloydEngine, StreamingSparsifier

    p, n, s, seed, K = 200, 2500, 26, 9, 6
    p2 = 256
    rng = np.random.default_rng(2)
    d = np.sign(rng.standard_normal(p2))
    src = rng.integers(0, 256, size=(n, p)).astype(np.uint8)
    sign = torch.tensor(d, device="cuda")
    shards = {}
    for layout in ("csc", "records"):
        sp_ = StreamingSparsifier(gpu_ctx, p, n, s, seed, sign, first=7, layout=layout)
        c0 = 0
        for m in (900, 3, 1597):
            sp_.append(src[c0:c0 + m])
            c0 += m
        shards[layout] = (sp_, sp_.finish())
    torch.cuda.synchronize()
    assert shards["records"][0].records and not shards["csc"][0].records
    csc = shards["csc"][0]
    rows = csc.ir[: n * s].cpu().numpy().view(np.uint16).astype(np.int64).reshape(n, s)
    vals = csc.x[: n * s].cpu().numpy().reshape(n, s)
    for i in (0, 1, 899, 900, 903, n - 1):
        r, v = shards["records"][1].column(i)
        assert np.array_equal(r, rows[i]) and np.array_equal(v, vals[i]), i
    c = torch.tensor(np.ascontiguousarray((rng.standard_normal((p2, K)) * 30.0).T), device="cuda")
    out = {}
    for layout in ("csc", "records"):
        eng = LloydEngine(shards[layout][1], K, s / p)
        cc = c.clone()
        eng.iterate(cc)
        out[layout] = (eng.assign.cpu().numpy(), eng.mind.cpu().numpy())
    assert np.array_equal(out["csc"][0], out["records"][0]) and np.array_equal(out["csc"][1], out["records"][1])
