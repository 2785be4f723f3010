"""-m gpu: RCCL inside libspkm.so (include/spkm.h Part 3: spkm_comm_*, spkm_allreduce_f64_dev, spkm_lloyd_iter).
The single-rank communicator runs on any GPU box; the 2-rank test needs two devices and is skipped otherwise, so the
first multi-GPU box that runs the suite exercises RCCL over xGMI."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import parts, random_csc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_communicator_whole_iteration_in_one_call(oracle):
    from sparsifiedkmeans_amd import _lib
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard, attach_rccl, comm_size, detach_rccl, torch_context

    ctx = torch_context(0)                                   # its own context: the communicator hangs off it
    assert comm_size(ctx) == 0
    assert attach_rccl(ctx) == 1 and comm_size(ctx) == 1
    with pytest.raises(_lib.SpkmError):                      # one communicator per context
        ident = (__import__("ctypes").c_uint8 * 128)()
        _lib.check(_lib.lib().spkm_comm_unique_id(ident))
        _lib.check(_lib.lib().spkm_comm_init(ctx.handle, 1, 0, ident))
    p, n, K, s = 256, 20000, 12, 16
    X = random_csc(p, n, s, seed=77)
    C0 = np.random.default_rng(5).standard_normal((K, p))
    eng = LloydEngine(Shard.from_scipy(ctx, X), K, s / p)
    centers = torch.tensor(C0, device="cuda")
    hist = [eng.iterate(centers).cpu().numpy().copy() for _ in range(4)]     # spkm_lloyd_iter, ncclAllReduce on 1 rank
    ref = oracle.lloyd(p, n, *parts(X), C0.T, s / p, maxiter=4, tol=0.0)
    assert np.array_equal(eng.assign.cpu().numpy(), ref["assign"])
    assert np.allclose(eng.mind.cpu().numpy(), ref["mind"], rtol=1e-9, atol=0)
    assert np.abs(centers.cpu().numpy().T - ref["centers"]).max() <= 1e-9 * np.abs(ref["centers"]).max()
    assert np.allclose(np.sqrt([h[1] for h in hist]), ref["obj"], rtol=1e-9)
    assert np.allclose(np.sqrt([h[0] for h in hist]), ref["dff"], rtol=1e-6)
    # the stand-alone all-reduce on one rank leaves the buffer alone
    buf = torch.arange(10, dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().spkm_allreduce_f64_dev(ctx.handle, buf.data_ptr(), 10))
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy(), np.arange(10.0))
    detach_rccl(ctx)
    assert comm_size(ctx) == 0


def test_lloyd_iter_without_a_communicator_equals_the_three_calls(gpu_ctx, oracle):
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, K, s = 128, 9000, 5, 8
    X = random_csc(p, n, s, seed=3)
    C0 = np.random.default_rng(1).standard_normal((K, p))
    shard = Shard.from_scipy(gpu_ctx, X)
    for unbiased in (True, False):
        a = LloydEngine(shard, K, s / p, unbiased=unbiased)
        b = LloydEngine(shard, K, s / p, unbiased=unbiased)
        ca, cb = torch.tensor(C0, device="cuda"), torch.tensor(C0, device="cuda")
        for _ in range(3):
            a.iterate(ca)
            b.assign_accumulate_step(cb)
            b.allreduce_step()
            b.finalize_step(cb)
            assert np.array_equal(a.assign.cpu().numpy(), b.assign.cpu().numpy())
            assert np.array_equal(a.mind.cpu().numpy(), b.mind.cpu().numpy())
            cb.copy_(ca)                                     # (sums are atomics: keep the two loops on identical centres)
        ref = oracle.lloyd(p, n, *parts(X), C0.T, s / p, unbiased=unbiased, maxiter=3, tol=0.0)
        assert np.array_equal(a.assign.cpu().numpy(), ref["assign"])


def test_lloyd_iter_host_hands_the_host_what_it_decides_on_without_a_copy(gpu_ctx, oracle):
    """spkm_lloyd_iter_host = spkm_lloyd_iter + [dff^2, obj^2, cluster sizes] in host memory when the call returns (pinned
    memory the device maps, a sequence number behind the values): the same numbers as the device outputs of the same
    call, the oracle's assignments and objective, and a K that grows between calls (the mapped buffer is re-made)."""
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard

    p, n, s = 128, 9000, 8
    X = random_csc(p, n, s, seed=3)
    shard = Shard.from_scipy(gpu_ctx, X)
    for K in (5, 200):        # (the buffer starts out with room for 125 cluster sizes)
        C0 = np.random.default_rng(1).standard_normal((K, p))
        eng = LloydEngine(shard, K, s / p)
        c = torch.tensor(C0, device="cuda")
        ref = oracle.lloyd(p, n, *parts(X), C0.T, s / p, maxiter=4, tol=0.0)
        for it in range(4):
            used = c.clone()
            host = eng.iterate_host(c)
            # (no synchronisation before looking at `host`: the call itself waited for the values)
            assert host.shape == (2 + K,)
            torch.cuda.synchronize()
            out = eng.out.cpu().numpy()
            pk = p * K
            assert host[0] == out[0] and (host[1] == out[1] or (np.isnan(host[1]) and np.isnan(out[1])))
            assert np.array_equal(host[2:], eng.reduce[2 * pk: 2 * pk + K].cpu().numpy())
            assert np.array_equal(host[2:], np.bincount(eng.assign.cpu().numpy(), minlength=K).astype(np.float64))
            assert host[0] == float(((used - c) ** 2).sum().item()) or abs(host[0] - float(((used - c) ** 2).sum().item())) <= 1e-12 * host[0]
            assert abs(np.sqrt(host[1]) - ref["obj"][it]) <= 1e-9 * ref["obj"][it]
        assert np.array_equal(eng.assign.cpu().numpy(), ref["assign"])


def _check_two_ranks(tmp_path, oracle, env_extra, port, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_rccl_worker.py"), str(tmp_path), "4"]
    subprocess.run(cmd, check=True, env=env, timeout=timeout)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(2)]
    p, n, K, s = 256, 20000, 12, 16
    X = random_csc(p, n, s, seed=77)
    C0 = np.random.default_rng(5).standard_normal((K, p))
    ref = oracle.lloyd(p, n, *parts(X), C0.T, s / p, maxiter=4, tol=0.0)
    assert np.array_equal(r[0]["centers"], r[1]["centers"])                 # every rank finalises the same centres
    assign = np.concatenate([r[0]["assign"], r[1]["assign"]])
    assert int(r[0]["hi"]) == int(r[1]["lo"]) and assign.size == n
    assert np.array_equal(assign, ref["assign"])
    assert np.abs(r[0]["centers"].T - ref["centers"]).max() <= 1e-9 * np.abs(ref["centers"]).max()
    assert np.allclose(np.sqrt(r[0]["hist"][:, 1]), ref["obj"], rtol=1e-9)   # obj2 is a global sum


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_two_ranks_over_rccl_match_one_process(tmp_path, oracle):
    _check_two_ranks(tmp_path, oracle, {}, 29531)


def test_communicator_that_cannot_form_fails_on_every_rank_and_the_loop_falls_back(tmp_path, oracle):
    """Two ranks on ONE device: RCCL refuses the communicator.  Every rank must come back with SPKM_ERR_COMM (no rank
    left waiting in ncclCommInitRank), nothing stays attached and LloydEngine.iterate exchanges through
    torch.distributed (gloo here) -- same answers as one process."""
    _check_two_ranks(tmp_path, oracle, {"SPKM_TEST_ONE_DEVICE": "1", "SPKM_TEST_BACKEND": "gloo"}, 29533, timeout=240)
