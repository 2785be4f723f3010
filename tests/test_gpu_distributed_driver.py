"""-m gpu: the host driver with two ranks.  The GPU box has one device, so both ranks share cuda:0 and talk
over gloo (torch's gloo backend accepts device tensors); RCCL itself is exercised by the driver's own
multi-GPU bench.  What is pinned here: two ranks cluster exactly the dataset one process would."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, start, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.distributed import shard_range
    from sparsifiedkmeans_amd.kmeans import kmeans_sparsified

    X, centres, labels = synth.gmm_dense(128, 4001, 6, seed=12)
    lo, hi = shard_range(4001, rank, world)
    S = X[:, [0, 700, 1400, 2100, 2800, 3500]].T if start == "matrix" else start
    IDX, C, SUMD, D, OUT = kmeans_sparsified(X[:, lo:hi].T, 6, Sparsify=True, SparsityLevel=0.1, Start=S, rng=5,
                                             first=lo, n_total=4001, MaxIter=30)
    q.put((rank, lo, hi, IDX, C, SUMD, D, OUT["iterations"], OUT["objectives"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("start", ["matrix", "Arthur", "sample"])
def test_two_ranks_cluster_the_same_dataset(gpu_ctx, start):
    sys.path.insert(0, ROOT)
    from sparsifiedkmeans_amd import synth
    from sparsifiedkmeans_amd.kmeans import kmeans_sparsified

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, start, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    X, centres, labels = synth.gmm_dense(128, 4001, 6, seed=12)
    S = X[:, [0, 700, 1400, 2100, 2800, 3500]].T if start == "matrix" else start
    IDX, C, SUMD, D, OUT = kmeans_sparsified(X.T, 6, Sparsify=True, SparsityLevel=0.1, Start=S, rng=5, MaxIter=30)
    got_idx = np.concatenate([r[3] for r in res])
    got_d = np.concatenate([r[6] for r in res])
    assert [r[1:3] for r in res] == [(0, 2000), (2000, 4001)]
    assert np.array_equal(got_idx, IDX)                                   # same assignments as one process
    assert np.allclose(got_d, D, rtol=1e-9, atol=0)
    for r in res:
        assert np.abs(r[4] - C).max() <= 1e-9 * np.abs(C).max()           # centres: summation order only
        assert np.allclose(r[5], SUMD, rtol=1e-9)
        assert r[7][0] == OUT["iterations"][0]
    assert np.array_equal(res[0][4], res[1][4])                            # bit-identical on both ranks
