"""CPU, world_size 2 over gloo: the data-parallel protocol of the Lloyd iteration -- contiguous
point shards, ONE sum all-reduce of [sums | counts | nk | obj2], identical finalise on every
rank, and the MAXLOC + broadcast of EmptyAction='singleton'.  The per-shard arithmetic is done
by the CPU oracle here (the HIP kernels need a GPU; their own parity is covered by -m gpu), so
this test pins the exchange, the buffer layout and the partition."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from sparsifiedkmeans_amd import distributed as D
    from util import parts, random_csc

    p, n, K, gamma = 64, 1001, 5, 0.125
    X = random_csc(p, n, 8, seed=11)              # every rank builds the same global matrix ...
    C = np.random.default_rng(5).standard_normal((p, K))
    C[:, 3] = 50.0                                 # ... with one centre nobody wants (empty cluster)
    first, last = D.shard_range(n, rank, world)
    Xl = X[:, first:last].tocsc()                  # ... and keeps only its own block of points
    lay = D.reduce_layout(p, K)
    hist = []
    for it in range(3):
        a, d = O.assign(p, last - first, *parts(Xl), C, gamma)
        S, Cnt, nk = O.accumulate(p, last - first, K, *parts(Xl), a)
        buf = torch.zeros(lay["length"], dtype=torch.float64)
        buf[lay["sums"]] = torch.from_numpy(np.ascontiguousarray(S.T).ravel())
        buf[lay["counts"]] = torch.from_numpy(np.ascontiguousarray(Cnt.T).ravel())
        buf[lay["nk"]] = torch.from_numpy(nk.astype(np.float64))
        buf[lay["obj2"]] = float(np.sum(d * d))
        D.allreduce_(buf)                          # the ONE exchange of the iteration
        Sg = buf[lay["sums"]].numpy().reshape(K, p).T
        Cg = buf[lay["counts"]].numpy().reshape(K, p).T
        nkg = buf[lay["nk"]].numpy().astype(np.int64)
        C = O.finalize_centers(Sg, Cg, nkg, gamma, C)
        owner = gidx = None
        if (nkg == 0).any():                       # EmptyAction='singleton' (kmeans_sparsified.m:436-437)
            li = int(np.argmax(d))
            owner, gidx, val = D.global_first_argmax(float(d[li]), li, first)
            col = torch.zeros(p, dtype=torch.float64)
            if rank == owner:
                col = torch.from_numpy(X[:, gidx].toarray().ravel().copy())
            D.broadcast_column(col, owner)
            for k in np.flatnonzero(nkg == 0):
                C[:, k] = col.numpy()
        hist.append((C.copy(), float(buf[lay["obj2"]]), nkg.copy(), owner, gidx, a.copy()))
    q.put((rank, first, last, hist))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_lloyd_matches_single_process():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    from util import parts, random_csc

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    # single-process reference on the whole matrix
    p, n, K, gamma = 64, 1001, 5, 0.125
    X = random_csc(p, n, 8, seed=11)
    C = np.random.default_rng(5).standard_normal((p, K))
    C[:, 3] = 50.0
    assert [r[1:3] for r in res] == [(0, 500), (500, 1001)]
    for it in range(3):
        a, d = O.assign(p, n, *parts(X), C, gamma)
        S, Cnt, nk = O.accumulate(p, n, K, *parts(X), a)
        C = O.finalize_centers(S, Cnt, nk, gamma, C)
        if (nk == 0).any():
            imax = int(np.argmax(d))
            for k in np.flatnonzero(nk == 0):
                C[:, k] = X[:, imax].toarray().ravel()
        for rank, first, last, hist in res:
            Cr, obj2, nkr, owner, gidx, ar = hist[it]
            assert np.array_equal(ar, a[first:last])                        # assignments: bit-identical per shard
            assert np.array_equal(nkr, nk)
            assert abs(obj2 - np.sum(d * d)) <= 1e-12 * np.sum(d * d)
            assert np.abs(Cr - C).max() <= 1e-12 * np.abs(C).max()          # all-reduce changes summation order only
            if (nk == 0).any():
                assert gidx == imax and owner == (0 if imax < 500 else 1)
        # both ranks hold bit-identical centres (no broadcast needed)
        assert np.array_equal(res[0][3][it][0], res[1][3][it][0])
        C = res[0][3][it][0]
