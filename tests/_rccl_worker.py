"""Worker of tests/test_gpu_rccl.py (launched with torch.distributed.run, one process per GPU): a few Lloyd iterations
on this rank's block of a seeded dataset, the all-reduce issued by libspkm.so's own RCCL communicator."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, iters = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    one_device = bool(os.environ.get("SPKM_TEST_ONE_DEVICE"))       # both ranks on cuda:0: RCCL must refuse, not hang
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    backend = os.environ.get("SPKM_TEST_BACKEND", "nccl")
    dist.init_process_group(backend, device_id=torch.device("cuda", local) if backend == "nccl" else None)
    from sparsifiedkmeans_amd import distributed as D_
    from sparsifiedkmeans_amd.engine import LloydEngine, Shard, attach_rccl, comm_size, detach_rccl, torch_context
    from util import random_csc

    from sparsifiedkmeans_amd import _lib

    ctx = torch_context(local)
    if one_device:
        # a communicator that cannot be formed is reported on EVERY rank (the ranks vote before and after
        # ncclCommInitRank), nothing stays attached, and the iteration falls back to torch.distributed
        try:
            attach_rccl(ctx)
            raise AssertionError("two ranks on one device formed a communicator")
        except _lib.SpkmError as e:
            assert e.status == _lib.ERR_COMM, e
        assert comm_size(ctx) == 0
    else:
        assert attach_rccl(ctx) == world and comm_size(ctx) == world
    p, n, K, s = 256, 20000, 12, 16
    X = random_csc(p, n, s, seed=77)
    lo, hi = D_.shard_range(n, rank, world)
    shard = Shard.from_scipy(ctx, X[:, lo:hi].tocsc())
    eng = LloydEngine(shard, K, s / p)
    C0 = np.random.default_rng(5).standard_normal((K, p))
    centers = torch.tensor(C0, device=f"cuda:{local}")
    hist = []
    for _ in range(iters):
        hist.append(eng.iterate(centers).cpu().numpy().copy())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), centers=centers.cpu().numpy(), assign=eng.assign.cpu().numpy(),
             mind=eng.mind.cpu().numpy(), hist=np.array(hist), lo=lo, hi=hi)
    if not one_device:
        detach_rccl(ctx)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
