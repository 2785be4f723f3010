"""-m gpu: `python bench.py --gpus 2` is a complete command -- with no launcher around it, it starts one process per
rank itself (torch.distributed.run, 127.0.0.1 rendezvous), runs the sharded Lloyd loop with the per-iteration
all-reduce and prints ONE JSON line.  On the one-GPU box both ranks share cuda:0 and the exchange goes through gloo
(SPKM_BENCH_ONE_DEVICE / SPKM_BENCH_BACKEND: RCCL refuses two ranks on one device); on a multi-GPU box the same test
runs the real thing over RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(gpus, extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--n-total", "2e6", "--steps", "6",
           "--warmup", "1", "--cpu-sample", "0", "--no-regimes"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    import torch

    two = torch.cuda.device_count() >= 2
    env = {} if two else {"SPKM_BENCH_ONE_DEVICE": "1", "SPKM_BENCH_BACKEND": "gloo"}
    one = _run(1, {})
    res = _run(2, env)
    assert res["n_gpus"] == 2 and res["steps"] == 6 and res["value"] > 0
    assert res["config"]["n_per_gpu"] == 1_000_000 and res["config"]["n_total"] == 2_000_000
    assert ("RCCL" in res["config"]["allreduce"]) == two or "torch.distributed" in res["config"]["allreduce"]
    # the same global dataset and start on 1 and on 2 ranks: the same Lloyd run (objective to summation order)
    assert abs(res["config"]["final_obj"] - one["config"]["final_obj"]) <= 1e-9 * one["config"]["final_obj"]
    for key in ("roofline", "metric", "unit", "ms_per_step", "scaling"):
        assert key in res
