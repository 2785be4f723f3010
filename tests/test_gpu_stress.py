"""-m gpu: a short run of tools/stress_parity.py -- random shapes, orders, starts with duplicate centroids and empty
clusters; every iteration of the fused call compared with the oracle (assignments and distances bit for bit, centres to
1e-9).  The long form (minutes, other seeds) is run by hand; its last result is in profiles/r02_stress_parity.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [3, 4])
def test_random_shapes_against_the_oracle(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), "30", str(seed)],
                       capture_output=True, text=True, timeout=600)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "0 failures" in r.stdout, tail
    assert int(r.stdout.strip().splitlines()[-1].split()[0]) >= 1, tail      # it did run
