import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    from sparsifiedkmeans_amd.engine import torch_context

    return torch_context(0)


@pytest.fixture(autouse=True)
def _restore_switches(request, monkeypatch):
    """after every -m gpu test: environment back to what it was, and the session context's switches re-read"""
    yield
    if request.node.get_closest_marker("gpu") is None or "gpu_ctx" not in request.fixturenames:
        return
    monkeypatch.undo()
    request.getfixturevalue("gpu_ctx").reload_switches()
