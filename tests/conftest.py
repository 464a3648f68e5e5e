import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _poison_store_once_gradient_slots(request):
    """GPU tests run with the store-once gradient slots (engine.ParamSet) pre-filled with NaN: a weight whose gradient GEMM did
    not run -- and was not zeroed by Tape.backward either -- then fails every gradient comparison instead of reading stale memory."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from toist_amd import engine
    old = engine.POISON_FRESH
    engine.POISON_FRESH = True
    yield
    engine.POISON_FRESH = old
