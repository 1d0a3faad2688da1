import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]

    return load


@pytest.fixture(scope="session")
def dev():
    import torch

    if os.environ.get("PL_EMULATE") == "1":
        # development aid (tests/emu_backend.py): run `-m gpu` tests against the CPU-emulated kernels, e.g.
        #   PL_EMULATE=1 python -m pytest tests/test_gpu_parity.py -m gpu -k canny
        # Small cases only (a fiber per work-item); never a substitute for the run on the MI355X.
        from emu_backend import emulated_device

        with emulated_device():
            yield torch.device("cuda:0")
        return
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    yield torch.device("cuda:0")
