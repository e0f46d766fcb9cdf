import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Some GPU tests use torch tensors next to libsvo_hip.so (the RCCL gather path).  PyTorch-ROCm ships its own copy
    # of the HIP runtime: it has to be the first one loaded into the process, otherwise torch finds no device.
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load
