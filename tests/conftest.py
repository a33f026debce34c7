import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hc():
    """The product package (its directory name has a hyphen)."""
    return importlib.import_module("hipcomp-core_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    return O


@pytest.fixture(scope="session")
def reflib(hc):
    """The reference's own low-level build (oracle/_ref), or None if absent."""
    from oracle import oracle as O
    if not os.path.exists(O.REF_LIB_PATH):
        return None
    return hc.HipcompLibrary(O.REF_LIB_PATH)


@pytest.fixture(scope="session")
def cuda():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    torch.cuda.set_device(0)
    return torch.device("cuda:0")
