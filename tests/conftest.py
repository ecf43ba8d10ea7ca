import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def smml():
    import importlib
    return importlib.import_module("subspace-multimodal-learning_amd")


@pytest.fixture(scope="session")
def cuda(smml):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    smml.lib()                      # raises if the HIP library is missing: no silent fallback
    from importlib import import_module
    capi = import_module("subspace-multimodal-learning_amd._capi")
    capi.check(capi.lib().smml_device_check(0), "device check")
    return torch.device("cuda:0")
