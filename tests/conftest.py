import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The kernel tests reach into the Python twin of the operator module (backend.py: kernel maps, plans, tuning switches),
# so the test process starts on the python host; tests/test_gpu_native_host.py switches to the native C++ host layer
# (minkowskiengine_amd.set_host) and checks that it gives the same bits.  The product default is the native host.
os.environ.setdefault("ME_AMD_HOST", "python")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
