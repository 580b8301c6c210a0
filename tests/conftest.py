import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The kernel-internals tests reach into the Python twin of the operator module (backend.py: kernel maps, plans, tuning
# switches), so the test process STARTS on the python host.  Every oracle / reference-fixture test that goes through the
# public API takes the `host_layer` fixture below and runs TWICE: on the Python twin and on the native C++ host layer
# (csrc_host/ -> _me_host.so), which is the product default — the oracle checks what ships.
os.environ.setdefault("ME_AMD_HOST", "python")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=["python", "native"])
def host_layer(request):
    """Run the test under both host layers (minkowskiengine_amd.set_host): the ctypes twin and the native extension.
    A missing native layer FAILS (it is the product default), it does not skip."""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import host
    if request.param == "native" and host.native_module() is None:
        pytest.fail(f"native host layer not available: {host.native_error()}")
    prev = ME.get_host()
    ME.set_host(request.param)
    try:
        yield request.param
    finally:
        ME.set_host(prev)
