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


def _poison_device_memory(dev, gib):
    """ME_TEST_POISON_GIB=<n>: fill n GiB of device memory with a pattern and hand it back to torch's caching allocator, so
    that every later `torch.empty` of the session starts from garbage instead of the zeros of a fresh box — a kernel that
    reads a buffer it never wrote (as an index!) then fails HERE and not on the one box of a pool whose memory holds the
    previous job's data (round 6: a suite run aborted three times on one reused box and passed on every fresh one).
    0x7f817f81 = a large positive int32, a huge int64 pair, a NaN as fp32 and as bf16."""
    import torch
    chunk = 1 << 30
    blocks = []
    for _ in range(int(gib)):
        t = torch.empty(chunk // 4, dtype=torch.int32, device=dev)
        t.fill_(0x7f817f81)
        blocks.append(t)
    # (requests of up to 1 MiB come from the allocator's SMALL pool — 2 MiB blocks of their own: counters, flags,
    # descriptors and the maps of small scenes live there)
    for _ in range(1024):
        t = torch.empty((1 << 20) // 4, dtype=torch.int32, device=dev)
        t.fill_(0x7f817f81)
        blocks.append(t)
    torch.cuda.synchronize()
    del blocks            # cached, not returned to the driver: the allocator splits these blocks for what follows


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    gib = float(os.environ.get("ME_TEST_POISON_GIB", "0") or 0)
    if gib > 0:
        _poison_device_memory(dev, gib)
    return dev


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
