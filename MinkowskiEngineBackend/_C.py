"""`MinkowskiEngineBackend._C` — the operator module the reference's Python package imports (setup.py:312,
MinkowskiEngine/__init__.py): here it IS this repository's operator module — the native C++ extension
(minkowskiengine_amd/_me_host.so, csrc_host/) when it is built, else its Python twin minkowskiengine_amd.backend (same
enums, CoordinateMapKey, CoordinateMapManagerGPU_*, <Op>{Forward,Backward}GPU names, argument order and meaning)."""
import sys

from minkowskiengine_amd import host as _host

sys.modules[__name__] = _host.backend()
