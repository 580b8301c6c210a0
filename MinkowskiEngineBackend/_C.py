"""`MinkowskiEngineBackend._C` — the module object IS `minkowskiengine_amd.backend` (same enums, CoordinateMapKey,
CoordinateMapManagerGPU_c10 / _default, <Op>{Forward,Backward}GPU functions as pybind/extern.hpp:515-838 registers
for the hot path).  Operators outside the hot path (interpolation, channelwise convolution, spmm,
the CPU operators) are absent: `get_minkowski_function` (MinkowskiCommon.py:110-120) raises for them."""
import sys

from minkowskiengine_amd import backend as _backend

sys.modules[__name__] = _backend
