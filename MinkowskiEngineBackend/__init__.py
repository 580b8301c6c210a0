"""Drop-in for the reference's native extension package: `MinkowskiEngineBackend._C` (built by the reference's
setup.py:312 from pybind/minkowski.cu) resolves to the MI355X operator module `minkowskiengine_amd.backend`
(HIP kernels behind the C ABI of include/me_amd.h).  With the repository root on sys.path the reference's OWN
Python package (`MinkowskiEngine/*.py`, unmodified) imports and runs its GPU path on top of it — see
INTEGRATION.md and tests/test_reference_package.py."""
from . import _C  # noqa: F401
