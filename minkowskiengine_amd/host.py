"""Which operator module serves the Python API: the NATIVE one (csrc_host/ -> _me_host.so: C++ manager, operators and
autograd functions over the C ABI — what the reference builds as `MinkowskiEngineBackend._C`) or the Python twin
(`backend.py`, ctypes over the same C ABI — the test harness that the kernel tests poke into).

    ME_AMD_HOST=native   (default when _me_host.so is built)     ME_AMD_HOST=python
    minkowskiengine_amd.set_host("native" | "python")            # switchable at run time; objects made under one host
                                                                 # (keys, managers, sparse tensors) stay with that host
Both run the same kernels with the same plans, so results are bit-identical (tests/test_gpu_native_host.py)."""
import importlib.machinery
import importlib.util
import os

from . import backend as _python_backend

_HERE = os.path.dirname(os.path.abspath(__file__))
NATIVE_PATH = os.path.join(_HERE, "_me_host.so")
_native = None
_native_error = None


def native_module():
    """-> the native operator module, or None when it is not built / cannot be loaded (reason in native_error())"""
    global _native, _native_error
    if _native is None and _native_error is not None and "not built" in _native_error and os.path.exists(NATIVE_PATH):
        _native_error = None          # built since the first look (__graft_entry__.build() on a clean tree): look again
    if _native is None and _native_error is None:
        if not os.path.exists(NATIVE_PATH):
            _native_error = f"{NATIVE_PATH} not built (python -c 'import __graft_entry__ as g; g.build()')"
        else:
            try:
                import torch  # noqa: F401  (libtorch must be loaded first)
                from . import _lib
                _lib.load()       # libme_amd.so by its in-tree path (the extension links it through $ORIGIN as well)
                loader = importlib.machinery.ExtensionFileLoader("_me_host", NATIVE_PATH)
                spec = importlib.util.spec_from_file_location("_me_host", NATIVE_PATH, loader=loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                _native = mod
            except Exception as e:  # noqa: BLE001
                _native_error = f"{type(e).__name__}: {e}"
    return _native


def native_error():
    return _native_error


_mode = os.environ.get("ME_AMD_HOST", "auto")
if _mode not in ("auto", "native", "python"):
    raise RuntimeError(f"ME_AMD_HOST must be auto, native or python, got {_mode!r}")
if _mode == "native" and native_module() is None:
    raise RuntimeError(f"ME_AMD_HOST=native but the native host layer is not available: {native_error()}")
_current = "native" if (_mode != "python" and native_module() is not None) else "python"
if _mode == "auto" and _current == "python":
    import warnings
    warnings.warn("minkowskiengine_amd: the native host layer (_me_host.so) is not available — "
                  f"{native_error()} — falling back to the slower Python host (ME_AMD_HOST=python silences this, "
                  "ME_AMD_HOST=native makes it an error)", RuntimeWarning, stacklevel=2)


def get_host():
    return _current


def set_host(name):
    global _current
    if name == "native" and native_module() is None:
        raise RuntimeError(f"native host layer not available: {native_error()}")
    if name not in ("native", "python"):
        raise ValueError(name)
    _current = name


def is_native():
    return _current == "native"


def backend():
    """the operator module in charge: the native extension or minkowskiengine_amd.backend"""
    return _native if _current == "native" else _python_backend


def backend_of(obj):
    """The operator module that OWNS `obj` — a coordinate map key, a native / Python manager or the CoordinateManager
    wrapper of either: objects stay with the host they were made under, whatever set_host() says now."""
    if obj is not None:
        flag = getattr(obj, "_native", None)          # CoordinateManager wrapper (coordinate_manager.py)
        if isinstance(flag, bool):
            return _native if flag else _python_backend
        if _native is not None and isinstance(obj, (_native.CoordinateMapKey, _native.CoordinateMapManagerGPU_c10)):
            return _native
        if isinstance(obj, (_python_backend.CoordinateMapKey, _python_backend.CoordinateMapManagerGPU_c10)):
            return _python_backend
    return backend()


def key_like(key, *args):
    """A new CoordinateMapKey of the host that owns `key`: (coordinate size of `key`) by default, or the given
    constructor arguments."""
    return backend_of(key).CoordinateMapKey(*(args if args else (key.get_coordinate_size(),)))


def invalidate_packed_weights(params=None):
    """Repack cached weight images (both hosts) at their next use: for weight updates the tensor version counter cannot
    see, i.e. writes through `p.data`.  params=None: every image; an iterable of tensors: only the images of weights that
    live inside the storage of those tensors (what the package's optimizer-step hook passes: the stepping optimizer's
    own parameters; a kernel that is an offset view of a parameter is matched by its address range).
    If NONE of the cached images belongs to `params` the optimizer does not hold the model's tensors at all — fp32 master
    copies written back through `.data.copy_` (Apex / DeepSpeed style), exactly the updates the hook exists for: then
    every image goes stale (the global epoch), as before round 5."""
    if params is None:
        _python_backend.invalidate_packed_weights()
        if _native is not None:
            _native.invalidate_packed_weights()
        return
    ranges = sorted({(int(p.data_ptr()), int(p.data_ptr()) + int(p.numel()) * int(p.element_size())) for p in params
                     if p.numel() > 0})
    merged = []                      # (overlapping parameter storages — a flat buffer and its views — become one range)
    for a, b in ranges:
        if merged and a <= merged[-1][1]:
            merged[-1] = (merged[-1][0], max(merged[-1][1], b))
        else:
            merged.append((a, b))
    ranges = merged
    m0, t0 = _python_backend.invalidate_packed_weights(None, ranges)
    m1, t1 = (0, 0)
    if _native is not None:
        m1, t1 = _native.invalidate_packed_weights_in([a for a, _ in ranges], [b for _, b in ranges])
    if (t0 + t1) > 0 and (m0 + m1) == 0:
        invalidate_packed_weights(None)


def set_grad_destination(param, dest):
    """Both hosts: the gradient of `param` (a convolution kernel, a batch-norm weight / bias; fp32) is written into
    `dest` (None: forget it) instead of a fresh tensor — distributed.GradientArena."""
    _python_backend.set_grad_destination(param, dest)
    if _native is not None:
        _native.set_grad_destination(param, dest)


def arm_grad_destinations(ptrs=None):
    """ptrs=None: every registered destination; a list of parameter addresses: only those (GradientArena arms its own)"""
    if ptrs is None:
        _python_backend.arm_grad_destinations()
        if _native is not None:
            _native.arm_grad_destinations()
        return
    _python_backend.arm_grad_destinations(ptrs)
    if _native is not None:
        _native.arm_grad_destinations_for(list(ptrs))


def drop_grad_destinations(ptrs):
    """forget the destinations of these parameter addresses on both hosts (an arena that goes away)"""
    _python_backend.drop_grad_destinations(ptrs)
    if _native is not None:
        _native.drop_grad_destinations(list(ptrs))


def clear_grad_destinations():
    _python_backend.clear_grad_destinations()
    if _native is not None:
        _native.clear_grad_destinations()


def _key_types():
    return (_python_backend.CoordinateMapKey,) + ((_native.CoordinateMapKey,) if _native is not None else ())


class _KeyMeta(type):
    def __call__(cls, *args, **kwargs):
        return backend().CoordinateMapKey(*args, **kwargs)

    def __instancecheck__(cls, obj):
        return isinstance(obj, _key_types())


class CoordinateMapKey(metaclass=_KeyMeta):
    """`CoordinateMapKey(coordinate_size)` / `CoordinateMapKey(tensor_stride, string_id)` of the host in charge
    (src/coordinate_map_key.hpp:44-157); isinstance() accepts the keys of either host."""
