"""CoordinateManager: thin Python wrapper that owns a backend coordinate-map manager
(reference: MinkowskiEngine/MinkowskiCoordinateManager.py:107-440)."""
import collections
import contextlib
import os
import threading
import weakref

import torch

from . import host as _host
from .backend import CoordinateMapType, GPUMemoryAllocatorType, MinkowskiAlgorithm, RegionType
from .host import CoordinateMapKey
from .common import convert_to_int_list

_allocator_type = GPUMemoryAllocatorType.PYTORCH
_coordinate_map_type = CoordinateMapType.CUDA
_minkowski_algorithm = MinkowskiAlgorithm.DEFAULT


# Map prefetch (not in the reference): when on, a SparseTensor that creates its own CoordinateManager replays, right after
# its coordinates are inserted, the map-building requests that the PREVIOUS scene's manager served (strided maps, kernel
# maps, tile plans) — a training loop builds the same maps for every scene, and building them in one burst keeps the
# host read-backs of the build out of the forward pass (backend.CoordinateMapManagerGPU_c10.prefetch, docs/HISTORY.md 9.8).
_map_prefetch = os.environ.get("ME_AMD_MAP_PREFETCH", "0") != "0"
_recent_managers = collections.deque(maxlen=4)   # weak references to the latest CoordinateManagers made by SparseTensors
# (D, native host?, tag) -> [recipe, misses]: the request log a manager left behind when it was destroyed — a scene's manager
# usually dies right after its step, which is exactly when its log is complete
_published_recipes = {}
_PUBLISH_PATIENCE = 8   # that many successive shorter logs replace a longer published one (the network has changed)


def set_map_prefetch(enabled=True):
    global _map_prefetch
    _map_prefetch = bool(enabled)


def map_prefetch_enabled():
    return _map_prefetch


# Two networks of the same dimension in one process (train + eval, student + teacher, alternating models): the scenes of
# the smaller one would replay the larger one's recipe — the longest log wins — and build maps and plans it never uses,
# every scene.  A tag keeps their request logs apart: managers created inside `with map_prefetch_tag("eval"):`
# publish and replay only logs of the same tag (per thread; the default tag is "").
_prefetch_tag = threading.local()


@contextlib.contextmanager
def map_prefetch_tag(tag):
    prev = getattr(_prefetch_tag, "value", "")
    _prefetch_tag.value = str(tag)
    try:
        yield
    finally:
        _prefetch_tag.value = prev


def _prefetch_from_previous(manager):
    """Called by SparseTensor for a freshly created manager whose coordinates have just been inserted.  The recipe is
    the LONGEST request log among the latest managers that are still alive: with a loader thread (utils.ScenePrefetcher)
    the directly preceding manager has usually not run its step yet — its log is empty unless it was itself built from
    a complete recipe, in which case it is complete at once (a replay logs what it serves) — and the log the last
    destroyed manager left behind (CoordinateManager.__del__)."""
    if _map_prefetch:
        best = None
        for ref in _recent_managers:
            prev = ref()
            if (prev is None or prev is manager or prev.D != manager.D or prev._native != manager._native
                    or prev._tag != manager._tag):
                continue                         # (a request log belongs to the host layer — and the tag — that wrote it)
            r = prev.recipe()
            if best is None or len(r) > len(best):
                best = r
        pub = _published_recipes.get((manager.D, manager._native, manager._tag))
        if pub is not None and (best is None or len(pub[0]) > len(best)):
            best = pub[0]
        if best:
            manager.prefetch(best)
    _recent_managers.append(weakref.ref(manager))


def set_coordinate_map_type(coordinate_map_type):
    global _coordinate_map_type
    _coordinate_map_type = coordinate_map_type


def set_gpu_allocator(backend):
    """MinkowskiCoordinateManager.py:63-89.  Both values use the torch caching allocator here."""
    assert isinstance(backend, GPUMemoryAllocatorType)
    global _allocator_type
    _allocator_type = backend


def set_memory_manager_backend(backend):
    set_gpu_allocator(backend)


class CoordinateManager:
    def __init__(self, D=0, num_threads=-1, coordinate_map_type=None, allocator_type=None,
                 minkowski_algorithm=None):
        if D < 1:
            raise ValueError(f"Invalid rank D > 0, D = {D}.")
        if num_threads < 0:
            num_threads = min(os.cpu_count() or 1, 20)
        if coordinate_map_type is None:
            coordinate_map_type = _coordinate_map_type
        if allocator_type is None:
            allocator_type = _allocator_type
        if minkowski_algorithm is None:
            minkowski_algorithm = _minkowski_algorithm
        if coordinate_map_type == CoordinateMapType.CPU:
            raise RuntimeError("minkowskiengine_amd has no CPU coordinate map: the MI355X path keeps maps in HBM. "
                               "Pass GPU coordinates.")
        B = _host.backend()       # the native operator module, or backend.py (host.py)
        self._CoordinateManagerClass = (B.CoordinateMapManagerGPU_c10
                                        if allocator_type == GPUMemoryAllocatorType.PYTORCH
                                        else B.CoordinateMapManagerGPU_default)
        self._manager = self._CoordinateManagerClass(int(minkowski_algorithm), num_threads)
        self._native = _host.is_native()
        self._tag = getattr(_prefetch_tag, "value", "")     # see map_prefetch_tag
        self.D = D
        self.minkowski_algorithm = minkowski_algorithm

    def __del__(self):
        # leave the request log behind for the managers of later scenes (see _prefetch_from_previous)
        try:
            if not _map_prefetch:
                return
            r = self._manager.recipe()
            if not r:
                return
            key = (self.D, self._native, self._tag)
            pub = _published_recipes.get(key)
            if pub is None or len(r) >= len(pub[0]):
                _published_recipes[key] = [r, 0]
            else:
                pub[1] += 1
                if pub[1] >= _PUBLISH_PATIENCE:
                    _published_recipes[key] = [r, 0]
        except Exception:      # interpreter shutdown
            pass

    # ---- build-request log (not in the reference; see set_map_prefetch) ---------------------------------------------
    def recipe(self):
        return self._manager.recipe()

    def prefetch(self, recipe):
        return self._manager.prefetch(recipe)

    def record_stream(self, stream):
        """See backend.CoordinateMapManagerGPU_c10.record_stream (maps built on a side stream)."""
        self._manager.record_stream(stream)

    # ---- maps -----------------------------------------------------------------------------------
    def insert_and_map(self, coordinates, tensor_stride=1, string_id=""):
        """-> (CoordinateMapKey, (unique_map, inverse_map)); MinkowskiCoordinateManager.py:153-179"""
        tensor_stride = convert_to_int_list(tensor_stride, self.D)
        return self._manager.insert_and_map(coordinates, tensor_stride, string_id)

    def stride(self, coordinate_map_key, stride, string_id=""):
        stride = convert_to_int_list(stride, self.D)
        return self._manager.stride(coordinate_map_key, stride, string_id)

    def size(self, coordinate_map_key):
        return self._manager.size(coordinate_map_key)

    def exists_coordinate_map_key(self, coordinate_map_key):
        return self._manager.exists(coordinate_map_key)

    def get_coordinates(self, coords_key_or_tensor_strides):
        key = coords_key_or_tensor_strides
        if not isinstance(key, CoordinateMapKey):
            key = _host.key_like(self, convert_to_int_list(key, self.D), "")
        return self._manager.get_coordinates(key)

    def origin(self):
        """Key of the origin map: one row per batch index (MinkowskiCoordinateManager.py:224-226)."""
        return self._manager.origin()

    def origin_map(self, key):
        return self._manager.origin_map(key)

    def origin_map_size(self):
        return self._manager.origin_map_size()

    def stride_map(self, in_key, stride_key):
        """(in rows, rows of the strided map) as two int64 tensors (MinkowskiCoordinateManager.py:429-430)"""
        return self._manager.stride_map(in_key, stride_key)

    def union_map(self, in_keys, out_key):
        """one int64 [2, n_i] tensor per input key: (its rows, rows of the union map created under out_key)"""
        return self._manager.union_map(in_keys, out_key)

    def get_unique_coordinate_map_key(self, tensor_stride):
        ts = convert_to_int_list(tensor_stride, self.D)
        sid = self._manager.get_random_string_id(ts, "")
        return _host.key_like(self, sid[0], sid[1])

    def get_coordinate_map_keys(self, tensor_stride):
        return self._manager.get_coordinate_map_keys(convert_to_int_list(tensor_stride, self.D))

    # ---- kernel maps ----------------------------------------------------------------------------
    def kernel_map(self, in_key, out_key, stride=1, kernel_size=3, dilation=1, region_type=RegionType.HYPER_CUBE,
                   region_offset=None, is_transpose=False, is_pool=False):
        """dict {k: int32 [2, n_k]}; MinkowskiCoordinateManager.py:377-421"""
        D = in_key.get_coordinate_size() - 1
        if region_offset is None:
            region_offset = torch.IntTensor()
        return self._manager.kernel_map(in_key, out_key, convert_to_int_list(kernel_size, D),
                                        convert_to_int_list(stride, D), convert_to_int_list(dilation, D),
                                        region_type, region_offset, is_transpose, is_pool)

    get_kernel_map = kernel_map   # MinkowskiCoordinateManager.py:349-375 (older name, same arguments)

    def number_of_unique_batch_indices(self):
        """MinkowskiCoordinateManager.py:334-335"""
        return self._manager.origin_map_size()

    def print_coordinate_map(self, coordinate_map_key):
        """the map's line of repr(manager) (MinkowskiCoordinateManager.py: `_manager.print_coordinate_map`,
        pybind/extern.hpp:777-779)"""
        return self._manager.print_coordinate_map(coordinate_map_key)

    def __repr__(self):
        return f"{self.__class__.__name__}(\n{self._manager!r}\n)"
