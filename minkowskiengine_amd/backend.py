"""Host-side mirror of the reference's operator module ``MinkowskiEngineBackend._C``
(/root/reference/pybind/extern.hpp:515-838) for the hot path, written over the C ABI of
include/me_amd.h.

Exported with the reference's names and argument meaning:
  enums      RegionType, ConvolutionMode, MinkowskiAlgorithm, CoordinateMapType,
             GPUMemoryAllocatorType, PoolingMode, BroadcastMode          (extern.hpp:669-741)
  classes    CoordinateMapKey (extern.hpp:744-764),
             CoordinateMapManagerGPU_c10 / CoordinateMapManagerGPU_default (extern.hpp:767-806)
  functions  ConvolutionForwardGPU / ConvolutionBackwardGPU,
             ConvolutionTransposeForwardGPU / ConvolutionTransposeBackwardGPU (extern.hpp:53-181)
             is_cuda_available, cuda_version, cudart_version, get_gpu_memory_info

Errors are RuntimeError, like the reference's ASSERT (src/utils.hpp:141-150).  There is no CPU
implementation here: CPU tensors are rejected (the reference's CPU path is the test oracle).
"""
import ctypes
import enum
import os
import weakref
import random
import string

import threading

import torch

from . import _lib


# ------------------------------------------------------------------------------------------------
# enums (pybind/extern.hpp:669-741)
# ------------------------------------------------------------------------------------------------
class GPUMemoryAllocatorType(enum.IntEnum):
    PYTORCH = 0
    CUDA = 1


class CUDAKernelMapMode(enum.IntEnum):
    MEMORY_EFFICIENT = 0
    SPEED_OPTIMIZED = 1


class MinkowskiAlgorithm(enum.IntEnum):
    DEFAULT = 0
    MEMORY_EFFICIENT = 1
    SPEED_OPTIMIZED = 2


class CoordinateMapType(enum.IntEnum):
    CPU = 0
    CUDA = 1


class RegionType(enum.IntEnum):
    HYPER_CUBE = 0
    HYPER_CROSS = 1
    CUSTOM = 2


class PoolingMode(enum.IntEnum):
    LOCAL_SUM_POOLING = 0
    LOCAL_AVG_POOLING = 1
    LOCAL_MAX_POOLING = 2
    GLOBAL_SUM_POOLING_DEFAULT = 3
    GLOBAL_AVG_POOLING_DEFAULT = 4
    GLOBAL_MAX_POOLING_DEFAULT = 5
    GLOBAL_SUM_POOLING_KERNEL = 6
    GLOBAL_AVG_POOLING_KERNEL = 7
    GLOBAL_MAX_POOLING_KERNEL = 8
    GLOBAL_SUM_POOLING_PYTORCH_INDEX = 9
    GLOBAL_AVG_POOLING_PYTORCH_INDEX = 10
    GLOBAL_MAX_POOLING_PYTORCH_INDEX = 11


class BroadcastMode(enum.IntEnum):
    ELEMENTWISE_ADDITON = 0
    ELEMENTWISE_MULTIPLICATION = 1


class ConvolutionMode(enum.IntEnum):
    DEFAULT = 0
    DIRECT_GEMM = 1
    COPY_GEMM = 2


def is_cuda_available():
    return True


def cuda_version():
    return -1


def cudart_version():
    return -1


def get_gpu_memory_info():
    free, total = torch.cuda.mem_get_info()
    return free, total


# roctx ranges (insert / stride / kernel map / plan / forward / dgrad / wgrad) for `rocprofv3
# --marker-trace`, when ME_AMD_ROCTX=1; the marker library is resolved at run time
_ROCTX = None
if os.environ.get("ME_AMD_ROCTX", "0") != "0":
    for _name in ("librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"):
        try:
            _ROCTX = ctypes.CDLL(_name)
            _ROCTX.roctxRangePushA.argtypes = [ctypes.c_char_p]
            break
        except (OSError, AttributeError):
            _ROCTX = None


class _roctx:
    """with _roctx("me:kernel_map"): a roctx range around the launches of the block (no-op unless ME_AMD_ROCTX=1)"""
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _ROCTX is not None:
            _ROCTX.roctxRangePushA(self.name.encode())

    def __exit__(self, *exc):
        if _ROCTX is not None:
            _ROCTX.roctxRangePop()
        return False


def _ranged(name):
    """decorator: the call runs inside a roctx range (ME_AMD_ROCTX=1), else untouched"""
    def deco(fn):
        if _ROCTX is None:
            return fn

        def wrapped(*a, **kw):
            with _roctx(name):
                return fn(*a, **kw)
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        return wrapped
    return deco


def _check(cond, *msg):
    if not cond:
        raise RuntimeError("assertion failed. " + " ".join(str(m) for m in msg))


# ------------------------------------------------------------------------------------------------
# CoordinateMapKey (src/coordinate_map_key.hpp:44-157)
# ------------------------------------------------------------------------------------------------
class CoordinateMapKey:
    """Identity of a coordinate map: (tensor_stride, string_id); may be created unset and filled in
    by an operator (lazy key, coordinate_map_key.hpp:52-53, 91-98)."""

    __slots__ = ("_coordinate_size", "_key")

    def __init__(self, arg, string_id=""):
        if isinstance(arg, int):
            self._coordinate_size = int(arg)
            self._key = None
        else:
            ts = [int(s) for s in arg]
            self._coordinate_size = len(ts) + 1
            self._key = (tuple(ts), str(string_id))

    def get_coordinate_size(self):
        return self._coordinate_size

    def is_key_set(self):
        return self._key is not None

    def set_key(self, *args):
        if len(args) == 1:
            tensor_stride, string_id = args[0]
        else:
            tensor_stride, string_id = args
        ts = tuple(int(s) for s in tensor_stride)
        _check(self._coordinate_size - 1 == len(ts), "Invalid tensor_stride size:", ts,
               "coordinate_size:", self._coordinate_size)
        self._key = (ts, str(string_id))

    def get_key(self):
        _check(self._key is not None, "Key not set")
        return (list(self._key[0]), self._key[1])

    def _hashable(self):
        _check(self._key is not None, "Key not set")
        return self._key

    def get_tensor_stride(self):
        _check(self._key is not None, "Key not set")
        return list(self._key[0])

    def __eq__(self, other):
        if not isinstance(other, CoordinateMapKey):
            return NotImplemented
        if self._key is None or other._key is None:
            return False
        return self._key == other._key

    def __hash__(self):
        return hash(self._hashable())

    def __repr__(self):
        if self._key is None:
            return "coordinate map key: (unset)"
        s = "coordinate map key:" + str(list(self._key[0]))
        if self._key[1]:
            s += ":" + self._key[1]
        return s


# ------------------------------------------------------------------------------------------------
# device objects
# ------------------------------------------------------------------------------------------------
def _ptr(t):
    """device address for a c_void_p argument (plain int / None: ctypes converts them without an object)"""
    return t.data_ptr() if t is not None else None


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class _on:
    """`with _on(dev):` = torch.cuda.device(dev) without the switch when `dev` is already current (the
    per-layer host time matters: a bf16 MinkUNet34C step is as long on the host as on the GPU)."""
    __slots__ = ("dev", "guard")

    def __init__(self, dev):
        self.dev = dev
        self.guard = None

    def __enter__(self):
        idx = self.dev.index
        if idx is not None and idx != torch.cuda.current_device():
            self.guard = torch.cuda.device(self.dev)
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
            self.guard = None
        return False


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


_SUPERCELL_SHIFT = {1: 12, 2: 6, 3: 4, 4: 3, 5: 2, 6: 2, 7: 1}   # log2 side: <= 4096 cells per supercell
_MIN_SUPERCELL_SHIFT = {1: 6, 2: 4, 3: 3, 4: 2, 5: 1, 6: 1, 7: 1}
_MIN_SUPERCELLS = 1024      # aim for at least this many supercells (workgroups of the probe kernel)
_MAX_SUPERCELLS = 1 << 22   # beyond this the dense supercell directory is not built (flat-table probes instead)


class _SpatialIndex:
    """Rows of a coordinate map in supercell order (csrc/coords.hip, me_spatial_index_build): `order` int32 [n]
    position -> row, `pos_of_row` its inverse, `coords_sorted` the coordinates in position order, `dir_start` uint32
    [m + 1] the first position of every supercell of the dense directory described by `grid`."""
    __slots__ = ("grid", "m", "order", "pos_of_row", "coords_sorted", "dir_start")


class _CoordinateMapGPU:
    """One coordinate map resident in HBM: unique int32 coordinates [n, D+1] in row order and the
    open-addressing table {hash tag | row} (replaces CoordinateMapGPU,
    src/coordinate_map_gpu.cuh:47-223); `bbox` (host ints: column minima, then maxima) came back with the
    insert's own read-back and sizes the spatial index, which is built on first use."""

    __slots__ = ("coords", "table", "capacity", "tensor_stride", "n", "bbox", "_spatial", "_zorder", "_zorder_inv")

    def __init__(self, coords, table, capacity, tensor_stride, n, bbox=None):
        self.coords, self.table, self.capacity = coords, table, capacity
        self.tensor_stride, self.n = tuple(tensor_stride), int(n)
        self.bbox = bbox
        self._spatial = False    # False: not tried yet; None: not available
        self._zorder = None
        self._zorder_inv = None

    def zorder_inv(self):
        """row -> its position in zorder() (int32 [n])"""
        if self._zorder_inv is None and self.n > 0:
            z = self.zorder()
            inv = torch.empty_like(z)
            inv[z.long()] = torch.arange(self.n, dtype=torch.int32, device=z.device)
            self._zorder_inv = inv
        return self._zorder_inv

    def zorder(self):
        """Rows in Z-order (Morton keys of the coordinates in units of the tensor stride, batch index on top;
        me_coords_spatial_keys): runs of consecutive entries are spatially compact at EVERY length — what the halo
        kernel's tiles need (the supercell order of spatial() is compact only down to a supercell).  int32 [n]."""
        if self._zorder is None and self.n > 0:
            lib = _lib.load()
            dev = self.coords.device
            ncol = int(self.coords.shape[1])
            ts = (ctypes.c_int32 * len(self.tensor_stride))(*self.tensor_stride)
            bb = None
            if self.bbox is not None and len(self.bbox) == 2 * ncol:
                bb = (ctypes.c_int32 * (2 * ncol))(*[int(v) for v in self.bbox])
            order = torch.empty(self.n, dtype=torch.int32, device=dev)
            ws = torch.empty(int(lib.me_coords_zorder_workspace_bytes(self.n)), dtype=torch.uint8, device=dev)
            with _on(dev):
                # the library's own radix sort over the key bytes that can differ inside the bounding box (round 6: no
                # torch.argsort on the map path)
                _lib.check(lib.me_coords_zorder(_ptr(self.coords), self.n, ncol, ts, bb, _ptr(order), _ptr(ws), ws.numel(),
                                                _stream(dev)))
            self._zorder = order
        return self._zorder

    def spatial(self):
        """-> _SpatialIndex or None (empty map, no bounding box, or a bounding box of too many supercells)."""
        if self._spatial is not False:
            return self._spatial
        self._spatial = None
        ncol = int(self.coords.shape[1])
        D = ncol - 1
        if self.n == 0 or self.bbox is None or D not in _SUPERCELL_SHIFT or any(t <= 0 for t in self.tensor_stride):
            return None
        lib = _lib.load()
        g = _lib.MeSpatialGrid()
        g.ncol = ncol
        g.sc_min[0] = int(self.bbox[0])
        g.sc_dim[0] = int(self.bbox[ncol]) - int(self.bbox[0]) + 1
        # supercell side: the largest (<= 4096 cells) that still gives the probe kernel a few workgroups per CU — a
        # 70^3 scene in 16^3 supercells is only 125 workgroups for 256 CUs
        sh = _SUPERCELL_SHIFT[D]
        while True:
            m = g.sc_dim[0]
            for d in range(D):
                ts = int(self.tensor_stride[d])
                lo = (int(self.bbox[1 + d]) // ts) >> sh          # python floor division / arithmetic shift
                hi = (int(self.bbox[ncol + 1 + d]) // ts) >> sh
                g.shift[d], g.tensor_stride[d] = sh, ts
                g.sc_min[1 + d], g.sc_dim[1 + d] = lo, hi - lo + 1
                m *= hi - lo + 1
            if m >= _MIN_SUPERCELLS or sh <= _MIN_SUPERCELL_SHIFT.get(D, 1):
                break
            sh -= 1
        m = int(lib.me_spatial_cells(ctypes.byref(g)))
        if m < 1 or m > _MAX_SUPERCELLS:
            return None
        dev = self.coords.device
        sp = _SpatialIndex()
        sp.grid, sp.m = g, m
        sp.order = torch.empty(self.n, dtype=torch.int32, device=dev)
        sp.pos_of_row = torch.empty(self.n, dtype=torch.int32, device=dev)
        sp.coords_sorted = torch.empty((self.n, ncol), dtype=torch.int32, device=dev)
        sp.dir_start = torch.empty(m + 1, dtype=torch.int32, device=dev)
        ws = _workspace(lib.me_spatial_index_workspace_bytes(self.n, m), dev)
        with _on(dev):
            _lib.check(lib.me_spatial_index_build(_ptr(self.coords), self.n, ctypes.byref(g), _ptr(sp.order),
                                                  _ptr(sp.pos_of_row), _ptr(sp.coords_sorted), _ptr(sp.dir_start),
                                                  _ptr(ws), ws.numel(), _stream(dev)))
        self._spatial = sp
        return sp


def _insert(coords, tensor_stride):
    """coords: int32 [N, D+1] contiguous on the GPU -> (_CoordinateMapGPU, unique_map, inverse_map)."""
    lib = _lib.load()
    dev = coords.device
    n, ncol = int(coords.shape[0]), int(coords.shape[1])
    cap = int(lib.me_hash_capacity(n))
    table = torch.empty(cap, dtype=torch.int64, device=dev)
    coords_unique = torch.empty((max(n, 1), ncol), dtype=torch.int32, device=dev)
    unique_map = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    inverse_map = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    wsb = int(lib.me_insert_workspace_bytes(n))
    ws = _workspace(wsb, dev)
    n_unique = ctypes.c_int64(0)
    bbox = (ctypes.c_int32 * (2 * ncol))()
    with _on(dev):
        _lib.check(lib.me_coords_insert_and_map_bbox(_ptr(coords), n, ncol, _ptr(table), cap, _ptr(coords_unique),
                                                     _ptr(unique_map), _ptr(inverse_map), ctypes.byref(n_unique),
                                                     bbox, _ptr(ws), ws.numel(), _stream(dev)))
    nu = int(n_unique.value)
    cmap = _CoordinateMapGPU(coords_unique[:nu], table, cap, tensor_stride, nu, bbox=list(bbox) if n > 0 else None)
    return cmap, unique_map[:nu], inverse_map[:n]


class _LazyOffsets:
    """The per-offset pair prefix of a kernel map, read back from the device WITHOUT stalling the build: the copy into
    pinned host memory is enqueued right behind the counting kernels, the host waits for it only when a host value
    is first needed (launch geometry of the weight gradient, plan sizes, the dict view)."""
    __slots__ = ("pinned", "event", "values")

    def __init__(self, k_offsets_dev):
        self.pinned = torch.empty(k_offsets_dev.numel(), dtype=torch.int64, pin_memory=True)
        self.pinned.copy_(k_offsets_dev, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(k_offsets_dev.device))
        self.values = None

    def get(self):
        if self.values is None:
            self.event.synchronize()
            self.values = self.pinned.tolist()
            self.pinned = self.event = None
        return self.values


class KernelMapGPU:
    """Kernel map resident in HBM (replaces gpu_kernel_map, src/kernel_map.cuh:48-429).

    * ``in_pairs`` / ``out_pairs`` + ``k_offsets``: the reference's per-offset (in row, out row)
      lists, concatenated; offset k owns [k_offsets[k], k_offsets[k+1]).
    * neighbour tables nbr[k, target] -> source ROW or -1 for both sides.  Maps built by the LDS-bucketed probe hold
      them in POSITION space (target = rank of the target row in its map's supercell order, ``order`` maps positions
      back to rows): ``table_pos(target)`` -> (table, order | None) is what the kernels consume (tiles of
      consecutive positions are spatially compact); ``table(target)`` is the row-space view (tests, pooling).
    * the tile plans of the target-stationary convolution are built on first use and cached.
    A transposed view (``swapped()``) shares all device buffers with in/out roles exchanged
    (src/coordinate_map_manager.cpp:763-774).
    """

    def __init__(self, volume, n_in, n_out, k_offsets, k_offsets_dev, in_pairs, out_pairs, store=None,
                 flip=False, in_map=None, out_map=None):
        self.volume, self.n_in, self.n_out = int(volume), int(n_in), int(n_out)
        self.in_map, self.out_map = in_map, out_map   # _CoordinateMapGPU of either side (spatial tile order)
        self._k_offsets = k_offsets           # host list of volume+1 ints, or _LazyOffsets
        self.k_offsets_dev = k_offsets_dev    # int64 [volume+1] on the device
        self.in_pairs_buf, self.out_pairs_buf = in_pairs, out_pairs   # may be longer than n_pairs (upper bound)
        self._store = store if store is not None else {}
        self._flip = flip
        self._launch_cache = {}               # per view: launch geometry and device addresses of the plans
        self._recipe, self._recipe_key = None, None   # the owning manager's request log (CoordinateMapManager.prefetch)
        # sides on which a row has AT MOST one pair by construction (bit 0: "in" rows, bit 1: "out" rows; csrc_host twin:
        # KernelMap::one_pair_sides) — with n_pairs == rows of the side: exactly one -> row-wise launch (conv_rowwise.hip)
        self.one_pair_sides = 0

    @property
    def k_offsets(self):
        if isinstance(self._k_offsets, _LazyOffsets):
            self._k_offsets = self._k_offsets.get()
        return self._k_offsets

    @property
    def n_pairs(self):
        return int(self.k_offsets[-1])

    @property
    def in_pairs(self):
        return self.in_pairs_buf[:self.n_pairs]

    @property
    def out_pairs(self):
        return self.out_pairs_buf[:self.n_pairs]

    @property
    def device(self):
        return self.k_offsets_dev.device

    def _name(self, kind, target):
        if self._flip:
            target = "in" if target == "out" else "out"
        return kind + "_" + target

    def swapped(self):
        km = KernelMapGPU(self.volume, self.n_out, self.n_in, self._k_offsets, self.k_offsets_dev,
                          self.out_pairs_buf, self.in_pairs_buf, store=self._store, flip=not self._flip,
                          in_map=self.out_map, out_map=self.in_map)
        km.one_pair_sides = ((self.one_pair_sides & 1) << 1) | ((self.one_pair_sides >> 1) & 1)
        return km

    def table_pos(self, target):
        """(table, order): table [volume, n_tgt] of source ROWS indexed by target POSITION, order int32 [n_tgt]
        position -> target row, or None when positions are rows (flat-table maps)."""
        name = self._name("nbr", target)
        order = self._store.get(self._name("order", target))
        if name not in self._store:
            lib = _lib.load()
            dev = self.device
            # the missing table is the transpose of the existing one: scatter the pair lists
            n_tgt = self.n_out if target == "out" else self.n_in
            src_pairs = self.in_pairs_buf if target == "out" else self.out_pairs_buf   # values stored
            tgt_pairs = self.out_pairs_buf if target == "out" else self.in_pairs_buf   # rows indexed
            pos = self._store.get(self._name("pos", target))
            tbl = torch.empty((self.volume, max(n_tgt, 1)), dtype=torch.int32, device=dev)
            bound = (self.n_out if target == "in" else self.n_in) * self.volume
            with _on(dev):
                _lib.check(lib.me_kernel_map_transpose_ordered(_ptr(tgt_pairs), _ptr(src_pairs),
                                                               _ptr(self.k_offsets_dev), self.volume, bound, n_tgt,
                                                               _ptr(pos), _ptr(tbl), _stream(dev)))
            self._store[name] = tbl
        return self._store[name], order

    def table(self, target):
        """target 'out': [volume, n_out] -> in row; target 'in': [volume, n_in] -> out row (ROW space)."""
        tbl, order = self.table_pos(target)
        if order is None:
            return tbl
        name = self._name("nbrrow", target)
        if name not in self._store:
            row = torch.empty_like(tbl)
            row[:, order.long()] = tbl[:, :order.numel()]
            self._store[name] = row
        return self._store[name]

    def order(self, target, tile_order=None):
        """Target rows in tile order (int32 [n_tgt]: tile position -> row) for the convolution kernels' stores, or
        None when tiles are runs of consecutive rows.  `tile_order` "spatial": the supercell order of the target's
        coordinate map (tiles are spatially compact: their gathers hit the L2) — the map's own position space when
        it was built by the LDS-bucketed probe, else the spatial index of the target map, built on first use;
        "rows": runs of consecutive rows (a position-space table is then read through pos_of_row)."""
        tile_order = tile_order or self._tile_order(target)
        native = self._store.get(self._name("order", target))
        if native is not None:
            return native if tile_order == "spatial" else None
        return self._flat_order(target, tile_order)

    def _tile_order(self, target, matrix_bound=False, src_bytes=0):
        """"rows" | "spatial" for a launch family.  `matrix_bound`: the fp32 kernels on the bf16 matrix pipe
        (csrc/conv_f32x3.hip) — their tile time does not depend on how even the tiles are, so they always take the
        spatially compact tiles (config 2: 1.5 - 1.6x the compulsory HBM traffic instead of 3.2x at the same
        speed, profiles/r02_pmc_traffic_tile_order_final.log); the latency-bound kernels keep row tiles while the
        neighbour table is small enough for the plan builder's scattered reads."""
        if _TILE_ORDER != "auto":
            return _TILE_ORDER
        if matrix_bound and self._store.get(self._name("order", target)) is None:
            # Z-order tiles on row-space tables (csrc_host/manager.cpp KernelMap::tile_order: smaller source halos)
            cmap = self.out_map if target == "out" else self.in_map
            if cmap is not None and cmap.n > 0:
                return "zorder"
        if matrix_bound or (_TILE_SPATIAL_MIN_SRC_BYTES > 0 and src_bytes >= _TILE_SPATIAL_MIN_SRC_BYTES):
            return "spatial"
        n_tgt = self.n_out if target == "out" else self.n_in
        return "rows" if self.volume * n_tgt * 4 <= _TILE_ORDER_ROWS_MAX_BYTES else "spatial"

    def _flat_order(self, target, tile_order):
        """Tile permutation of a row-space (flat-table) map: the supercell order of the target's coordinate map for
        "spatial" tiles (None when that map has no spatial index: the tiles are then row tiles), the argsort of the
        Z-order keys with ME_AMD_SPATIAL_TILES=1 (round-1 experiment), else None."""
        cmap = self.out_map if target == "out" else self.in_map
        if tile_order == "zorder" and cmap is not None and cmap.n > 0:
            return cmap.zorder()
        if tile_order == "spatial" and cmap is not None and cmap.n > 0:
            sp = cmap.spatial()
            if sp is not None:
                return sp.order
        if not _SPATIAL_TILES:
            return None
        name = self._name("zorder", target)
        if name not in self._store:
            if cmap is None or cmap.n == 0:
                self._store[name] = None
            else:
                self._store[name] = cmap.zorder()       # (me_coords_zorder: the library's own sort)
        return self._store[name]

    @_ranged("me:tile_plan")
    def plan(self, target, tile_rows, batch_groups, tile_order=None):
        """Tile plan with `target` rows stationary, tiles of `tile_rows` rows and batches of at most
        `batch_groups` groups: (plan_src, plan_dst, batch_desc, tile_bptr, item_gptr); built once per
        (target, tile_rows, batch_groups, tile order)."""
        tile_order = tile_order or self._tile_order(target)
        name = self._name("plan", target) + f"_{int(tile_rows)}_{int(batch_groups)}_{tile_order}"
        if name not in self._store:
            lib = _lib.load()
            dev = self.device
            n_tgt = self.n_out if target == "out" else self.n_in
            tbl, native = self.table_pos(target)
            max_groups = int(lib.me_plan_max_groups(n_tgt, self.volume, self.n_pairs, tile_rows))
            n_tiles = int(lib.me_plan_num_tiles(n_tgt, tile_rows))
            plan_src = torch.empty(max_groups * _lib.ME_GROUP_ROWS, dtype=torch.int32, device=dev)
            plan_dst = torch.empty(max_groups * _lib.ME_GROUP_ROWS, dtype=torch.int32, device=dev)
            batch_desc = torch.empty(2 * max_groups, dtype=torch.int32, device=dev)
            tile_bptr = torch.empty(int(lib.me_plan_tile_bptr_elems(n_tgt, tile_rows)), dtype=torch.int32, device=dev)
            item_gptr = torch.empty(n_tiles * self.volume + 1, dtype=torch.int32, device=dev)
            ws = _workspace(lib.me_plan_workspace_bytes(n_tgt, self.volume, tile_rows), dev)
            # a position-space table is read in its own order (tiles = runs of positions); a row-space table goes
            # through the optional Z-order permutation
            if native is not None:
                gather_order = None if tile_order == "spatial" else self._store[self._name("pos", target)]
            else:
                gather_order = self._flat_order(target, tile_order)
            with _on(dev):
                _lib.check(lib.me_plan_build(_ptr(tbl), _ptr(gather_order), n_tgt, self.volume, tile_rows,
                                             batch_groups, _ptr(plan_src), _ptr(plan_dst), _ptr(batch_desc),
                                             _ptr(tile_bptr), _ptr(item_gptr), _ptr(ws), ws.numel(), _stream(dev)))
            self._store[name] = (plan_src, plan_dst, batch_desc, tile_bptr, item_gptr)
        return self._store[name]

    def to_dict(self):
        """{k: int32 [2, n_k]} with row 0 = in rows, row 1 = out rows; only non-empty k
        (src/coordinate_map_manager.cpp:1358-1387)."""
        out = {}
        for k in range(self.volume):
            b, e = self.k_offsets[k], self.k_offsets[k + 1]
            if e > b:
                out[k] = torch.stack((self.in_pairs_buf[b:e], self.out_pairs_buf[b:e]))
        return out


# Kernel-map build: "auto" = the LDS-bucketed build (spatial index + k_kmap_probe_lds, position-space tables) where it
# pays — measured (profiles/r02_rocprof_kernel_stats_kmap_{conv3d,conv4d}_{flat,lds}.csv): 2.1x faster than the flat-table probe on config 5 (K = 81, 32 M
# probes), on par on config 2 (K = 27, 2.7 M probes), where the one-off spatial index of the map (~110 us) makes the
# cold path slower; True / False (ME_AMD_SPATIAL_MAPS=1 / 0) force one path.
_SPATIAL_MAPS = {"1": True, "0": False}.get(os.environ.get("ME_AMD_SPATIAL_MAPS", "auto"), "auto")
_SPATIAL_MIN_VOLUME = 64            # auto: kernels of at least this many offsets ...
_SPATIAL_MIN_PROBES = 1 << 24       # ... or maps of at least this many (row, offset) probes
# Tile order of the convolution plans: "rows" = runs of consecutive rows (a position-space table is read through
# pos_of_row), "spatial" = runs of positions in the supercell order of the target map (spatially compact tiles: 2 -
# 2.6x less HBM-side traffic in the convolution; flat-table maps take the order from the target coordinate map's
# spatial index).  "auto" decides per launch family (KernelMapGPU._tile_order): the fp32 kernels on the bf16 matrix
# pipe always take spatial tiles (same speed, docs/HISTORY.md 9.12); the others — 3 - 12 % slower on spatial tiles of uniform
# random scenes — keep row tiles while the table is small enough for the plan builder's scattered reads (<= 32 MiB:
# L2 / Infinity Cache resident) and take spatial tiles beyond (config 5: 130 MB table — the row-order plan took
# 2.0 ms instead of 0.37)
_TILE_ORDER = os.environ.get("ME_AMD_TILE_ORDER", "auto")
_TILE_ORDER_ROWS_MAX_BYTES = 32 << 20
# ... and (round 4) the bf16 launches whose SOURCE matrix no longer fits the eight 4 MB L2s: MinkUNet34C's 96-channel
# layers on 160k - 200k voxels (31 - 38 MB) run 8 - 13 % faster on spatial tiles (both column slabs of a tile and its
# neighbours find the gathered rows in L2), every smaller layer 0 - 9 % slower (profiles/r04_tile_dispatch_sweep.log).
# ME_AMD_TILE_SPATIAL_SRC_MB: the threshold in MiB, 0 = never.
_TILE_SPATIAL_MIN_SRC_BYTES = int(os.environ.get("ME_AMD_TILE_SPATIAL_SRC_MB", "28")) << 20


def _build_kernel_map_lds(in_map, out_map, region, volume):
    """LDS-bucketed build (csrc/coords.hip k_kmap_probe_lds): the neighbour table comes out in the position space of
    the out map; no host synchronisation.  None when the pair of maps / the region is not eligible."""
    if _SPATIAL_MAPS is False or in_map.tensor_stride != out_map.tensor_stride or in_map.n == 0 or out_map.n == 0:
        return None
    if _SPATIAL_MAPS == "auto" and volume < _SPATIAL_MIN_VOLUME and out_map.n * volume < _SPATIAL_MIN_PROBES:
        return None
    lib = _lib.load()
    sq, sl = out_map.spatial(), in_map.spatial()
    if sq is None or sl is None:
        return None
    if int(lib.me_kernel_map_probe_lds_bytes(ctypes.byref(region), ctypes.byref(sq.grid), ctypes.byref(sl.grid))) < 0:
        return None
    dev = in_map.coords.device
    n_out, n_in = out_map.n, in_map.n
    nbr = torch.empty((volume, n_out), dtype=torch.int32, device=dev)
    ws = _workspace(lib.me_kernel_map_workspace_bytes(n_out, volume), dev)
    k_offsets_dev = torch.empty(volume + 1, dtype=torch.int64, device=dev)
    # pair lists at their upper bound (the pair count stays on the device until somebody needs it on the host)
    in_pairs = torch.empty(n_out * volume, dtype=torch.int32, device=dev)
    out_pairs = torch.empty(n_out * volume, dtype=torch.int32, device=dev)
    with _on(dev):
        st = _stream(dev)
        _lib.check(lib.me_kernel_map_probe_lds(ctypes.byref(sq.grid), _ptr(sq.coords_sorted), _ptr(sq.dir_start), n_out,
                                               ctypes.byref(sl.grid), _ptr(sl.coords_sorted), _ptr(sl.order),
                                               _ptr(sl.dir_start), ctypes.byref(region), _ptr(nbr),
                                               _ptr(k_offsets_dev), _ptr(ws), ws.numel(), st))
        lazy = _LazyOffsets(k_offsets_dev)
        _lib.check(lib.me_kernel_map_compact_ordered(_ptr(nbr), _ptr(sq.order), n_out, volume, _ptr(in_pairs),
                                                     _ptr(out_pairs), _ptr(ws), ws.numel(), st))
    store = {"nbr_out": nbr, "order_out": sq.order, "pos_out": sq.pos_of_row, "order_in": sl.order,
             "pos_in": sl.pos_of_row}
    return KernelMapGPU(volume, n_in, n_out, lazy, k_offsets_dev, in_pairs, out_pairs, store=store, in_map=in_map,
                        out_map=out_map)


def _build_kernel_map(in_map, out_map, region):
    """Iterate OUT coordinates, look up the IN map (src/coordinate_map_cpu.hpp:569-670)."""
    lib = _lib.load()
    dev = in_map.coords.device
    volume = int(lib.me_region_volume(ctypes.byref(region)))
    _check(volume > 0, "invalid kernel region")
    km = _build_kernel_map_lds(in_map, out_map, region, volume)
    if km is not None:
        return km
    n_out, n_in = out_map.n, in_map.n
    nbr = torch.empty((volume, max(n_out, 1)), dtype=torch.int32, device=dev)
    ws = _workspace(lib.me_kernel_map_workspace_bytes(n_out, volume), dev)
    koffs = (ctypes.c_int64 * (volume + 1))()
    k_offsets_dev = torch.empty(volume + 1, dtype=torch.int64, device=dev)
    with _on(dev):
        _lib.check(lib.me_kernel_map_probe(_ptr(in_map.table), in_map.capacity, _ptr(in_map.coords),
                                           _ptr(out_map.coords), n_out, ctypes.byref(region), _ptr(nbr), koffs,
                                           _ptr(k_offsets_dev), _ptr(ws), ws.numel(), _stream(dev)))
        k_offsets = [int(v) for v in koffs]
        n_pairs = k_offsets[-1]
        in_pairs = torch.empty(max(n_pairs, 1), dtype=torch.int32, device=dev)
        out_pairs = torch.empty(max(n_pairs, 1), dtype=torch.int32, device=dev)
        _lib.check(lib.me_kernel_map_compact(_ptr(nbr), n_out, volume, _ptr(in_pairs), _ptr(out_pairs), _ptr(ws),
                                             ws.numel(), _stream(dev)))
    return KernelMapGPU(volume, n_in, n_out, k_offsets, k_offsets_dev, in_pairs[:max(n_pairs, 0)],
                        out_pairs[:max(n_pairs, 0)], store={"nbr_out": nbr}, in_map=in_map, out_map=out_map)


# ------------------------------------------------------------------------------------------------
# CoordinateMapManager (src/coordinate_map_manager.{hpp,cpp,cu})
# ------------------------------------------------------------------------------------------------
_CHARSET = string.digits + string.ascii_uppercase + string.ascii_lowercase


class CoordinateMapManagerGPU_c10:
    """Owner of the coordinate maps and the kernel-map cache of one process / one GPU
    (src/coordinate_map_manager.hpp:130-156, 530-554).  Device memory comes from the torch caching
    allocator, as with the reference's c10 allocator."""

    def __init__(self, algorithm=MinkowskiAlgorithm.DEFAULT, num_threads=0):
        self.algorithm = MinkowskiAlgorithm(algorithm)
        self.num_threads = num_threads
        self._maps = {}          # (tensor_stride tuple, string_id) -> _CoordinateMapGPU
        self._kernel_maps = {}   # kernel_map_key (src/types.hpp:183-192) -> KernelMapGPU
        self._origin_maps = {}
        self._prune_rows = {}
        self._stride_maps = {}
        # Not in the reference: the map-building requests this manager served on a cache miss, in order — strided
        # maps, kernel maps, tile-plan / weight-gradient configurations.  `prefetch(recipe)` replays such a list on a
        # NEW scene right after its coordinates are inserted, so that every host read-back of the build (output sizes,
        # pair counts) happens in one burst before the forward pass instead of draining the launch queue in the
        # middle of it (docs/HISTORY.md 9.8).
        self._recipe = []

    # ---- keys -----------------------------------------------------------------------------------
    @staticmethod
    def _k(key):
        if isinstance(key, CoordinateMapKey):
            return key._hashable()
        ts, sid = key
        return (tuple(int(s) for s in ts), str(sid))

    def exists(self, key):
        return self._k(key) in self._maps

    def _get(self, key):
        k = self._k(key)
        _check(k in self._maps, "coordinate map not found", k)
        return self._maps[k]

    def get_random_string_id(self, tensor_stride, string_id):
        """src/coordinate_map_manager.hpp:473-485"""
        ts = tuple(int(s) for s in tensor_stride)
        while True:
            rnd = "".join(random.choice(_CHARSET) for _ in range(5))
            key = (ts, (string_id + "-" + rnd) if string_id else rnd)
            if key not in self._maps:
                return (list(key[0]), key[1])

    def get_coordinate_map_keys(self, tensor_stride):
        ts = tuple(int(s) for s in tensor_stride)
        return [CoordinateMapKey(list(k[0]), k[1]) for k in self._maps if k[0] == ts]

    # ---- maps -----------------------------------------------------------------------------------
    @_ranged("me:insert_and_map")
    def insert_and_map(self, coordinates, tensor_stride, string_id=""):
        """src/coordinate_map_manager.cpp:349-399 -> (CoordinateMapKey, (unique_map, inverse_map))."""
        _check(isinstance(coordinates, torch.Tensor) and coordinates.dim() == 2, "coordinates must be 2-D")
        _check(coordinates.is_contiguous(), "coordinates must be contiguous")
        _check(coordinates.dtype == torch.int32, "coordinates must be int32")
        _check(coordinates.is_cuda, "coordinates must be on the GPU (the MI355X path has no CPU map)")
        ts = tuple(int(s) for s in tensor_stride)
        _check(coordinates.shape[1] - 1 == len(ts), "The coordinate dimension (coordinate_size - 1):",
               coordinates.shape[1] - 1, " must match the size of tensor stride:", list(ts))
        _lib.preload_device(coordinates.device.index)       # (once per device; never at import)
        key = (ts, str(string_id))
        if key in self._maps:
            k = self.get_random_string_id(ts, string_id)
            key = (tuple(k[0]), k[1])
        if coordinates.data_ptr() % 16 != 0:   # a view into a larger buffer: the kernels use 16-byte loads
            coordinates = coordinates.clone()
        cmap, unique_map, inverse_map = _insert(coordinates, ts)
        self._maps[key] = cmap
        return CoordinateMapKey(list(key[0]), key[1]), (unique_map, inverse_map)

    @_ranged("me:stride")
    def stride(self, in_key, kernel_stride, string_id=""):
        """py_stride: src/coordinate_map_manager.cpp:402-429 -> CoordinateMapKey of the strided map."""
        ik = self._k(in_key)
        _check(ik in self._maps, "coordinate map not found", ik)
        stride = [int(s) for s in kernel_stride]
        _check(len(stride) == len(ik[0]), "stride size mismatch.")
        _check(all(s > 0 for s in stride), "Invalid stride", stride)
        out_ts = tuple(t * s for t, s in zip(ik[0], stride))
        ok = (out_ts, ik[1] if string_id == "" else str(string_id))
        if ok not in self._maps:
            in_map = self._maps[ik]
            lib = _lib.load()
            dev = in_map.coords.device
            ncol = len(out_ts) + 1
            strided = torch.empty((max(in_map.n, 1), ncol), dtype=torch.int32, device=dev)
            ts_arr = (ctypes.c_int32 * len(out_ts))(*out_ts)
            with _on(dev):
                _lib.check(lib.me_coords_stride(_ptr(in_map.coords), in_map.n, ncol, ts_arr, _ptr(strided),
                                                _stream(dev)))
            cmap, _, inverse = _insert(strided[:in_map.n], out_ts)
            self._maps[ok] = cmap
            self._recipe.append(("stride", ik, tuple(stride), str(string_id)))
        return CoordinateMapKey(list(ok[0]), ok[1])

    def stride_map(self, in_key, strided_key):
        """stride_map_th (src/coordinate_map_manager.cpp:977-1034; pybind/extern.hpp:803) -> (in_rows, out_rows), two
        int64 [n_in] tensors: every row of `in_key` and the row of `strided_key` that holds its floored coordinate
        (CoordinateMapCPU::stride_map, src/coordinate_map_cpu.hpp:672-720).  Pairs come in input-row order (the
        reference's order is its hash-table iteration order); an input voxel whose strided coordinate is absent
        from `strided_key` is left out, as in the reference."""
        ik, sk = self._k(in_key), self._k(strided_key)
        _check(ik in self._maps, "coordinate map not found", ik)
        _check(sk in self._maps, "coordinate map not found", sk)
        _check(all(b % a == 0 for a, b in zip(ik[0], sk[0])),
               "The tensor stride of the strided map must be divisible by the tensor stride of the input map.",
               "strided_map_stride:", list(sk[0]), "in_map_stride:", list(ik[0]))
        cached = self._stride_maps.get((ik, sk))
        if cached is not None:
            return cached
        in_map, out_map = self._maps[ik], self._maps[sk]
        lib = _lib.load()
        dev = in_map.coords.device
        ncol = len(sk[0]) + 1
        strided = torch.empty((max(in_map.n, 1), ncol), dtype=torch.int32, device=dev)
        rows = torch.empty(max(in_map.n, 1), dtype=torch.int32, device=dev)
        ts_arr = (ctypes.c_int32 * len(sk[0]))(*sk[0])
        with _on(dev):
            _lib.check(lib.me_coords_stride(_ptr(in_map.coords), in_map.n, ncol, ts_arr, _ptr(strided), _stream(dev)))
            _lib.check(lib.me_coords_find(_ptr(out_map.table), out_map.capacity, _ptr(out_map.coords), ncol,
                                          _ptr(strided), in_map.n, _ptr(rows), _stream(dev)))
        rows = rows[:in_map.n].long()
        in_rows = torch.arange(in_map.n, dtype=torch.int64, device=dev)
        found = rows >= 0
        if not bool(found.all()):
            in_rows, rows = in_rows[found], rows[found]
        self._stride_maps[(ik, sk)] = (in_rows, rows)
        return in_rows, rows

    def _register(self, ts, cmap, string_id=""):
        """insert under (ts, string_id), with a random suffix if that key is taken -> key tuple"""
        key = (tuple(ts), str(string_id))
        if key in self._maps:
            k = self.get_random_string_id(ts, string_id)
            key = (tuple(k[0]), k[1])
        self._maps[key] = cmap
        return key

    def stride_region(self, in_key, kernel_size, kernel_dilation, region_type, out_tensor_stride,
                      expand_coordinates, is_transpose, region_tensor_stride=None):
        """src/coordinate_map_manager.cpp:436-466 -> (CoordinateMapKey, created): the map with
        `out_tensor_stride`; it is reused when it exists and `expand_coordinates` is false, otherwise generated
        from the kernel region around every input coordinate (CoordinateMapCPU::stride_region,
        src/coordinate_map_cpu.hpp:446-487): all of them for a transposed kernel, only those aligned to the out
        tensor stride otherwise.  New rows are ordered by input row, then kernel offset."""
        ik = self._k(in_key)
        _check(ik in self._maps, "coordinate map not found", ik)
        _check(int(region_type) != int(RegionType.CUSTOM), "Not implemented yet.")
        out_ts = tuple(int(t) for t in out_tensor_stride)
        ok = (out_ts, "")
        exists = ok in self._maps
        if exists and not expand_coordinates:
            return CoordinateMapKey(list(ok[0]), ok[1]), False
        in_map = self._maps[ik]
        lib = _lib.load()
        dev = in_map.coords.device
        ncol = len(out_ts) + 1
        # the offsets of a transposed kernel step by the OUT tensor stride (convolution_transpose_cpu.cpp:81-91),
        # those of a regular kernel by the IN tensor stride (convolution_cpu.cpp:88-98)
        rts = out_ts if region_tensor_stride is None else tuple(int(t) for t in region_tensor_stride)
        region = _lib.make_region(ncol, int(region_type), [int(v) for v in kernel_size],
                                  [int(v) for v in kernel_dilation], rts)
        volume = int(lib.me_region_volume(ctypes.byref(region)))
        cand = torch.empty((max(in_map.n * volume, 1), ncol), dtype=torch.int32, device=dev)
        aligned = None if is_transpose else torch.empty(max(in_map.n * volume, 1), dtype=torch.uint8, device=dev)
        ts_arr = (ctypes.c_int32 * len(out_ts))(*out_ts)
        with _on(dev):
            _lib.check(lib.me_coords_expand_region(_ptr(in_map.coords), in_map.n, ncol, ctypes.byref(region),
                                                   None if is_transpose else ts_arr, _ptr(cand), _ptr(aligned),
                                                   _stream(dev)))
        cand = cand[:in_map.n * volume]
        if aligned is not None:
            cand = cand[aligned[:in_map.n * volume].bool()].contiguous()
        cmap, _, _ = _insert(cand, out_ts)
        key = self._register(out_ts, cmap)
        return CoordinateMapKey(list(key[0]), key[1]), True

    def prune(self, in_key, keep):
        """src/coordinate_map_manager.cpp:558-576: new map (tensor stride of `in_key`, string id "pruned") with the
        rows of `in_key` where `keep` is true, in row order."""
        ik = self._k(in_key)
        _check(ik in self._maps, "coordinate map not found", ik)
        in_map = self._maps[ik]
        _check(keep.dim() == 1 and keep.numel() == in_map.n, "Invalid range for pruning")
        rows = torch.nonzero(keep.to(in_map.coords.device).bool(), as_tuple=False).squeeze(1)
        cmap, _, _ = _insert(in_map.coords[rows].contiguous(), ik[0])
        key = self._register(ik[0], cmap, "pruned")
        self._prune_rows[(ik, key)] = rows.to(torch.int32)
        return CoordinateMapKey(list(key[0]), key[1])

    def _pruning_rows(self, in_key, out_key):
        """int32 [n_out]: the row of `in_key` every row of the pruned map `out_key` came from (the one-offset
        kernel map the reference builds for PruningForward, src/pruning_cpu.cpp:82)."""
        rows = self._prune_rows.get((self._k(in_key), self._k(out_key)))
        if rows is None:   # maps that were not produced by prune(): look the coordinates up
            in_map, out_map = self._get(in_key), self._get(out_key)
            lib = _lib.load()
            dev = in_map.coords.device
            rows = torch.empty(max(out_map.n, 1), dtype=torch.int32, device=dev)
            with _on(dev):
                _lib.check(lib.me_coords_find(_ptr(in_map.table), in_map.capacity, _ptr(in_map.coords),
                                              in_map.coords.shape[1], _ptr(out_map.coords), out_map.n, _ptr(rows),
                                              _stream(dev)))
            rows = rows[:out_map.n]
            _check(bool((rows >= 0).all()), "the pruned map is not a subset of the input map")
            self._prune_rows[(self._k(in_key), self._k(out_key))] = rows
        return rows

    def union_map(self, in_keys, out_key):
        """src/coordinate_map_manager.cpp (union_map): the union of the maps `in_keys` (same tensor stride) is
        created under `out_key`; returns one int64 [2, n_i] tensor per input (row 0 = its rows, row 1 = rows of
        the union).  Union rows are ordered by first occurrence over the concatenated inputs."""
        _check(len(in_keys) > 1, "Number of input coordinate keys must be > 1")
        iks = [self._k(k) for k in in_keys]
        for ik in iks:
            _check(ik in self._maps, "coordinate map not found", ik)
            _check(ik[0] == iks[0][0], "Invalid tensor stride", ik[0], iks[0][0])
        maps = [self._maps[ik] for ik in iks]
        allc = torch.cat([m.coords for m in maps], 0).contiguous()
        cmap, _, inverse = _insert(allc, iks[0][0])
        if not out_key.is_key_set():
            key = self._register(iks[0][0], cmap, "union")
            out_key.set_key((list(key[0]), key[1]))
        else:
            self._maps[self._k(out_key)] = cmap
        out, s0 = [], 0
        for m in maps:
            rows = torch.arange(m.n, dtype=torch.int64, device=allc.device)
            out.append(torch.stack((rows, inverse[s0:s0 + m.n])))
            s0 += m.n
        return out

    def get_coordinates(self, key):
        return self._get(key).coords

    # ---- origin map (one row per batch index) --------------------------------------------------
    def origin(self):
        """CoordinateMapKey of the origin map: tensor stride all zeros, one row per batch index, rows in ASCENDING
        batch index — row b of a global pooling belongs to the b-th smallest batch index, whatever order the
        points arrive in (the reference's GPU map sorts the unique batch indices, src/coordinate_map_gpu.cu:765-772;
        its CPU map orders them by hash-table iteration, src/coordinate_map_manager.cpp:470-508)."""
        _check(len(self._maps) > 0, "origin() needs at least one coordinate map")
        any_key = next(iter(self._maps))
        D = len(any_key[0])
        okey = (tuple([0] * D), "")
        if okey not in self._maps:
            # the map with the smallest tensor stride holds every batch index
            cands = [k for k in self._maps if all(t > 0 for t in k[0])]
            base_key = min(cands, key=lambda k: (sum(k[0]), k[1]))
            base = self._maps[base_key]
            batches = torch.unique(base.coords[:, 0])          # sorted
            oc = torch.zeros((batches.numel(), base.coords.shape[1]), dtype=torch.int32, device=base.coords.device)
            oc[:, 0] = batches
            cmap, _, _ = _insert(oc.contiguous(), okey[0])      # unique rows: insertion order = sorted order
            self._maps[okey] = cmap
        return CoordinateMapKey(list(okey[0]), okey[1])

    def origin_map_size(self):
        return self._get(self.origin()).n

    def _origin_rows(self, in_key):
        """int32 [n_in]: origin-map row (output row of a global pooling) of every row of `in_key`;
        the kernel map of src/coordinate_map_cpu.hpp:1067-1100 with one offset, kept as a row table."""
        ik = self._k(in_key)
        self.origin()
        rows = self._origin_maps.get(ik)
        if rows is None:
            in_map, omap = self._get(ik), self._get(self.origin())
            lib = _lib.load()
            dev = in_map.coords.device
            q = torch.zeros_like(in_map.coords)
            q[:, 0] = in_map.coords[:, 0]
            rows = torch.empty(max(in_map.n, 1), dtype=torch.int32, device=dev)
            with _on(dev):
                _lib.check(lib.me_coords_find(_ptr(omap.table), omap.capacity, _ptr(omap.coords), q.shape[1],
                                              _ptr(q), in_map.n, _ptr(rows), _stream(dev)))
            rows = rows[:in_map.n]
            # a map may hold batch indices the origin map (built from the finest map at its first use) does not
            # know: the pooling / broadcast kernels would index row -1.  Checked once per map (one read-back).
            _check(in_map.n == 0 or bool((rows >= 0).all()),
                   "the origin map does not contain every batch index of this coordinate map")
            self._origin_maps[ik] = rows
        return rows

    def origin_map(self, in_key):
        """{0: int32 [2, n_in]} (row 0 = in rows, row 1 = origin rows), the reference's origin_map_th layout."""
        rows = self._origin_rows(in_key)
        return {0: torch.stack((torch.arange(rows.numel(), dtype=torch.int32, device=rows.device), rows))}

    def size(self, key):
        return self._get(key).n

    # ---- kernel maps ----------------------------------------------------------------------------
    @_ranged("me:kernel_map")
    def _kernel_map(self, in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type,
                    offset, is_transpose, is_pool):
        """src/coordinate_map_manager.cpp:655-823 -> KernelMapGPU (cached)."""
        _check(int(region_type) != int(RegionType.CUSTOM), "Not implemented yet.")
        ks = tuple(int(v) for v in kernel_size)
        st = tuple(int(v) for v in kernel_stride)
        dl = tuple(int(v) for v in kernel_dilation)
        _check(len(ks) == len(st) == len(dl), "kernel size mismatch")
        ik, ok = self._k(in_key), self._k(out_key)
        key = (ik, ok, ks, st, dl, int(region_type), bool(is_transpose), bool(is_pool))
        km = self._kernel_maps.get(key)
        if km is not None:
            return km
        in_map, out_map = self._get(ik), self._get(ok)
        _check(len(ks) + 1 == in_map.coords.shape[1], "kernel size mismatch")
        if ik == ok and all(k == 1 for k in ks):
            # a 1x1 kernel on one map: every row is paired with itself (volume-1 shortcut of
            # src/coordinate_map_cpu.hpp:605-616) — no probe, no host synchronisation
            n = in_map.n
            rows = torch.arange(n, dtype=torch.int32, device=in_map.coords.device)
            km = KernelMapGPU(1, n, n, [0, n], torch.tensor([0, n], dtype=torch.int64, device=rows.device), rows, rows,
                              store={"nbr_out": rows.view(1, n) if n else rows.new_empty((1, 1))},
                              in_map=in_map, out_map=out_map)
            km.one_pair_sides = 3
            self._kernel_maps[key] = km
            return km
        if not is_transpose:
            # (pooling with stride == kernel uses the same generic path: the result is identical)
            region = _lib.make_region(len(ks) + 1, int(region_type), ks, dl, in_map.tensor_stride)
            km = _build_kernel_map(in_map, out_map, region)
            km.one_pair_sides = _one_pair_sides(ks, dl, int(region_type), in_map.tensor_stride, out_map.tensor_stride)
        else:
            swapped_key = (ok, ik, ks, st, dl, int(region_type), False, bool(is_pool))
            fwd = self._kernel_maps.get(swapped_key)
            if fwd is None:
                # out -> in map with the (finer) out tensor stride, then swap
                region = _lib.make_region(len(ks) + 1, int(region_type), ks, dl, out_map.tensor_stride)
                fwd = _build_kernel_map(out_map, in_map, region)
                fwd.one_pair_sides = _one_pair_sides(ks, dl, int(region_type), out_map.tensor_stride, in_map.tensor_stride)
            km = fwd.swapped()
        self._kernel_maps[key] = km
        km._recipe, km._recipe_key = self._recipe, key
        self._recipe.append(("kernel_map", key))
        return km

    def device_tensors(self):
        """Every device tensor this manager holds (coordinate maps, hash tables, spatial indices, kernel maps, tile
        plans, cached launch configurations)."""
        seen, out = set(), []

        def walk(o, depth=0):
            if isinstance(o, torch.Tensor):
                if o.is_cuda and id(o) not in seen:
                    seen.add(id(o))
                    out.append(o)
                return
            if id(o) in seen or depth > 6:
                return
            if isinstance(o, (list, tuple, set)):
                seen.add(id(o))
                for v in o:
                    walk(v, depth + 1)
            elif isinstance(o, dict):
                seen.add(id(o))
                for v in o.values():
                    walk(v, depth + 1)
            elif isinstance(o, (_CoordinateMapGPU, KernelMapGPU, _SpatialIndex, _LazyOffsets)):
                seen.add(id(o))
                names = getattr(type(o), "__slots__", None) or list(vars(o))
                for name in names:
                    if name not in ("_recipe", "in_map", "out_map"):
                        walk(getattr(o, name, None), depth + 1)
        for store in (self._maps, self._kernel_maps, self._origin_maps, self._prune_rows, self._stride_maps):
            walk(store)
        return out

    def record_stream(self, stream):
        """Tell the caching allocator that `stream` uses this manager's buffers (torch.Tensor.record_stream on each):
        needed when the maps were built on a side stream — e.g. the next scene's maps during the current step's
        backward pass, docs/HISTORY.md 9.8 — and are consumed on another one."""
        for t in self.device_tensors():
            t.record_stream(stream)

    def recipe(self):
        """The build requests served so far (a list of plain tuples; see __init__)."""
        return list(self._recipe)

    def prefetch(self, recipe):
        """Replay the build requests of another scene's manager on this one: strided coordinate maps, kernel maps,
        tile plans and weight-gradient launch configurations are built now (results are cached under the same keys
        the layers will ask for).  Requests that do not apply to this manager are skipped.  Returns the number of
        requests replayed."""
        done = 0
        for op in recipe:
            if True:   # (requests that do not apply are skipped by the key checks below; real errors propagate)
                if op[0] == "stride":
                    _, ik, stride, sid = op
                    if ik in self._maps:
                        self.stride(CoordinateMapKey(list(ik[0]), ik[1]), list(stride), sid)
                        done += 1
                elif op[0] == "kernel_map":
                    ik, ok, ks, st, dl, rt, tr, pool = op[1]
                    if ik in self._maps and ok in self._maps:
                        self._kernel_map(CoordinateMapKey(list(ik[0]), ik[1]), CoordinateMapKey(list(ok[0]), ok[1]), ks,
                                         st, dl, RegionType(rt), None, tr, pool)
                        done += 1
                elif op[0] == "conv_cfg":
                    _, key, target, c_src, c_dst, bf16 = op
                    km = self._kernel_maps.get(key)
                    if km is not None:
                        n_tgt = km.n_out if target == "out" else km.n_in
                        if bf16 and c_src == 8 and _lib.load().me_conv_stem_use_bf16(n_tgt, km.volume, c_src, c_dst):
                            km.table_pos(target)       # (the stacked-offset kernel reads the neighbour table: no plan)
                        elif not (bf16 and _halo_launch_cfg(km, target, n_tgt, c_src, c_dst, count=False) is not None):
                            _conv_launch_cfg(km, target, n_tgt, c_src, c_dst, bf16)
                        done += 1
                elif op[0] == "rowwise_cfg":       # (one pair per row: the launch reads the pair lists, no plan)
                    _, key, target, c_src, c_dst = op
                    km = self._kernel_maps.get(key)
                    if km is not None:
                        _rowwise_cfg(km, target, km.n_out if target == "out" else km.n_in, c_src, c_dst)
                        done += 1
                elif op[0] == "wgrad_cfg":
                    _, key, c_in, c_out, bf16 = op
                    km = self._kernel_maps.get(key)
                    if km is not None:
                        _wgrad_launch_cfg(km, c_in, c_out, bf16)
                        done += 1
        return done

    def kernel_map(self, in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                   is_transpose, is_pool):
        """kernel_map_th: dict {k: int32 [2, n_k]} (src/coordinate_map_manager.cpp:1358-1387)."""
        return self._kernel_map(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type,
                                offset, is_transpose, is_pool).to_dict()

    def print_coordinate_map(self, key):
        """one map's line of __repr__ (manager_type::to_string(key), pybind/extern.hpp:777-779,
        src/coordinate_map_manager.hpp:383-387)"""
        k = self._k(key)
        m = self._get(key)
        return f"{list(k[0])}{':' + k[1] if k[1] else ''} : CoordinateMapGPU:{m.n}x{m.coords.shape[1]}"

    def __repr__(self):
        s = f"{self.__class__.__name__}(\n"
        for k, m in self._maps.items():
            s += f"\t{list(k[0])}{':' + k[1] if k[1] else ''}:\tCoordinateMapGPU:{m.n}x{m.coords.shape[1]}\n"
        for k, km in self._kernel_maps.items():
            s += f"\t{list(k[0][0])}->{list(k[1][0])}:\tgpu_kernel_map: number of unique maps:{km.volume}, pairs:{km.n_pairs}\n"
        return s + f"\talgorithm={self.algorithm.name}\n)"


CoordinateMapManagerGPU_default = CoordinateMapManagerGPU_c10


# ------------------------------------------------------------------------------------------------
# convolution operators (src/convolution_gpu.cu:45-244, src/convolution_transpose_gpu.cu)
# ------------------------------------------------------------------------------------------------
# bf16 tile kernel with batch fusion (me_conv_target_bf16_fused): "auto" = maps with fewer than 24 pairs per (tile,
# offset) item, whose batches mostly hold ONE 16-row group (config 5 in bf16: forward 170 -> 144 us, dgrad 137 -> 98,
# 973 -> 1156 Mpoints/s; sparse config 2: +8 %; dense layers lose 20 - 35 % on it and MinkUNet34C as a whole 3 %:
# profiles/r02_bench_bf16_batch_fusion.log); "1" / "0" force
_BF16_FUSE = os.environ.get("ME_AMD_BF16_FUSE", "auto")
_BF16_FUSE_MAX_PAIRS_PER_ITEM = 24.0
# fp32-MFMA tile kernel with multi-offset batches (me_conv_target_f32_fused): the same density rule; "0" disables
_F32_FUSE = os.environ.get("ME_AMD_F32_FUSE", "1") != "0"
_BF16_GATHER = os.environ.get("ME_AMD_BF16_GATHER", "0") != "0"   # bf16: output-stationary kernel (opt-in: measured slower than the tile-plan kernel)
# fp32 features: forward / dgrad on the bf16 matrix pipe with exactly split operands (csrc/conv_f32x3.hip; fp32-grade
# results, docs/HISTORY.md 9.7).  "auto": where it measured faster than the fp32-MFMA kernel k_conv_tile_f32 — layers with
# c_src * c_dst >= 8192 (64 -> 128 and wider: 1.1 - 1.6x on every map density tried; below that the per-batch costs of
# three operand planes outweigh the cheaper MFMAs, profiles/r02_tune_split_policy.log); "1": wherever supported
# (c_src % 8 == 0); "0": never.
_F32_SPLIT = {"0": False, "1": True}.get(os.environ.get("ME_AMD_F32_SPLIT", "auto"), "auto")


def _use_split(lib, c_src, c_dst):
    if _F32_SPLIT is False or not lib.me_conv_f32x3_supported(c_src, c_dst):
        return False
    return True if _F32_SPLIT is True else c_src * c_dst >= 8192


_ALGO = os.environ.get("ME_AMD_CONV_ALGO", "mfma")  # "naive" = VALU/atomics cross-check kernels
# Weight gradient on a side stream, concurrent with the input gradient of the same layer (_conv_backward).  OPT-IN
# ("1"): measured in round 3 (profiles/r03_wgrad_side_stream.md) it does not pay — both gradient launches already
# occupy every CU (one to three workgroups per CU by LDS), so they time-share instead of overlapping: MinkUNet34C bf16
# as a hipGraph 14.74 ms without, 14.62 - 14.86 ms with (any row threshold); the headline layer loses 3.5 % (0.342 vs
# 0.330 ms), MinkUNet34C fp32 1 ms; only config 5 gains 3 %.  ME_AMD_WGRAD_STREAM_MAX_ROWS bounds it to small maps.
_WGRAD_STREAM = os.environ.get("ME_AMD_WGRAD_STREAM", "0") != "0"
_WGRAD_STREAM_MAX_ROWS = int(os.environ.get("ME_AMD_WGRAD_STREAM_MAX_ROWS", "100000000"))
_SIDE_STREAMS = {}


def _side_stream(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _SIDE_STREAMS.get(idx)
    if s is None:
        s = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return s

_WGRAD_TUNING = False  # set by the tuning scripts, which flip the wgrad debug switches between calls: the
                       # workspace size is then re-queried on every call instead of cached per kernel map
_TILE_ROWS = int(os.environ.get("ME_AMD_TILE_ROWS", "0"))        # 0 = me_conv_plan_config (tuning overrides)
_BATCH_GROUPS = int(os.environ.get("ME_AMD_BATCH_GROUPS", "0"))
_SPATIAL_TILES = os.environ.get("ME_AMD_SPATIAL_TILES", "0") != "0"  # tiles of Z-order-sorted target rows (off: no gain measured while the gather is latency-hidden)


def _check_feat(name, t):
    _check(t.is_contiguous(), name, "must be contiguous")
    _check(t.is_cuda, name, "must be CUDA (ROCm) — the MI355X path has no CPU implementation")
    _check(t.dtype in (torch.float32, torch.bfloat16, torch.float64), name,
           "must be float32, bfloat16 or float64, got", t.dtype)


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (torch's current stream).
    bench.py switches it on to measure the average duration of each hot kernel inside the timed
    region; it is off (None) otherwise."""

    def __init__(self):
        self.events = {}
        self.flops = {}
        self.native = {}     # name -> [ms] measured inside the native host layer

    def record(self, name, device):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(device))
        return ev

    def add(self, name, start, end, flops=0.0):
        self.events.setdefault(name, []).append((start, end))
        self.flops[name] = self.flops.get(name, 0.0) + flops

    def _native_records(self):
        """launches timed by the native host layer (csrc_host/ops.cpp ScopedTimer) since the last call"""
        from . import host as _host
        nm = _host.native_module()
        if nm is None:
            return
        for name, ms, flops in nm.timing_records(True):
            self.native.setdefault(name, []).append(ms)
            self.flops[name] = self.flops.get(name, 0.0) + flops

    def summary(self):
        """{name: (launches, mean ms)} — call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(s.elapsed_time(e) for s, e in v) / len(v)) for k, v in self.events.items()}

    def totals(self):
        """{name: (launches, total ms, total algorithmic flops)} — call after torch.cuda.synchronize()."""
        self._native_records()
        out = {k: (len(v), sum(s.elapsed_time(e) for s, e in v), self.flops.get(k, 0.0))
               for k, v in self.events.items()}
        for k, v in self.native.items():
            n0, t0, _ = out.get(k, (0, 0.0, 0.0))
            out[k] = (n0 + len(v), t0 + sum(v), self.flops.get(k, 0.0))
        return out


KERNEL_TIMER = None  # set to a KernelTimer() to time launches


def _timed(name, device, launch, flops=0.0):
    if _ROCTX is not None:
        with _roctx("me:" + name):
            return _timed_inner(name, device, launch, flops)
    return _timed_inner(name, device, launch, flops)


def _timed_inner(name, device, launch, flops=0.0):
    if KERNEL_TIMER is None:
        return launch()
    s = KERNEL_TIMER.record(name, device)
    r = launch()
    KERNEL_TIMER.add(name, s, KERNEL_TIMER.record(name, device), flops)
    return r


def plan_config(n_tgt, volume, n_pairs, c_src, c_dst, bf16=False, split=False, with_split_k=False):
    """(tile_rows, batch_groups) of the tile plan for a (target rows, channels) problem; with_split_k: + the number of
    offset groups of a split-K launch (bf16 features on small maps, me_conv_plan_config_bf16_ex; 1 = not split)."""
    lib = _lib.load()
    t, g, sk = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(1)
    if bf16:
        _lib.check(lib.me_conv_plan_config_bf16_ex(n_tgt, volume, n_pairs, c_src, c_dst, ctypes.byref(t), ctypes.byref(g),
                                                   ctypes.byref(sk)))
        if _TILE_ROWS or _BATCH_GROUPS:      # geometry overrides (tests / tuning): the unsplit launch on them
            _lib.check(lib.me_conv_plan_config_bf16(n_tgt, volume, n_pairs, c_src, c_dst, ctypes.byref(t),
                                                    ctypes.byref(g)))
            sk = ctypes.c_int32(1)
    else:
        fn = lib.me_conv_plan_config_f32x3 if split else lib.me_conv_plan_config
        _lib.check(fn(n_tgt, volume, n_pairs, c_src, c_dst, ctypes.byref(t), ctypes.byref(g)))
    out = (_TILE_ROWS or int(t.value), _BATCH_GROUPS or int(g.value))
    return out + (int(sk.value),) if with_split_k else out


def _one_pair_sides(ks, dl, region_type, fine_ts, coarse_ts):
    """Sides of a hyper-cube map (looked-up map at `fine_ts`, iterated map at `coarse_ts`) on which a row has at most ONE
    pair by construction (csrc_host/manager.cpp one_pair_sides_of): a single-offset kernel -> both (3); windows that tile
    space without overlap (coarse stride = kernel_size x fine stride, no dilation) -> the fine ("in") side (1)."""
    if all(k == 1 for k in ks):
        return 3
    tiling = region_type == 0 and len(fine_ts) == len(ks) == len(coarse_ts) and \
        all(d == 1 and c == f * k for k, d, f, c in zip(ks, dl, fine_ts, coarse_ts))
    return 1 if tiling else 0


_ROWWISE = os.environ.get("ME_AMD_ROWWISE", "1") != "0"   # 0: one-pair-per-row sides stay on the tile-plan kernels
# a K = 1 FORWARD launch whose batch-norm statistics are wanted takes the row-wise kernel (no statistics epilogue: the batch
# norm reads the output once more) only on maps of at least this many rows (csrc_host/host.hpp Policy, measured per layer)
_ROWWISE_MIN_ROWS_WITH_STATS = int(os.environ.get("ME_AMD_ROWWISE_STATS_ROWS", "100000"))


def _rowwise_cfg(km, target, n_tgt, c_src, c_dst):
    """(src_rows, tgt_rows, elems) when the launch side has exactly one pair per target row and libme_amd takes the
    shape (csrc/conv_rowwise.hip: K = 1 layers, the fine side of kernel_size == stride maps), else None"""
    if not _ROWWISE or not (km.one_pair_sides & (1 if target == "in" else 2)):
        return None
    ck = ("rowwise", target, c_src, c_dst)
    cfg = km._launch_cache.get(ck)
    if cfg is None:
        lib = _lib.load()
        ok = bool(lib.me_conv_rowwise_supported_bf16(km.volume, c_src, c_dst)) and km.n_pairs == n_tgt
        src_rows = km.in_pairs_buf if target == "out" else km.out_pairs_buf
        tgt_rows = km.out_pairs_buf if target == "out" else km.in_pairs_buf
        cfg = km._launch_cache[ck] = (src_rows, tgt_rows, int(lib.me_conv_packed_weight_elems_bf16(km.volume, c_src, c_dst))) \
            if ok else False
        if ok and km._recipe is not None:
            km._recipe.append(("rowwise_cfg", km._recipe_key, target, c_src, c_dst))
    return cfg or None


def _conv_launch_cfg(km, target, n_tgt, c_src, c_dst, bf16):
    """Launch geometry of a (kernel map side, channel shape, dtype): tile plan + device addresses, computed once per
    kernel map -> (split, cfg)."""
    lib = _lib.load()
    volume = km.volume
    split = (not bf16) and _use_split(lib, c_src, c_dst)
    ck = (target, c_src, c_dst, bf16, split, _TILE_ROWS, _BATCH_GROUPS, _SPATIAL_TILES)
    cfg = km._launch_cache.get(ck)
    if cfg is None:
        tile_rows, batch_groups, split_k = plan_config(n_tgt, km.volume, km.n_pairs, c_src, c_dst, bf16, split,
                                                       with_split_k=True)
        n_src = km.n_in if target == "out" else km.n_out
        tile_order = km._tile_order(target, matrix_bound=split, src_bytes=n_src * c_src * 2 if bf16 else 0)
        plan_src, plan_dst, batch_desc, tile_bptr, _ = km.plan(target, tile_rows, batch_groups, tile_order)
        elems = int((lib.me_conv_packed_weight_elems_bf16 if bf16 else
                     (lib.me_conv_packed_weight_elems_f32x3 if split else lib.me_conv_packed_weight_elems))(
            volume, c_src, c_dst))
        order = km.order(target, tile_order)
        # batch fusion (bf16 tile kernel): on maps whose (tile, offset) items mostly hold one or two 16-row groups
        n_tiles = -(-n_tgt // tile_rows)
        per_item = (km.n_pairs - min(km.n_in, km.n_out)) / max(1, (volume - 1) * n_tiles) if volume > 1 else 1e9
        fuse = {"1": True, "0": False}.get(_BF16_FUSE, per_item < _BF16_FUSE_MAX_PAIRS_PER_ITEM)
        if split_k > 1:
            fuse = False
        cfg = (tile_rows, batch_groups, plan_src, plan_dst, batch_desc, tile_bptr, order, elems,
               _ptr(plan_src), _ptr(plan_dst), _ptr(batch_desc), _ptr(tile_bptr), _ptr(order), fuse, split_k)
        km._launch_cache[ck] = cfg
        if km._recipe is not None:
            km._recipe.append(("conv_cfg", km._recipe_key, target, c_src, c_dst, bool(bf16)))
    return split, cfg


# gradient destinations (distributed.GradientArena; csrc_host/ops.cpp twin): parameter storage address -> the fp32 buffer
# its gradient is written into (a slice of one flat all-reduce buffer)
_GRAD_DEST = {}     # address -> [dest, armed]: an entry is used ONCE per arming (see csrc_host/ops.cpp)


def set_grad_destination(param, dest):
    if dest is None:
        _GRAD_DEST.pop(param.data_ptr(), None)
    else:
        _GRAD_DEST[param.data_ptr()] = [dest, False]


def arm_grad_destinations(ptrs=None):
    """ptrs=None: every registered destination; an iterable of parameter addresses: only those (one arena's own —
    another arena's slices may hold live gradients: ADVICE r5)"""
    if ptrs is None:
        for e in _GRAD_DEST.values():
            e[1] = True
        return
    for a in ptrs:
        e = _GRAD_DEST.get(a)
        if e is not None:
            e[1] = True


def drop_grad_destinations(ptrs):
    for a in ptrs:
        _GRAD_DEST.pop(a, None)


def clear_grad_destinations():
    _GRAD_DEST.clear()


def _grad_destination(param, shape):
    """a FRESH alias of the registered buffer in `shape` (autograd takes a gradient without a copy only when nobody else
    holds the tensor object), or None"""
    e = _GRAD_DEST.get(param.data_ptr()) if param is not None else None
    if e is None or not e[1]:
        return None
    d = e[0]
    n = 1
    for v in shape:
        n *= int(v)
    if d.dtype != torch.float32 or d.numel() != n or d.device != param.device:
        return None
    e[1] = False
    return d.view(tuple(shape))


def _halo_launch_cfg(km, target, n_tgt, c_src, c_dst, count=True):
    """Halo plan (csrc/conv_halo.hip) of a launch side when libme_amd's policy (me_conv_halo_use_bf16: ME_AMD_HALO, shape,
    density) sends it to the output-stationary kernel, else None -> the tile-plan kernels.  Tiles are runs of target
    rows in the Z-order of the target's coordinate map (compact at every length: small halos).
    The plan is built at the me_conv_halo_min_uses()-th LAUNCH on this side (count=False: a recipe replay asks, no launch
    follows) — it costs more than one launch saves, so a scene that is used once never builds it."""
    lib = _lib.load()
    if not lib.me_conv_halo_use_bf16(n_tgt, km.volume, km.n_pairs, c_src, c_dst):
        return None
    uk = ("halo_uses", target, c_src, c_dst)
    uses = km._launch_cache.get(uk, 0) + (1 if count else 0)
    km._launch_cache[uk] = uses
    if uses < int(lib.me_conv_halo_min_uses()):
        return None
    t, cap = ctypes.c_int32(0), ctypes.c_int32(0)
    if not lib.me_conv_halo_config_bf16(n_tgt, km.volume, km.n_pairs, c_src, c_dst, ctypes.byref(t), ctypes.byref(cap)):
        return None
    tile_rows, s_cap = int(t.value), int(cap.value)
    name = km._name("halo", target) + f"_{tile_rows}_{s_cap}"
    if name not in km._store:
        dev = km.device
        tbl, native = km.table_pos(target)
        cmap = km.out_map if target == "out" else km.in_map
        out_order = cmap.zorder() if cmap is not None else None
        if out_order is None or out_order.numel() != n_tgt:
            km._store[name] = None
            return None
        # a position-space table (LDS-bucketed map build) is read through pos_of_row
        col_order = out_order if native is None else km._store[km._name("pos", target)][out_order.long()].contiguous()
        # the halo slots in the Z-order of the SOURCE map: rows that are gathered together sit in neighbouring slots
        smap = km.in_map if target == "out" else km.out_map
        src_order = smap.zorder() if smap is not None else None
        src_pos = smap.zorder_inv() if smap is not None else None
        tiles = int(lib.me_halo_plan_num_tiles(n_tgt, tile_rows))
        halo_cnt = torch.empty(tiles, dtype=torch.int32, device=dev)
        halo_rows = torch.empty(tiles * s_cap, dtype=torch.int32, device=dev)
        lidx = torch.empty(tiles * km.volume * tile_rows, dtype=torch.int16, device=dev)
        kmask = torch.empty(tiles * km.volume, dtype=torch.int32, device=dev)
        with _on(dev), _roctx("me:halo_plan"):
            _lib.check(lib.me_halo_plan_build(_ptr(tbl), _ptr(col_order), _ptr(src_pos), _ptr(src_order), n_tgt, km.volume, tile_rows, s_cap,
                                              _ptr(halo_cnt), _ptr(halo_rows), _ptr(lidx), _ptr(kmask), _stream(dev)))
        km._store[name] = (tile_rows, s_cap, halo_cnt, halo_rows, lidx, kmask, tbl, col_order, out_order)
        if km._recipe is not None and int(lib.me_conv_halo_min_uses()) <= 1:
            # (forced mode only: under the policy the plan is a reward for reuse, not a request the next scene inherits)
            km._recipe.append(("conv_cfg", km._recipe_key, target, c_src, c_dst, True))
    return km._store[name]


def _wgrad_launch_cfg(km, c_in, c_out, bf16):
    lib = _lib.load()
    ck = ("wgrad", c_in, c_out, bf16)
    cfg = km._launch_cache.get(ck)
    if cfg is None:
        volume = km.volume
        koffs = (ctypes.c_int64 * (volume + 1))(*km.k_offsets)
        wsb = int((lib.me_conv_wgrad_workspace_bytes_bf16 if bf16 else lib.me_conv_wgrad_workspace_bytes)(
            koffs, volume, c_in, c_out))
        cfg = (koffs, wsb, _ptr(km.in_pairs_buf), _ptr(km.out_pairs_buf), _ptr(km.k_offsets_dev))
        km._launch_cache[ck] = cfg
        if km._recipe is not None:
            km._recipe.append(("wgrad_cfg", km._recipe_key, c_in, c_out, bool(bf16)))
    return cfg


# ------------------------------------------------------------------------------------------------
# packed weights: cached per (weight tensor, direction), every stale image repacked by ONE launch
# ------------------------------------------------------------------------------------------------
_PACK_CACHE = os.environ.get("ME_AMD_PACK_CACHE", "1") != "0"   # "0": pack per launch (round-2 behaviour)


class _PackEntry:
    __slots__ = ("ref", "ptr", "shape", "dtype", "mode", "transposed", "version", "epoch", "packed", "job")


class _WeightPacker:
    """Packed MFMA images of the convolution weights of one device (csrc/pack.hip).

    The images depend on the weights alone, and the weights change at the optimizer step — not between the forward
    and the backward launch of a layer, and not between layers.  An image is therefore kept next to its weight
    tensor, keyed by (storage address, direction, kernel family) and validated by the tensor's VERSION COUNTER (every
    in-place update — optimizer step, copy_, load_state_dict — bumps it; a new storage is a new key).  On a miss ALL
    images whose weight has moved on are repacked together by me_conv_pack_weights_multi: in a training loop that is
    the first convolution after the optimizer step — one launch per step instead of two per layer (MinkUNet34C: 126),
    with the job table cached on the device while the set of stale images repeats.
    The version counter does NOT see writes through `.data` (`p.data.add_(1)` leaves `p._version` alone): code that
    updates weights that way calls `invalidate_packed_weights()`; the package does so itself after every
    `torch.optim` step (global step post-hook, `__init__.py`).  An image is valid for (version, epoch).
    Entries hold only a weak reference to the weight: a temporary weight tensor (tests, functional calls) is packed
    per call as before.  Not for concurrent use of one layer from two streams (the image buffer is reused in place;
    stream order protects the previous step's launches)."""

    def __init__(self, dev):
        self.dev = dev
        self.entries = {}
        self.table_key, self.table = None, None     # device copy of the last job table

    @staticmethod
    def _alive(ent):
        t = ent.ref()
        return t is not None and t.data_ptr() == ent.ptr

    def get(self, kernel, mode, transposed, c_src, c_dst, elems):
        key = (kernel.data_ptr(), mode, bool(transposed))
        ent = self.entries.get(key)
        if ent is not None and self._alive(ent) and ent.shape == tuple(kernel.shape) and ent.dtype == kernel.dtype:
            if ent.version == kernel._version and ent.epoch == _PACK_EPOCH[0]:
                return ent.packed
        else:
            import weakref
            if len(self.entries) > 4096:
                self.entries = {k: e for k, e in self.entries.items() if self._alive(e)}
            ent = _PackEntry()
            # (a full view of a parameter — a 1x1 convolution passes kernel.unsqueeze(0) — is anchored at its base:
            # the view object dies with the autograd node, the parameter lives on; they share the version counter)
            base = kernel._base
            anchor = base if (base is not None and base.data_ptr() == kernel.data_ptr() and
                              base.numel() == kernel.numel()) else kernel
            ent.ref, ent.ptr, ent.shape, ent.dtype = weakref.ref(anchor), kernel.data_ptr(), tuple(kernel.shape), kernel.dtype
            ent.mode, ent.transposed, ent.version, ent.epoch = mode, bool(transposed), None, -1
            ent.packed = torch.empty(elems, dtype=torch.bfloat16, device=self.dev)
            job = _lib.MePackJob()
            job.w, job.wp, job.volume = ent.ptr, ent.packed.data_ptr(), int(kernel.shape[0])
            job.c_src, job.c_dst, job.transposed = int(c_src), int(c_dst), 1 if transposed else 0
            job.w_is_f32, job.mode = 1 if kernel.dtype == torch.float32 else 0, mode
            _lib.check(_lib.load().me_conv_pack_job_init(ctypes.byref(job)))
            ent.job = job
            self.entries[key] = ent
        self._repack_stale(ent, kernel)
        return ent.packed

    def _pack_one(self, ent, version):
        """one image through the single-layer pack entry points, which take the geometry as arguments: no job table
        to upload (an upload is not capturable into a hipGraph either)"""
        lib = _lib.load()
        job = ent.job
        with _on(self.dev):
            if job.mode == _lib.ME_PACK_BF16:
                _lib.check(lib.me_conv_pack_weights_bf16(job.w, job.w_is_f32, job.volume, job.c_src, job.c_dst,
                                                         job.transposed, job.wp, _stream(self.dev)))
            else:
                _lib.check(lib.me_conv_pack_weights_f32x3(job.w, job.volume, job.c_src, job.c_dst, job.transposed,
                                                          job.wp, _stream(self.dev)))
        ent.version, ent.epoch = version, _PACK_EPOCH[0]

    def _repack_stale(self, wanted, kernel):
        # a NEW entry (first use of a weight tensor, or a temporary one — a padded / reshaped copy made per step) is
        # packed on its own, so that the set of established images — and with it the cached job table — repeats from
        # step to step
        epoch = _PACK_EPOCH[0]
        if wanted.version is None:
            self._pack_one(wanted, kernel._version)
        stale = []
        for ent in self.entries.values():
            if ent.version is None:
                continue
            t = kernel if ent is wanted else ent.ref()
            if t is not None and t.data_ptr() == ent.ptr and (t._version != ent.version or ent.epoch != epoch):
                stale.append((ent, t._version))
        if not stale:
            return
        if len(stale) > 1024:       # (the wanted image is never the one that is dropped)
            keep = stale[-1024:]
            if not any(e is wanted for e, _ in keep):
                keep[0:1] = [s for s in stale if s[0] is wanted][:1] or keep[0:1]
            stale = keep
        lib = _lib.load()
        if len(stale) == 1:
            self._pack_one(stale[0][0], stale[0][1])
            return
        key = tuple(id(e) for e, _ in stale)
        if key != self.table_key:
            jobs = (_lib.MePackJob * len(stale))(*[e.job for e, _ in stale])
            prefix = [0]
            for e, _ in stale:
                prefix.append(prefix[-1] + int(e.job.threads))
            jb = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(self.dev)
            pf = torch.tensor(prefix, dtype=torch.int64).to(self.dev)
            self.table_key, self.table = key, (jb, pf, prefix[-1], [e for e, _ in stale])   # (entries kept alive)
        jb, pf, total, _ = self.table
        with _on(self.dev):
            _lib.check(lib.me_conv_pack_weights_multi(jb.data_ptr(), len(stale), pf.data_ptr(), total,
                                                      _stream(self.dev)))
        for e, v in stale:
            e.version, e.epoch = v, epoch


_PACKERS = {}
_PACK_EPOCH = [0]


def invalidate_packed_weights(ptrs=None, ranges=None):
    """Every cached weight image is repacked at its next use.  For weight updates the tensor version counter cannot
    see — writes through `p.data` (Apex / DeepSpeed-style optimizers, EMA, clipping); in-place operations on the
    parameter itself (`with torch.no_grad(): p.add_(...)`, `copy_`, torch.optim) are seen without it.
    ptrs: a set of storage addresses (`p.data_ptr()`) / ranges: sorted [(begin, end)) byte ranges of parameter storage —
    only the images of weights that live THERE go stale (the optimizer-step hook passes its own parameters: frozen /
    teacher networks keep their images).  -> (images matched, images cached)."""
    if ptrs is None and ranges is None:
        _PACK_EPOCH[0] += 1
        return (0, 0)
    matched = total = 0
    for pk in _PACKERS.values():
        for e in pk.entries.values():
            total += 1
            hit = ptrs is not None and e.ptr in ptrs
            if not hit and ranges:
                import bisect
                i = bisect.bisect_right(ranges, (e.ptr, float("inf"))) - 1
                hit = i >= 0 and ranges[i][0] <= e.ptr < ranges[i][1]
            if hit:
                e.epoch = -1
                matched += 1
    return (matched, total)


def _packed_weights(kernel, mode, transposed, c_src, c_dst, elems):
    """bf16 / split image of `kernel` for a launch with (c_src, c_dst) channels, from the device's _WeightPacker"""
    dev = kernel.device
    pk = _PACKERS.get(dev.index)
    if pk is None:
        pk = _PACKERS[dev.index] = _WeightPacker(dev)
    return pk.get(kernel, mode, transposed, c_src, c_dst, elems)


def _conv_target(src_feat, kernel, km, target, n_tgt, name="conv_target", transposed=False):
    """dst[t] = sum over plan entries of src[s] @ W[k].
    transposed=False: W[k] = kernel[k] ([c_src, c_dst]);  transposed=True (dgrad): W[k] = kernel[k]^T."""
    lib = _lib.load()
    dev = src_feat.device
    volume = int(kernel.shape[0])
    c_src, c_dst = (int(kernel.shape[2]), int(kernel.shape[1])) if transposed else \
        (int(kernel.shape[1]), int(kernel.shape[2]))
    bf16 = src_feat.dtype == torch.bfloat16
    out = torch.empty((n_tgt, c_dst), dtype=src_feat.dtype, device=dev)
    if n_tgt == 0:
        return out
    if src_feat.dtype == torch.float64:
        # float64 features (the reference's AT_DISPATCH_FLOATING_TYPES double instantiation, src/convolution_gpu.cu:
        # 137-155; its gradcheck dtype): csrc/f64.hip — plain double FMAs on the neighbour table, no plan, no packing
        _check(kernel.dtype == torch.float64, "float64 features need a float64 kernel, got", kernel.dtype)
        tbl = km.table(target)
        with _on(dev):
            _timed(name, dev, lambda: _lib.check(lib.me_conv_target_f64(
                src_feat.data_ptr(), src_feat.shape[0], c_src, kernel.contiguous().data_ptr(), 1 if transposed else 0,
                volume, c_dst, _ptr(tbl), out.data_ptr(), n_tgt, _stream(dev))),
                flops=2.0 * km.n_pairs * c_src * c_dst if KERNEL_TIMER else 0.0)
        return out
    if bf16 and _BF16_GATHER and lib.me_conv_gather_supported_bf16(c_src, c_dst):
        # output-stationary kernel on the neighbour table itself (no tile plan): csrc/conv_bf16.hip k_conv_gather_bf16
        _check(kernel.dtype in (torch.float32, torch.bfloat16), "kernel must be float32 or bfloat16")
        ck = ("gather", target, c_src, c_dst)
        cfg = km._launch_cache.get(ck)
        if cfg is None:
            tbl, order = km.table_pos(target)
            cfg = (tbl, order, int(lib.me_conv_gather_weight_elems_bf16(volume, c_src, c_dst)), _ptr(tbl), _ptr(order))
            km._launch_cache[ck] = cfg
        _, _, elems, p_tbl, p_order = cfg
        stream = _stream(dev)
        with _on(dev):
            packed = torch.empty(elems, dtype=torch.bfloat16, device=dev)
            _lib.check(lib.me_conv_gather_pack_weights_bf16(kernel.data_ptr(), 1 if kernel.dtype == torch.float32 else 0,
                                                            volume, c_src, c_dst, 1 if transposed else 0,
                                                            packed.data_ptr(), stream))
            _timed(name, dev, lambda: _lib.check(lib.me_conv_gather_bf16(
                src_feat.data_ptr(), src_feat.shape[0], c_src, packed.data_ptr(), km.volume, c_dst, p_tbl, p_order,
                out.data_ptr(), n_tgt, stream)), flops=2.0 * km.n_pairs * c_src * c_dst if KERNEL_TIMER else 0.0)
        return out
    if (bf16 and c_src == 8 and src_feat.shape[0] < (1 << 28) and
            lib.me_conv_stem_use_bf16(n_tgt, volume, c_src, c_dst)):
        # at most 8 source channels (a stem): four offsets per MFMA step straight off the neighbour table and the layer's
        # own kernel tensor — no tile plan, no packed image (csrc/conv_stem.hip)
        _check(kernel.dtype in (torch.float32, torch.bfloat16), "kernel must be float32 or bfloat16")
        ck = ("stem", target)
        cfg = km._launch_cache.get(ck)
        if cfg is None:
            tbl, order = km.table_pos(target)
            cfg = km._launch_cache[ck] = (tbl, order, _ptr(tbl), _ptr(order), int(lib.me_conv_stem_tile_rows()))
        _, _, p_tbl, p_order, tile_rows = cfg
        kern = kernel if kernel.is_contiguous() else kernel.contiguous()
        stream = _stream(dev)
        with _on(dev):
            want_stats = bool(_CONV_BN_STATS and _BN_STATS_HINT[0] and name == "conv_forward")
            part = torch.empty(2, -(-n_tgt // tile_rows), c_dst, dtype=torch.float32, device=dev) if want_stats else None
            _timed(name, dev, lambda: _lib.check(lib.me_conv_stem_bf16(
                src_feat.data_ptr(), src_feat.shape[0], c_src, kern.data_ptr(), 1 if kern.dtype == torch.float32 else 0,
                1 if transposed else 0, volume, c_dst, p_tbl, None, p_order, out.data_ptr(), n_tgt,
                part[0].data_ptr() if want_stats else None, part[1].data_ptr() if want_stats else None, stream)),
                flops=2.0 * km.n_pairs * c_src * c_dst)
        if want_stats:
            _bn_partials_put(out, part, tile_rows)
        return out
    rw = _rowwise_cfg(km, target, n_tgt, c_src, c_dst) if bf16 else None
    if rw is not None and name == "conv_forward" and _CONV_BN_STATS and _BN_STATS_HINT[0] and \
            n_tgt < _ROWWISE_MIN_ROWS_WITH_STATS:
        rw = None       # (the tile-plan kernel leaves the batch-norm statistics behind: cheaper on small maps)
    if rw is not None:
        # exactly one pair per target row (K = 1 layers — the reference's input.F.mm(kernel) —, the fine side of a
        # kernel_size == stride map): rows stream through the MFMA registers, no plan (csrc/conv_rowwise.hip)
        _check(kernel.dtype in (torch.float32, torch.bfloat16), "kernel must be float32 or bfloat16")
        src_rows, tgt_rows, elems = rw
        stream = _stream(dev)
        with _on(dev):
            if _PACK_CACHE:
                packed = _packed_weights(kernel, _lib.ME_PACK_BF16, transposed, c_src, c_dst, elems)
            else:
                packed = torch.empty(elems, dtype=torch.bfloat16, device=dev)
                _lib.check(lib.me_conv_pack_weights_bf16(kernel.data_ptr(), 1 if kernel.dtype == torch.float32 else 0,
                                                         volume, c_src, c_dst, 1 if transposed else 0,
                                                         packed.data_ptr(), stream))
            _timed(name, dev, lambda: _lib.check(lib.me_conv_rowwise_bf16(
                src_feat.data_ptr(), src_feat.shape[0], c_src, packed.data_ptr(), km.volume, c_dst, _ptr(src_rows),
                _ptr(tgt_rows), km.k_offsets_dev.data_ptr(), n_tgt, out.data_ptr(), n_tgt, stream)),
                flops=2.0 * km.n_pairs * c_src * c_dst if KERNEL_TIMER else 0.0)
        return out
    halo = _halo_launch_cfg(km, target, n_tgt, c_src, c_dst) if bf16 else None
    if halo is not None:
        # output-stationary launch on the LDS-staged source halo (csrc/conv_halo.hip): the same packed weights, the
        # batch-norm partials per halo tile
        _check(kernel.dtype in (torch.float32, torch.bfloat16), "kernel must be float32 or bfloat16")
        tile_rows, s_cap, halo_cnt, halo_rows, lidx, kmask, tbl, col_order, out_order = halo
        elems = int(lib.me_conv_packed_weight_elems_bf16(volume, c_src, c_dst))
        stream = _stream(dev)
        with _on(dev):
            if _PACK_CACHE:
                packed = _packed_weights(kernel, _lib.ME_PACK_BF16, transposed, c_src, c_dst, elems)
            else:
                packed = torch.empty(elems, dtype=torch.bfloat16, device=dev)
                _lib.check(lib.me_conv_pack_weights_bf16(kernel.data_ptr(), 1 if kernel.dtype == torch.float32 else 0,
                                                         volume, c_src, c_dst, 1 if transposed else 0,
                                                         packed.data_ptr(), stream))
            want_stats = bool(_CONV_BN_STATS and _BN_STATS_HINT[0] and name == "conv_forward")
            part = torch.empty(2, -(-n_tgt // tile_rows), c_dst, dtype=torch.float32, device=dev) if want_stats else None
            _timed(name, dev, lambda: _lib.check(lib.me_conv_halo_bf16(
                src_feat.data_ptr(), src_feat.shape[0], c_src, packed.data_ptr(), km.volume, c_dst, _ptr(halo_cnt),
                _ptr(halo_rows), _ptr(lidx), _ptr(kmask), _ptr(tbl), _ptr(col_order), _ptr(out_order), out.data_ptr(),
                n_tgt, tile_rows, s_cap, part[0].data_ptr() if want_stats else None,
                part[1].data_ptr() if want_stats else None, stream)), flops=2.0 * km.n_pairs * c_src * c_dst)
        if want_stats:
            _bn_partials_put(out, part, tile_rows)
        return out
    split, cfg = _conv_launch_cfg(km, target, n_tgt, c_src, c_dst, bf16)
    tile_rows, batch_groups, _, _, _, _, _, elems, p_src, p_dst, p_desc, p_bptr, p_order, fuse, split_k = cfg
    flops = 2.0 * km.n_pairs * c_src * c_dst
    stream = _stream(dev)
    with _on(dev):
        if bf16:
            # bf16 features: weights (fp32 master copy or bf16) are rounded to bf16 while being packed
            _check(kernel.dtype in (torch.float32, torch.bfloat16), "kernel must be float32 or bfloat16")
            if _PACK_CACHE:
                packed = _packed_weights(kernel, _lib.ME_PACK_BF16, transposed, c_src, c_dst, elems)
            else:
                packed = torch.empty(elems, dtype=torch.bfloat16, device=dev)
                _lib.check(lib.me_conv_pack_weights_bf16(kernel.data_ptr(), 1 if kernel.dtype == torch.float32 else 0,
                                                         volume, c_src, c_dst, 1 if transposed else 0,
                                                         packed.data_ptr(), stream))
            # one entry point for the bf16 forward / dgrad launch (me_conv_target_bf16_ex): batch fusion, the batch-norm
            # statistics of the output and split-K (small maps: G offset groups through an fp32 workspace) by argument
            want_stats = bool(_CONV_BN_STATS and _BN_STATS_HINT[0] and name == "conv_forward" and n_tgt > 0
                              and lib.me_conv_stats_supported_bf16(c_src, c_dst))
            part = None
            if want_stats:
                # the tiles' (mean, M2) partials ride along with the output (bn_stats picks them up when this very
                # tensor is normalised next: _BN_PARTIALS)
                n_tiles = -(-n_tgt // tile_rows)
                part = torch.empty(2, n_tiles, c_dst, dtype=torch.float32, device=dev)
            ws = None
            if split_k > 1:
                ws = torch.empty(int(lib.me_conv_splitk_workspace_bytes(n_tgt, tile_rows, c_dst, split_k)) // 4,
                                 dtype=torch.float32, device=dev)
            _timed(name, dev, lambda: _lib.check(lib.me_conv_target_bf16_ex(
                src_feat.data_ptr(), src_feat.shape[0], c_src, packed.data_ptr(), km.volume, c_dst, p_src, p_dst,
                p_desc, p_bptr, p_order, out.data_ptr(), n_tgt, tile_rows, batch_groups, 1 if fuse else 0, split_k,
                _ptr(ws), part[0].data_ptr() if want_stats else None, part[1].data_ptr() if want_stats else None,
                stream)), flops=flops)
            if want_stats:
                _bn_partials_put(out, part, tile_rows)
            return out
        _check(kernel.dtype == torch.float32, "float32 features need a float32 kernel, got", kernel.dtype)
        if split:
            if _PACK_CACHE:
                packed = _packed_weights(kernel, _lib.ME_PACK_F32X3, transposed, c_src, c_dst, elems)
            else:
                packed = torch.empty(elems, dtype=torch.bfloat16, device=dev)
                _lib.check(lib.me_conv_pack_weights_f32x3(kernel.data_ptr(), volume, c_src, c_dst,
                                                          1 if transposed else 0, packed.data_ptr(), stream))
            _timed(name, dev, lambda: _lib.check(lib.me_conv_target_f32x3(
                src_feat.data_ptr(), src_feat.shape[0], c_src, packed.data_ptr(), km.volume, c_dst, p_src, p_dst,
                p_desc, p_bptr, p_order, out.data_ptr(), n_tgt, tile_rows, batch_groups, stream)), flops=flops)
            return out
        packed = torch.empty(elems, dtype=torch.float32, device=dev)
        _lib.check(lib.me_conv_pack_weights_f32(kernel.data_ptr(), volume, c_src, c_dst, 1 if transposed else 0,
                                                packed.data_ptr(), stream))
        fn32 = lib.me_conv_target_f32_fused if (fuse and _F32_FUSE) else lib.me_conv_target_f32
        _timed(name, dev, lambda: _lib.check(fn32(
            src_feat.data_ptr(), src_feat.shape[0], c_src, packed.data_ptr(), km.volume, c_dst, p_src, p_dst, p_desc,
            p_bptr, p_order, out.data_ptr(), n_tgt, tile_rows, batch_groups, stream)), flops=flops)
    return out


def _conv_forward(in_feat, kernel, km, algo=None):
    algo = algo or _ALGO
    if algo == "naive":
        _check(in_feat.dtype == torch.float32, "the cross-check kernels are float32 only")
        lib = _lib.load()
        dev = in_feat.device
        out = torch.zeros((km.n_out, kernel.shape[2]), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.check(lib.me_conv_forward_naive_f32(_ptr(in_feat), kernel.shape[1], _ptr(kernel), kernel.shape[2],
                                                     _ptr(km.in_pairs), _ptr(km.out_pairs), _ptr(km.k_offsets_dev),
                                                     km.volume, km.n_pairs, _ptr(out), _stream(dev)))
        return out
    return _conv_target(in_feat, kernel, km, "out", km.n_out, name="conv_forward")


def _conv_backward(in_feat, grad_out, kernel, km, algo=None, need_grad_in=True):
    algo = algo or _ALGO
    lib = _lib.load()
    dev = in_feat.device
    volume, c_in, c_out = int(kernel.shape[0]), int(kernel.shape[1]), int(kernel.shape[2])
    if grad_out.dtype != in_feat.dtype:
        grad_out = grad_out.to(in_feat.dtype)
    bf16 = in_feat.dtype == torch.bfloat16
    if in_feat.dtype == torch.float64:     # csrc/f64.hip: deterministic double sums over the pair lists / the table
        _check(kernel.dtype == torch.float64, "float64 features need a float64 kernel, got", kernel.dtype)
        grad_w = torch.empty(kernel.shape, dtype=torch.float64, device=dev)
        in_feat, grad_out = in_feat.contiguous(), grad_out.contiguous()
        with _on(dev):
            _lib.check(lib.me_conv_wgrad_f64(in_feat.data_ptr(), c_in, grad_out.data_ptr(), c_out, _ptr(km.in_pairs_buf),
                                             _ptr(km.out_pairs_buf), _ptr(km.k_offsets_dev), volume, grad_w.data_ptr(),
                                             _stream(dev)))
        grad_in = _conv_target(grad_out, kernel, km, "in", km.n_in, name="conv_dgrad", transposed=True) \
            if need_grad_in else None
        return grad_in, grad_w
    if algo == "naive":
        _check(in_feat.dtype == torch.float32, "the cross-check kernels are float32 only")
        grad_in = torch.zeros((km.n_in, c_in), dtype=torch.float32, device=dev)
        grad_w = torch.zeros_like(kernel)
        with _on(dev):
            _lib.check(lib.me_conv_backward_naive_f32(_ptr(in_feat), c_in, _ptr(grad_out), c_out, _ptr(kernel),
                                                      _ptr(km.in_pairs), _ptr(km.out_pairs), _ptr(km.k_offsets_dev),
                                                      km.volume, km.n_pairs, _ptr(grad_in), _ptr(grad_w),
                                                      _stream(dev)))
        return grad_in, grad_w
    # wgrad: always accumulated and reduced in fp32; handed back in the kernel's dtype.  The two gradients of a layer
    # are independent (both read grad_out; dgrad reads the weights, wgrad the layer's input); with ME_AMD_WGRAD_STREAM=1
    # the weight gradient goes to a SIDE STREAM next to the input gradient (opt-in: see _WGRAD_STREAM).  Fork / join
    # inside this call (side waits for what the current stream has produced; the current stream waits for the side
    # stream before anything downstream can touch grad_w), so callers — autograd's AccumulateGrad, DDP's bucket
    # hooks, a hipGraph capture — see plain stream-ordered tensors.
    grad_w = _grad_destination(kernel, kernel.shape) if kernel.dtype == torch.float32 else None
    if grad_w is None:
        grad_w = torch.empty(kernel.shape, dtype=torch.float32, device=dev)
    cfg = _wgrad_launch_cfg(km, c_in, c_out, bf16)
    koffs, wsb, p_in, p_out, p_koffs = cfg
    if _WGRAD_TUNING:   # the debug switches change the workspace need
        wsb = int((lib.me_conv_wgrad_workspace_bytes_bf16 if bf16 else lib.me_conv_wgrad_workspace_bytes)(
            koffs, volume, c_in, c_out))
    fn = lib.me_conv_wgrad_bf16 if bf16 else lib.me_conv_wgrad_f32

    # (the workspace comes from the current stream's pool and is used on the side stream: safe, because the current
    # stream joins the side stream below before anything else can be enqueued on it)
    ws = _workspace(wsb, dev)

    def wgrad(stream):
        with _on(dev):
            launch = lambda: _lib.check(fn(
                in_feat.data_ptr(), in_feat.shape[0], c_in, grad_out.data_ptr(), grad_out.shape[0], c_out, p_in, p_out,
                koffs, p_koffs, volume, grad_w.data_ptr(), ws.data_ptr(), ws.numel(), stream))
            if KERNEL_TIMER is None:
                launch()
            else:   # (per-launch HIP events belong on the stream of the launch)
                with torch.cuda.stream(side) if side is not None else _on(dev):
                    _timed("conv_wgrad", dev, launch, flops=2.0 * km.n_pairs * c_in * c_out)

    side = _side_stream(dev) if (_WGRAD_STREAM and need_grad_in and km.n_pairs > 0 and
                                 max(km.n_in, km.n_out) <= _WGRAD_STREAM_MAX_ROWS) else None
    if side is not None:
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        wgrad(side.cuda_stream)
    # dgrad: the same target-stationary kernel; the weights are packed transposed per offset
    grad_in = _conv_target(grad_out, kernel, km, "in", km.n_in, name="conv_dgrad", transposed=True) \
        if need_grad_in else None
    if side is not None:
        main.wait_stream(side)
    else:
        wgrad(_stream(dev))
    return grad_in, grad_w if kernel.dtype == torch.float32 else grad_w.to(kernel.dtype)


def _prepare_conv(in_feat, kernel, kernel_size, kernel_stride, kernel_dilation, region_type, expand_coordinates,
                  in_key, out_key, manager, transpose):
    _check_feat("in_feat", in_feat)
    _check_feat("kernel", kernel)
    _check(in_feat.dim() == 2, "in_feat.dim():", in_feat.dim())
    _check(kernel.dim() == 3, "kernel.dim():", kernel.dim())
    _check(in_feat.shape[1] == kernel.shape[1], "Input feature size and kernel size mismatch")
    _check(manager.exists(in_key), "coordinate map not found")
    _check(in_feat.shape[0] == manager.size(in_key), "Invalid in_feat size", in_feat.shape[0], "!=",
           manager.size(in_key))
    if out_key.is_key_set():
        return
    ts = in_key.get_tensor_stride()
    st = [int(s) for s in kernel_stride]
    if not transpose:
        if expand_coordinates:
            # src/convolution_cpu.cpp:79-103: every kernel offset around every input voxel that falls on the
            # output grid becomes an output voxel
            out_ts = [t * s for t, s in zip(ts, st)]
            key, _ = manager.stride_region(in_key, kernel_size, kernel_dilation, region_type, out_ts, True, False,
                                           region_tensor_stride=ts)
            out_key.set_key(key.get_key())
        else:
            out_key.set_key(manager.stride(in_key, kernel_stride).get_key())
    else:
        # src/convolution_transpose_cpu.cpp:76-97: out tensor stride = in / stride; the existing map of that
        # stride is reused unless new coordinates are to be generated
        _check(all(t % s == 0 for t, s in zip(ts, st)), "Invalid up stride on tensor stride:", ts,
               "kernel stride:", st)
        out_ts = [t // s for t, s in zip(ts, st)]
        key, _ = manager.stride_region(in_key, kernel_size, kernel_dilation, region_type, out_ts,
                                       bool(expand_coordinates), True)
        out_key.set_key(key.get_key())


def ConvolutionForwardGPU(in_feat, kernel, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                          expand_coordinates, convolution_mode, in_key, out_key, manager):
    """src/convolution_gpu.cu:45-159 (CPU twin src/convolution_cpu.cpp:42-135)."""
    _prepare_conv(in_feat, kernel, kernel_size, kernel_stride, kernel_dilation, region_type, expand_coordinates, in_key,
                  out_key, manager, False)
    km = manager._kernel_map(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                             False, False)
    return _conv_forward(in_feat, kernel, km)


def ConvolutionBackwardGPU(in_feat, grad_out_feat, kernel, kernel_size, kernel_stride, kernel_dilation,
                           region_type, offset, convolution_mode, in_key, out_key, manager, need_grad_in=True):
    """src/convolution_gpu.cu:161-244 -> (grad_in_feat, grad_kernel).  need_grad_in (not in the reference, optional):
    False skips the input gradient (returned as None)."""
    _check_feat("in_feat", in_feat)
    _check_feat("kernel", kernel)
    if not grad_out_feat.is_contiguous():
        grad_out_feat = grad_out_feat.contiguous()
    _check_feat("grad_out_feat", grad_out_feat)
    _check(in_feat.shape[1] == kernel.shape[1], "Input feature size and kernel size mismatch")
    _check(grad_out_feat.shape[1] == kernel.shape[2], "Output feature size and kernel size mismatch")
    km = manager._kernel_map(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                             False, False)
    _check(grad_out_feat.shape[0] == km.n_out, "Invalid grad_out size")
    return _conv_backward(in_feat, grad_out_feat, kernel, km, need_grad_in=need_grad_in)


def ConvolutionTransposeForwardGPU(in_feat, kernel, kernel_size, kernel_stride, kernel_dilation, region_type,
                                   offset, expand_coordinates, convolution_mode, in_key, out_key, manager):
    """src/convolution_transpose_gpu.cu (CPU twin src/convolution_transpose_cpu.cpp:41-125)."""
    _prepare_conv(in_feat, kernel, kernel_size, kernel_stride, kernel_dilation, region_type, expand_coordinates, in_key,
                  out_key, manager, True)
    km = manager._kernel_map(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                             True, False)
    return _conv_forward(in_feat, kernel, km)


def ConvolutionTransposeBackwardGPU(in_feat, grad_out_feat, kernel, kernel_size, kernel_stride,
                                    kernel_dilation, region_type, offset, convolution_mode, in_key, out_key,
                                    manager, need_grad_in=True):
    """src/convolution_transpose_cpu.cpp:127-191 -> (grad_in_feat, grad_kernel)."""
    _check_feat("in_feat", in_feat)
    _check_feat("kernel", kernel)
    if not grad_out_feat.is_contiguous():
        grad_out_feat = grad_out_feat.contiguous()
    _check_feat("grad_out_feat", grad_out_feat)
    km = manager._kernel_map(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                             True, False)
    _check(grad_out_feat.shape[0] == km.n_out, "Invalid grad_out size")
    return _conv_backward(in_feat, grad_out_feat, kernel, km, need_grad_in=need_grad_in)


# ------------------------------------------------------------------------------------------------
# pooling / broadcast operators (src/local_pooling_cpu.cpp, src/local_pooling_transpose_cpu.cpp,
# src/global_pooling_cpu.cpp, src/broadcast_cpu.cpp; signatures pybind/extern.hpp:187-392)
# ------------------------------------------------------------------------------------------------
def _same_dtype(name, t, like):
    """feature-shaped operands of one pooling / broadcast call share the dtype of the input features"""
    _check(t.dtype == like.dtype, name, "must have the dtype of the input features:", t.dtype, "vs", like.dtype)


def _pool_sum(src, tbl, n_tgt, volume, src_count=None, average=False, want_count=False):
    """dtype dispatch: float32 rows run me_pool_sum_f32, bfloat16 rows me_pool_sum_bf16 (fp32 sums, one rounding
    at the store); the output has the dtype of `src`, counts are float32."""
    lib = _lib.load()
    dev = src.device
    c = int(src.shape[1])
    out = torch.empty((n_tgt, c), dtype=src.dtype, device=dev)
    cnt = torch.empty(max(n_tgt, 1), dtype=torch.float32, device=dev)[:n_tgt] if want_count else None
    if src_count is not None:
        _check(src_count.dtype == torch.float32, "num_nonzero must be float32")
    fn = _by_dtype(lib, "pool_sum", src)
    with _on(dev):
        _timed("pool_sum", dev, lambda: _lib.check(fn(
            _ptr(src), c, _ptr(tbl), n_tgt, volume, _ptr(src_count), 1 if average else 0, _ptr(out), _ptr(cnt),
            _stream(dev))))
    return out, cnt


def _by_dtype(lib, base, t):
    """`me_<base>_{f32,bf16,f64}` for a tensor's dtype (float64: csrc/f64.hip, the reference's double instantiation)"""
    return getattr(lib, f"me_{base}_" + {torch.float32: "f32", torch.bfloat16: "bf16", torch.float64: "f64"}[t.dtype])


def _prepare_pool(in_feat, kernel_stride, in_key, out_key, manager, transpose):
    _check_feat("in_feat", in_feat)
    _check(in_feat.dim() == 2, "in_feat.dim():", in_feat.dim())
    _check(manager.exists(in_key), "coordinate map not found")
    _check(in_feat.shape[0] == manager.size(in_key), "Invalid in_feat size", in_feat.shape[0], "!=",
           manager.size(in_key))
    if not out_key.is_key_set():
        if not transpose:
            out_key.set_key(manager.stride(in_key, kernel_stride).get_key())
        else:
            ts = in_key.get_tensor_stride()
            st = [int(s) for s in kernel_stride]
            _check(all(t % s == 0 for t, s in zip(ts, st)), "Invalid up stride on tensor stride:", ts)
            cand = ([t // s for t, s in zip(ts, st)], "")
            _check(manager.exists(cand), "pooling transpose needs an existing output map")
            out_key.set_key(cand)


def LocalPoolingForwardGPU(in_feat, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                           pooling_mode, in_key, out_key, manager):
    """src/local_pooling_cpu.cpp:43-122 -> (out_feat, num_nonzero | max_index)."""
    _prepare_pool(in_feat, kernel_stride, in_key, out_key, manager, False)
    km = manager._kernel_map(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                             False, True)
    mode = PoolingMode(int(pooling_mode))
    if mode == PoolingMode.LOCAL_MAX_POOLING:
        lib = _lib.load()
        dev = in_feat.device
        c = int(in_feat.shape[1])
        out = torch.empty((km.n_out, c), dtype=in_feat.dtype, device=dev)
        mask = torch.empty((km.n_out, c), dtype=torch.int32, device=dev)
        fn = _by_dtype(lib, "pool_max", in_feat)
        with _on(dev):
            _timed("pool_max", dev, lambda: _lib.check(fn(
                _ptr(in_feat), c, _ptr(km.table("out")), km.n_out, km.volume, _ptr(out), _ptr(mask), _stream(dev))))
        return out, mask
    _check(mode in (PoolingMode.LOCAL_SUM_POOLING, PoolingMode.LOCAL_AVG_POOLING), "Invalid pooling mode")
    avg = mode == PoolingMode.LOCAL_AVG_POOLING
    out, cnt = _pool_sum(in_feat, km.table("out"), km.n_out, km.volume, average=avg, want_count=avg)
    if cnt is None:
        cnt = torch.empty(0, dtype=torch.float32, device=in_feat.device)
    return out, cnt


def LocalPoolingBackwardGPU(in_feat, grad_out_feat, num_nonzero, kernel_size, kernel_stride, kernel_dilation,
                            region_type, offset, pooling_mode, in_key, out_key, manager):
    """src/local_pooling_cpu.cpp:124-214 -> grad_in_feat."""
    _check_feat("in_feat", in_feat)
    if not grad_out_feat.is_contiguous():
        grad_out_feat = grad_out_feat.contiguous()
    _check_feat("grad_out_feat", grad_out_feat)
    if grad_out_feat.dtype != in_feat.dtype:
        grad_out_feat = grad_out_feat.to(in_feat.dtype)
    km = manager._kernel_map(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                             False, True)
    _check(grad_out_feat.shape[0] == km.n_out, "Invalid grad_out size")
    mode = PoolingMode(int(pooling_mode))
    if mode == PoolingMode.LOCAL_MAX_POOLING:
        lib = _lib.load()
        dev = in_feat.device
        c = int(in_feat.shape[1])
        _check(num_nonzero.dtype == torch.int32, "the max-pooling mask must be int32")
        grad_in = torch.empty((km.n_in, c), dtype=in_feat.dtype, device=dev)
        fn = _by_dtype(lib, "pool_max_backward", in_feat)
        with _on(dev):
            _lib.check(fn(_ptr(grad_out_feat), c, _ptr(km.table("in")), km.n_in, km.volume,
                          _ptr(num_nonzero), _ptr(grad_in), _stream(dev)))
        return grad_in
    avg = mode == PoolingMode.LOCAL_AVG_POOLING
    grad_in, _ = _pool_sum(grad_out_feat, km.table("in"), km.n_in, km.volume,
                           src_count=num_nonzero if avg else None)
    return grad_in


def LocalPoolingTransposeForwardGPU(in_feat, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                                    generate_new_coordinates, pooling_mode, in_key, out_key, manager):
    """src/local_pooling_transpose_cpu.cpp:41-115: unpooling = sum over the transposed map -> (out, num_nonzero)."""
    _check(not generate_new_coordinates, "generate_new_coordinates (stride_region) is not part of the hot path yet")
    _prepare_pool(in_feat, kernel_stride, in_key, out_key, manager, True)
    km = manager._kernel_map(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                             True, True)
    out, cnt = _pool_sum(in_feat, km.table("out"), km.n_out, km.volume, average=False, want_count=True)
    return out, cnt


def LocalPoolingTransposeBackwardGPU(in_feat, grad_out_feat, num_nonzero, kernel_size, kernel_stride,
                                     kernel_dilation, region_type, offset, pooling_mode, in_key, out_key, manager):
    """src/local_pooling_transpose_cpu.cpp:117-190 -> grad_in_feat."""
    if not grad_out_feat.is_contiguous():
        grad_out_feat = grad_out_feat.contiguous()
    _check_feat("grad_out_feat", grad_out_feat)
    if grad_out_feat.dtype != in_feat.dtype:
        grad_out_feat = grad_out_feat.to(in_feat.dtype)
    km = manager._kernel_map(in_key, out_key, kernel_size, kernel_stride, kernel_dilation, region_type, offset,
                             True, True)
    grad_in, _ = _pool_sum(grad_out_feat, km.table("in"), km.n_in, km.volume)
    return grad_in


_GLOBAL_SUM = (PoolingMode.GLOBAL_SUM_POOLING_DEFAULT, PoolingMode.GLOBAL_SUM_POOLING_KERNEL,
               PoolingMode.GLOBAL_SUM_POOLING_PYTORCH_INDEX)
_GLOBAL_AVG = (PoolingMode.GLOBAL_AVG_POOLING_DEFAULT, PoolingMode.GLOBAL_AVG_POOLING_KERNEL,
               PoolingMode.GLOBAL_AVG_POOLING_PYTORCH_INDEX)
_GLOBAL_MAX = (PoolingMode.GLOBAL_MAX_POOLING_DEFAULT, PoolingMode.GLOBAL_MAX_POOLING_KERNEL,
               PoolingMode.GLOBAL_MAX_POOLING_PYTORCH_INDEX)


def _global_pool(src, src2, rows, n_batch, mode):
    """mode 0 sum / 1 avg / 2 max over the rows of each origin row -> (out, argmax | None, count | None)."""
    lib = _lib.load()
    dev = src.device
    n, c = int(src.shape[0]), int(src.shape[1])
    if src2 is not None:
        _same_dtype("the second factor", src2, src)
    # the kernels reduce in fp32 and write fp32 [n_batch, c]; bf16 inputs get their (tiny) result rounded here
    f64 = src.dtype == torch.float64
    out = torch.empty((n_batch, c), dtype=torch.float64 if f64 else torch.float32, device=dev)
    arg = torch.empty((n_batch, c), dtype=torch.int32, device=dev) if mode == 2 else None
    cnt = torch.empty(n_batch, dtype=torch.float32, device=dev) if mode != 2 else None
    if f64:
        with _on(dev):
            _lib.check(lib.me_global_pool_f64(_ptr(src), _ptr(src2), c, _ptr(rows), n, n_batch, mode, _ptr(out), _ptr(arg),
                                              _ptr(cnt), _stream(dev)))
        return out, arg, cnt
    ws = _workspace(lib.me_global_pool_workspace_bytes(n, n_batch, c), dev)
    fn = lib.me_global_pool_bf16 if src.dtype == torch.bfloat16 else lib.me_global_pool_f32
    with _on(dev):
        _lib.check(fn(_ptr(src), _ptr(src2), c, _ptr(rows), n, n_batch, mode, _ptr(out), _ptr(arg),
                      _ptr(cnt), _ptr(ws), ws.numel(), _stream(dev)))
    if src.dtype != torch.float32:
        out = out.to(src.dtype)
    return out, arg, cnt


def GlobalPoolingForwardGPU(in_feat, pooling_mode, in_key, out_key, manager):
    """src/global_pooling_cpu.cpp:43-238 -> (out_feat [batch, C], num_nonzero [batch] | max_index [batch, C]).
    max_index holds flat indices row * C + channel for every batch size (the reference returns plain row
    indices when the batch size is 1, global_pooling_cpu.cpp:99-101, and then mis-scatters them)."""
    _check_feat("in_feat", in_feat)
    _check(in_feat.dim() == 2, "Invalid in_feat.dim():", in_feat.dim())
    _check(manager.exists(in_key), "coordinate map not found")
    _check(in_feat.shape[0] == manager.size(in_key), "Invalid in_feat size")
    mode = PoolingMode(int(pooling_mode))
    _check(mode in _GLOBAL_SUM + _GLOBAL_AVG + _GLOBAL_MAX, "Invalid pooling mode")
    if not out_key.is_key_set():
        out_key.set_key(manager.origin().get_key())
    rows = manager._origin_rows(in_key)
    n_batch = manager.size(out_key)
    m = 2 if mode in _GLOBAL_MAX else (1 if mode in _GLOBAL_AVG else 0)
    out, arg, cnt = _global_pool(in_feat, None, rows, n_batch, m)
    return out, (arg if m == 2 else cnt)


def _broadcast(in_feat, glob, rows, n, c, multiply):
    """out[i] = in[i] (+ | *) glob[rows[i]]  (in_feat None: out[i] = glob[rows[i]]); dtype of `glob`."""
    lib = _lib.load()
    dev = glob.device
    if in_feat is not None:
        _same_dtype("in_feat", in_feat, glob)
    out = torch.empty((n, c), dtype=glob.dtype, device=dev)
    fn = _by_dtype(lib, "broadcast", glob)
    with _on(dev):
        _lib.check(fn(_ptr(in_feat), _ptr(glob), _ptr(rows), n, c, 1 if multiply else 0, _ptr(out), _stream(dev)))
    return out


def GlobalPoolingBackwardGPU(in_feat, grad_out_feat, num_nonzero, pooling_mode, in_key, out_key, manager):
    """src/global_pooling_cpu.cpp:240-326 -> grad_in_feat (dtype of in_feat)."""
    _check_feat("in_feat", in_feat)
    if not grad_out_feat.is_contiguous():
        grad_out_feat = grad_out_feat.contiguous()
    _check_feat("grad_out_feat", grad_out_feat)
    if grad_out_feat.dtype != in_feat.dtype:
        grad_out_feat = grad_out_feat.to(in_feat.dtype)
    mode = PoolingMode(int(pooling_mode))
    n, c = int(in_feat.shape[0]), int(in_feat.shape[1])
    dev = in_feat.device
    if mode in _GLOBAL_MAX:
        grad_in = torch.zeros((n, c), dtype=in_feat.dtype, device=dev)
        valid = num_nonzero.reshape(-1) >= 0
        grad_in.view(-1)[num_nonzero.reshape(-1)[valid].long()] = grad_out_feat.reshape(-1)[valid]
        return grad_in
    g = grad_out_feat
    if mode in _GLOBAL_AVG:
        g = ((g if g.dtype == torch.float64 else g.float()) / num_nonzero.clamp_min(1.0)[:, None]) \
            .to(in_feat.dtype).contiguous()
    rows = manager._origin_rows(in_key)
    return _broadcast(None, g, rows, n, c, False)


def BroadcastForwardGPU(in_feat, in_feat_glob, broadcast_mode, in_key, glob_key, manager):
    """src/broadcast_cpu.cpp:38-97: out[i] = in[i] (+ | *) glob[batch of i]."""
    _check_feat("in_feat", in_feat)
    _check_feat("in_feat_glob", in_feat_glob)
    _same_dtype("in_feat_glob", in_feat_glob, in_feat)
    _check(in_feat.shape[1] == in_feat_glob.shape[1], "feature sizes must match")
    _check(in_feat.shape[0] == manager.size(in_key), "Invalid in_feat size")
    _check(in_feat_glob.shape[0] == manager.size(glob_key), "Invalid in_feat_glob size")
    op = BroadcastMode(int(broadcast_mode))
    rows = manager._origin_rows(in_key)
    return _broadcast(in_feat, in_feat_glob, rows, int(in_feat.shape[0]), int(in_feat.shape[1]),
                      op == BroadcastMode.ELEMENTWISE_MULTIPLICATION)


def BroadcastBackwardGPU(in_feat, in_feat_glob, grad_out_feat, broadcast_mode, in_key, glob_key, manager):
    """src/broadcast_cpu.cpp:99-160 -> (grad_in_feat, grad_in_feat_glob)."""
    _check_feat("in_feat", in_feat)
    _check_feat("in_feat_glob", in_feat_glob)
    _same_dtype("in_feat_glob", in_feat_glob, in_feat)
    if not grad_out_feat.is_contiguous():
        grad_out_feat = grad_out_feat.contiguous()
    _check_feat("grad_out_feat", grad_out_feat)
    if grad_out_feat.dtype != in_feat.dtype:
        grad_out_feat = grad_out_feat.to(in_feat.dtype)
    op = BroadcastMode(int(broadcast_mode))
    rows = manager._origin_rows(in_key)
    n_batch = int(in_feat_glob.shape[0])
    if op == BroadcastMode.ELEMENTWISE_ADDITON:
        grad_in = grad_out_feat.clone()
        grad_glob, _, _ = _global_pool(grad_out_feat, None, rows, n_batch, 0)
    else:
        grad_in = _broadcast(grad_out_feat, in_feat_glob, rows, int(in_feat.shape[0]), int(in_feat.shape[1]), True)
        grad_glob, _, _ = _global_pool(grad_out_feat, in_feat, rows, n_batch, 0)
    return grad_in, grad_glob


# ------------------------------------------------------------------------------------------------
# batch normalisation over feature rows (csrc/norm.hip; the reference applies torch.nn.BatchNorm1d to the
# feature matrix, MinkowskiNormalization.py:35-82)
# ------------------------------------------------------------------------------------------------
def _bn_check(x):
    _check(x.is_cuda and x.is_contiguous() and x.dim() == 2, "batch norm input must be a contiguous GPU matrix")
    _check(x.dtype in (torch.float32, torch.bfloat16), "batch norm input must be float32 or bfloat16, got", x.dtype)
    _check(x.shape[0] > 0, "batch norm needs at least one row")


# Statistics a convolution left behind for the batch norm that follows it (me_conv_target_bf16_stats): keyed by the
# output's storage address, valid for THAT tensor object at THAT version only (weak reference + version counter: a
# recycled address, a view, an in-place update or a channel slice simply misses and bn_stats reads the matrix).  One
# entry per live convolution output; consumed (popped) by the first bn_stats on it.
_CONV_BN_STATS = os.environ.get("ME_AMD_CONV_BN_STATS", "1") != "0"
_BN_PARTIALS = {}
class _ThreadFlag(threading.local):
    """[0] of a per-thread flag (a loader thread never sees the training thread's value)"""
    value = False

    def __getitem__(self, i):
        return self.value

    def __setitem__(self, i, v):
        self.value = v


_BN_STATS_HINT = _ThreadFlag()     # set by the convolution module around its forward call: True in training mode


def conv_bn_stats_hint(flag):
    """MinkowskiConvolution tells the operator whether a training-mode batch norm may follow (module.training): the
    operator interface of the reference has no argument for it."""
    _BN_STATS_HINT[0] = bool(flag)


def _bn_partials_put(out, part, tile_rows):
    key = out.data_ptr()

    def _gone(ref, key=key):     # the output was never normalised and has been collected: drop its entry
        ent = _BN_PARTIALS.get(key)
        if ent is not None and ent[0] is ref:
            del _BN_PARTIALS[key]
    _BN_PARTIALS[key] = (weakref.ref(out, _gone), out._version, part, int(tile_rows), tuple(out.shape))


def _bn_partials_take(x):
    ent = _BN_PARTIALS.pop(x.data_ptr(), None) if _BN_PARTIALS else None
    if ent is None:
        return None
    ref, version, part, tile_rows, shape = ent
    if ref() is not x or x._version != version or tuple(x.shape) != shape:
        return None
    return part, tile_rows


def _bn_stat_ok(t, c):
    return t is None or (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.numel() == c)


def _bn_check_vec(name, t, c):
    _check(_bn_stat_ok(t, c), f"{name} must be a contiguous float32 GPU vector of one value per channel")


def bn_stats(x, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None):
    """-> (mean, rstd) float32 [c] of the batch; running statistics updated in place when given (and the int64
    num_batches_tracked buffer incremented by the same kernel).  The kernels read and write the running statistics as
    float32 [c]: buffers of another dtype / layout (bf16 after `model.bfloat16()`) go through float32 temporaries
    that are copied back, a counter that is not an int64 GPU scalar is incremented by torch."""
    _bn_check(x)
    lib = _lib.load()
    dev = x.device
    n, c = int(x.shape[0]), int(x.shape[1])
    _check((running_mean is None) == (running_var is None), "running_mean and running_var: both or none")
    _check(running_mean is None or (running_mean.numel() == c and running_var.numel() == c),
           "running statistics must hold one value per channel")
    rm_user, rv_user = running_mean, running_var
    direct = _bn_stat_ok(running_mean, c) and _bn_stat_ok(running_var, c)
    if not direct:
        running_mean = rm_user.to(device=dev, dtype=torch.float32).contiguous().clone()
        running_var = rv_user.to(device=dev, dtype=torch.float32).contiguous().clone()
    if num_batches_tracked is not None and not (num_batches_tracked.is_cuda and num_batches_tracked.dtype == torch.int64
                                                and num_batches_tracked.numel() == 1):
        num_batches_tracked.add_(1)
        num_batches_tracked = None
    mean = torch.empty(c, dtype=torch.float32, device=dev)
    rstd = torch.empty(c, dtype=torch.float32, device=dev)
    got = _bn_partials_take(x)
    if got is not None:
        part, tile_rows = got
        with _on(dev):
            _lib.check(lib.me_bn_stats_from_tiles(part[0].data_ptr(), part[1].data_ptr(), n, c, tile_rows, float(eps),
                                                  float(momentum), _ptr(mean), _ptr(rstd), _ptr(running_mean),
                                                  _ptr(running_var), _ptr(num_batches_tracked), _stream(dev)))
    else:
        ws = _workspace(int(lib.me_bn_workspace_bytes(n, c)), dev)
        with _on(dev):
            _lib.check(lib.me_bn_stats(_ptr(x), 1 if x.dtype == torch.bfloat16 else 0, n, c, float(eps),
                                       float(momentum), _ptr(mean), _ptr(rstd), _ptr(running_mean), _ptr(running_var),
                                       _ptr(num_batches_tracked), _ptr(ws), ws.numel(), _stream(dev)))
    if not direct:
        rm_user.copy_(running_mean)
        rv_user.copy_(running_var)
    return mean, rstd


def _bn_check_params(x, mean, rstd, gamma, beta):
    c = int(x.shape[1])
    _check(mean is not None and rstd is not None, "batch norm needs mean and rstd")
    for name, t in (("mean", mean), ("rstd", rstd), ("gamma", gamma), ("beta", beta)):
        _bn_check_vec(name, t, c)


def bn_apply(x, mean, rstd, gamma, beta, relu=False):
    _bn_check(x)
    _bn_check_params(x, mean, rstd, gamma, beta)
    lib = _lib.load()
    dev = x.device
    y = torch.empty_like(x)
    with _on(dev):
        _lib.check(lib.me_bn_apply(_ptr(x), 1 if x.dtype == torch.bfloat16 else 0, int(x.shape[0]), int(x.shape[1]),
                                   _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), 1 if relu else 0, _ptr(y),
                                   _stream(dev)))
    return y


def bn_apply_residual(x, skip, mean, rstd, gamma, beta, relu=True):
    """y = [relu] (bn(x) + skip): batch-norm apply, residual addition and ReLU of a ResNet block in one pass
    (bit-identical to the three separate kernels)."""
    _bn_check(x)
    _bn_check_params(x, mean, rstd, gamma, beta)
    _check(skip.shape == x.shape and skip.dtype == x.dtype and skip.is_contiguous(), "residual branch must match x")
    lib = _lib.load()
    dev = x.device
    y = torch.empty_like(x)
    with _on(dev):
        _lib.check(lib.me_bn_apply_residual(_ptr(x), _ptr(skip), 1 if x.dtype == torch.bfloat16 else 0,
                                            int(x.shape[0]), int(x.shape[1]), _ptr(mean), _ptr(rstd), _ptr(gamma),
                                            _ptr(beta), 1 if relu else 0, _ptr(y), _stream(dev)))
    return y


def bn_backward_residual(x, dy, yout, mean, rstd, gamma, beta=None, relu=True, need_dskip=True):
    """-> (dx, dskip, grad_gamma, grad_beta) of y = [relu] (bn(x) + skip); dskip is the (ReLU-masked) incoming
    gradient, dy itself without ReLU."""
    _bn_check(x)
    _bn_check_params(x, mean, rstd, gamma, beta)
    lib = _lib.load()
    dev = x.device
    n, c = int(x.shape[0]), int(x.shape[1])
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    dskip = torch.empty_like(x) if (need_dskip and relu) else None
    gg, gb = _grad_destination(gamma, (c,)), _grad_destination(beta, (c,))
    gg = torch.empty(c, dtype=torch.float32, device=dev) if gg is None else gg
    gb = torch.empty(c, dtype=torch.float32, device=dev) if gb is None else gb
    ws = _workspace(int(lib.me_bn_workspace_bytes(n, c)), dev)
    with _on(dev):
        _lib.check(lib.me_bn_backward_residual(_ptr(x), _ptr(dy), _ptr(yout), 1 if x.dtype == torch.bfloat16 else 0, n,
                                               c, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), 1 if relu else 0,
                                               _ptr(dx), _ptr(dskip), _ptr(gg), _ptr(gb), _ptr(ws), ws.numel(),
                                               _stream(dev)))
    return dx, (dskip if relu else dy) if need_dskip else None, gg, gb


def bn_backward(x, dy, mean, rstd, gamma, beta=None, relu=False):
    """-> (dx, grad_gamma, grad_beta) of training-mode batch norm (followed by a fused ReLU when `relu`)."""
    _bn_check(x)
    _bn_check_params(x, mean, rstd, gamma, beta)
    lib = _lib.load()
    dev = x.device
    n, c = int(x.shape[0]), int(x.shape[1])
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    gg, gb = _grad_destination(gamma, (c,)), _grad_destination(beta, (c,))
    gg = torch.empty(c, dtype=torch.float32, device=dev) if gg is None else gg
    gb = torch.empty(c, dtype=torch.float32, device=dev) if gb is None else gb
    ws = _workspace(int(lib.me_bn_workspace_bytes(n, c)), dev)
    with _on(dev):
        _lib.check(lib.me_bn_backward(_ptr(x), _ptr(dy), 1 if x.dtype == torch.bfloat16 else 0, n, c, _ptr(mean),
                                      _ptr(rstd), _ptr(gamma), _ptr(beta), 1 if relu else 0, _ptr(dx), _ptr(gg),
                                      _ptr(gb), _ptr(ws), ws.numel(), _stream(dev)))
    return dx, gg, gb


# ------------------------------------------------------------------------------------------------
# pruning (src/pruning_cpu.cpp:40-150, src/pruning_gpu.cu)
# ------------------------------------------------------------------------------------------------
def PruningForwardGPU(in_feat, keep, in_key, out_key, manager):
    """Rows of `in_feat` where `keep` is true, on a new coordinate map (created here unless `out_key` is set)."""
    _check_feat("in_feat", in_feat)
    _check(keep.dtype in (torch.bool, torch.uint8), "keep must be a boolean tensor")
    _check(in_feat.dim() == 2 and keep.dim() == 1, "in_feat.dim():", in_feat.dim(), "keep.dim():", keep.dim())
    _check(in_feat.shape[0] == keep.shape[0], "Input feature size and keep size mismatch")
    _check(manager.exists(in_key), "coordinate map not found")
    _check(in_feat.shape[0] == manager.size(in_key), "Invalid in_feat size", in_feat.shape[0], "!=",
           manager.size(in_key))
    if not out_key.is_key_set():
        out_key.set_key(manager.prune(in_key, keep).get_key())
    rows = manager._pruning_rows(in_key, out_key)
    return in_feat.index_select(0, rows.long())


def PruningBackwardGPU(grad_out_feat, in_key, out_key, manager):
    _check_feat("grad_out_feat", grad_out_feat)
    rows = manager._pruning_rows(in_key, out_key)
    grad_in = torch.zeros((manager.size(in_key), grad_out_feat.shape[1]), dtype=grad_out_feat.dtype,
                          device=grad_out_feat.device)
    grad_in.index_copy_(0, rows.long(), grad_out_feat)
    return grad_in
