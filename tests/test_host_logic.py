"""Host-side logic that needs no GPU: keys, kernel generator, argument checks, loud failure on
CPU tensors (there is no CPU fallback in the product)."""
import pytest
import torch

import minkowskiengine_amd as ME
from minkowskiengine_amd import backend as MEB


def test_coordinate_map_key_semantics():  # src/coordinate_map_key.hpp:44-157
    unset = ME.CoordinateMapKey(4)
    assert not unset.is_key_set() and unset.get_coordinate_size() == 4
    with pytest.raises(RuntimeError):
        unset.get_key()
    k1 = ME.CoordinateMapKey([1, 1, 1], "")
    k2 = ME.CoordinateMapKey([1, 1, 1], "")
    k3 = ME.CoordinateMapKey([2, 2, 2], "a")
    assert k1 == k2 and k1 != k3 and not (unset == unset)
    assert k1.get_key() == ([1, 1, 1], "") and k3.get_tensor_stride() == [2, 2, 2]
    unset.set_key([2, 2, 2], "a")
    assert unset == k3 and hash(unset) == hash(k3)
    with pytest.raises(RuntimeError):
        ME.CoordinateMapKey(3).set_key([1, 1, 1], "")
    assert "coordinate map key:[2, 2, 2]:a" == repr(k3)


def test_kernel_generator():  # MinkowskiKernelGenerator.py:245-345
    kg = ME.KernelGenerator(kernel_size=3, stride=2, dilation=1, dimension=3)
    assert kg.kernel_volume == 27 and kg.kernel_stride == [2, 2, 2] and not kg.requires_strided_coordinates
    assert ME.KernelGenerator(kernel_size=[3, 2, 2], dimension=3).kernel_volume == 12
    assert ME.KernelGenerator(kernel_size=3, region_type=ME.RegionType.HYPER_CROSS, dimension=3).kernel_volume == 7
    assert ME.KernelGenerator(kernel_size=1, dimension=3).requires_strided_coordinates


def test_convolution_module_shapes():  # MinkowskiConvolution.py:264-279, 332-340
    conv = ME.MinkowskiConvolution(8, 16, kernel_size=3, dimension=3, bias=True)
    assert tuple(conv.kernel.shape) == (27, 8, 16) and tuple(conv.bias.shape) == (1, 16)
    bound = 1.0 / (8 * 27) ** 0.5
    assert float(conv.kernel.detach().abs().max()) <= bound
    assert ME.MinkowskiConvolution(8, 16, kernel_size=1, dimension=3).use_mm
    assert not ME.MinkowskiConvolution(8, 16, kernel_size=1, stride=2, dimension=3).use_mm
    t = ME.MinkowskiConvolutionTranspose(8, 16, kernel_size=2, stride=2, dimension=3)
    assert t.is_transpose and tuple(t.kernel.shape) == (8, 8, 16)
    assert "MinkowskiConvolution(in=8, out=16, kernel_size=[3, 3, 3], stride=[1, 1, 1], dilation=[1, 1, 1])" == repr(
        ME.MinkowskiConvolution(8, 16, kernel_size=3, dimension=3))


def test_cpu_tensors_are_rejected_loudly():
    coords = torch.IntTensor([[0, 0, 0, 0], [0, 1, 0, 0]])
    feats = torch.rand(2, 4)
    with pytest.raises(RuntimeError):
        ME.SparseTensor(feats, coords)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    with pytest.raises(RuntimeError):
        mgr.insert_and_map(coords, [1, 1, 1], "")
    with pytest.raises(RuntimeError):
        mgr.insert_and_map(coords.float(), [1, 1, 1], "")
    with pytest.raises(ValueError):
        ME.get_minkowski_function("ConvolutionForward", feats)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from minkowskiengine_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_batched_coordinates():
    a = torch.IntTensor([[1, 2], [3, 4]])
    b = torch.IntTensor([[5, 6]])
    bc = ME.utils.batched_coordinates([a, b])
    assert bc.tolist() == [[0, 1, 2], [0, 3, 4], [1, 5, 6]] and bc.dtype == torch.int32


def test_enums_match_reference_names():  # pybind/extern.hpp:669-741
    assert int(ME.RegionType.HYPER_CUBE) == 0 and int(ME.RegionType.HYPER_CROSS) == 1
    assert hasattr(ME.ConvolutionMode, "DIRECT_GEMM") and hasattr(ME.MinkowskiAlgorithm, "SPEED_OPTIMIZED")
    assert hasattr(MEB, "ConvolutionForwardGPU") and hasattr(MEB, "ConvolutionTransposeBackwardGPU")
    assert MEB.is_cuda_available()


def test_tile_order_policy(monkeypatch):
    """KernelMapGPU._tile_order: the matrix-bound launch family (fp32 on the bf16 matrix pipe) takes spatially compact
    tiles, the others row tiles while the neighbour table is at most 32 MiB; ME_AMD_TILE_ORDER overrides both; a map
    without coordinate maps (nothing to take a spatial order from) falls back to row tiles (order None)."""
    offs = torch.zeros(28, dtype=torch.int64)
    pairs = torch.zeros(0, dtype=torch.int32)
    small = MEB.KernelMapGPU(27, 100000, 100000, [0] * 28, offs, pairs, pairs)
    big = MEB.KernelMapGPU(81, 400000, 400000, [0] * 82, torch.zeros(82, dtype=torch.int64), pairs, pairs)
    monkeypatch.setattr(MEB, "_TILE_ORDER", "auto")
    assert small._tile_order("out") == "rows" and small._tile_order("in") == "rows"
    assert small._tile_order("out", matrix_bound=True) == "spatial"
    assert big._tile_order("out") == "spatial"                      # 81 x 400k x 4 B = 130 MB table
    monkeypatch.setattr(MEB, "_TILE_ORDER", "rows")
    assert small._tile_order("out", matrix_bound=True) == "rows" and big._tile_order("out") == "rows"
    monkeypatch.setattr(MEB, "_TILE_ORDER", "auto")
    monkeypatch.setattr(MEB, "_SPATIAL_TILES", False)
    assert small.order("out", "spatial") is None and small.order("out") is None


def test_conv_statistics_registry_belongs_to_one_tensor_at_one_version():
    """backend._BN_PARTIALS hands a convolution's epilogue statistics to the batch norm that follows: an entry is valid
    for the very tensor object it was registered for, at the version it had; it is consumed by the first take, missed by
    a modified / other / re-allocated tensor, and dropped when its tensor is collected (host logic: CPU tensors do)."""
    import torch
    from minkowskiengine_amd import backend as MEB
    MEB._BN_PARTIALS.clear()
    y = torch.zeros(10, 8)
    part = torch.zeros(2, 1, 8)
    MEB._bn_partials_put(y, part, 16)
    assert MEB._bn_partials_take(torch.zeros(10, 8)) is None           # another tensor
    got = MEB._bn_partials_take(y)
    assert got is not None and got[0] is part and got[1] == 16
    assert MEB._bn_partials_take(y) is None                            # consumed
    MEB._bn_partials_put(y, part, 16)
    y.add_(1.0)                                                        # modified in place: version counter moved
    assert MEB._bn_partials_take(y) is None and not MEB._BN_PARTIALS
    MEB._bn_partials_put(y, part, 16)
    assert MEB._bn_partials_take(y[:, :4]) is None                     # a view of it (same storage address) is not it
    z = torch.zeros(10, 8)
    MEB._bn_partials_put(z, part, 16)
    assert len(MEB._BN_PARTIALS) == 1
    del z                                                              # never normalised: the entry goes with it
    assert not MEB._BN_PARTIALS
    MEB.conv_bn_stats_hint(True)
    assert MEB._BN_STATS_HINT[0] is True
    MEB.conv_bn_stats_hint(False)


def test_bf16_tile_shape_policy_is_the_documented_one():
    """conv_variant_bf16 through its host-side queries (no GPU needed: the policy functions are plain host code): the
    source-channel chunk per layer shape (docs/HISTORY.md 10.4: 256-channel chunks where a launch is at most two 128-column slabs
    wide, 96 for 96 / 192-channel layers, else 32 / 64 / 128), the statistics epilogue for every default shape, packed
    image sizes padded to the chunk, tile heights inside the ABI's range, and the library version of the header."""
    import ctypes
    import os
    import re
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    want_kc = {(256, 256): 256, (96, 96): 96, (64, 128): 64, (128, 64): 128, (3, 32): 32, (8, 32): 32, (384, 256): 128,
               (256, 384): 128, (192, 128): 96, (128, 96): 128, (32, 32): 32, (20, 24): 32, (128, 128): 128, (64, 64): 64}
    for (cs, cd), kc in want_kc.items():
        assert lib.me_conv_pack_chunk_bf16(cs, cd) == kc, (cs, cd)
        assert lib.me_conv_stats_supported_bf16(cs, cd) == 1, (cs, cd)
        pad = lambda v, m: -(-v // m) * m
        assert lib.me_conv_packed_weight_elems_bf16(27, cs, cd) == 27 * pad(cs, kc) * pad(cd, 16)
        t, g = ctypes.c_int32(), ctypes.c_int32()
        assert lib.me_conv_plan_config_bf16(21176, 27, 223174, cs, cd, ctypes.byref(t), ctypes.byref(g)) == 0
        assert 16 <= t.value <= 256 and 1 <= g.value <= 4
    assert lib.me_conv_pack_chunk_bf16(0, 8) == 0 and lib.me_conv_stats_supported_bf16(0, 8) == 0
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "me_amd.h")).read()
    latest = max(int(v) for v in re.findall(r"\((\d{3})\)\s+round", header))
    assert lib.me_version() == latest


def test_recipe_for_a_new_scene_is_the_longest_one_around():
    """Map prefetch (round 4): a new scene's manager replays the LONGEST request log among the latest managers still alive
    and the one the last destroyed manager left behind — with a loader thread the directly preceding manager has not run
    its step yet, and a scene's manager usually dies right after its step, when its log is complete."""
    import gc
    from minkowskiengine_amd import coordinate_manager as CM

    class FakeNative:
        def __init__(self, log):
            self.log = list(log)

        def recipe(self):
            return list(self.log)

    class FakeManager:
        def __init__(self, log, D=3, native=True, tag=""):
            self._manager, self.D, self._native, self.replayed, self._tag = FakeNative(log), D, native, None, tag

        def recipe(self):
            return self._manager.recipe()

        def prefetch(self, recipe):
            self.replayed = list(recipe)
            self._manager.log = list(recipe)      # (a replay logs what it serves)
            return len(recipe)

    old_flag, old_recent, old_pub = CM._map_prefetch, list(CM._recent_managers), dict(CM._published_recipes)
    try:
        CM._recent_managers.clear()
        CM._published_recipes.clear()
        CM.set_map_prefetch(True)
        full = [f"conv_cfg;{i}" for i in range(10)]
        a = FakeManager([])                       # scene 1: built before any log exists
        CM._prefetch_from_previous(a)
        assert a.replayed is None
        a._manager.log = list(full)               # ... its step fills its log
        b = FakeManager([])                       # scene 2 was built by the loader BEFORE that: empty log
        CM._recent_managers.append(__import__("weakref").ref(b))
        c = FakeManager([])
        CM._prefetch_from_previous(c)             # scene 3: the longest log around is scene 1's
        assert c.replayed == full
        other = FakeManager(["x"] * 20, D=4)      # another dimension's log is not a candidate
        CM._prefetch_from_previous(other)
        d = FakeManager([])
        CM._prefetch_from_previous(d)
        assert d.replayed == full
        # a destroyed manager leaves its log behind (CoordinateManager.__del__ on a stand-in)
        CM._recent_managers.clear()
        CM.CoordinateManager.__del__(a)
        del a, b, c, d
        gc.collect()
        e = FakeManager([])
        CM._prefetch_from_previous(e)
        assert e.replayed == full
        # shorter logs replace the published one only after _PUBLISH_PATIENCE of them in a row (the network changed)
        short = FakeManager(["conv_cfg;0"])
        for _ in range(CM._PUBLISH_PATIENCE - 1):
            CM.CoordinateManager.__del__(short)
        assert CM._published_recipes[(3, True, "")][0] == full
        CM.CoordinateManager.__del__(short)
        assert CM._published_recipes[(3, True, "")][0] == ["conv_cfg;0"]
        # tags keep the logs of two networks apart (map_prefetch_tag): a tagged scene sees neither the untagged published
        # log nor an untagged live manager, and publishes under its own key
        live = FakeManager(full)
        CM._recent_managers.append(__import__("weakref").ref(live))
        with CM.map_prefetch_tag("eval"):
            assert CM._prefetch_tag.value == "eval"
        assert getattr(CM._prefetch_tag, "value", "") == ""
        t1 = FakeManager([], tag="eval")
        CM._prefetch_from_previous(t1)
        assert t1.replayed is None
        t1._manager.log = ["kernel_map;e"]
        CM.CoordinateManager.__del__(t1)
        assert CM._published_recipes[(3, True, "eval")][0] == ["kernel_map;e"]
        t2 = FakeManager([], tag="eval")
        CM._prefetch_from_previous(t2)
        assert t2.replayed == ["kernel_map;e"]
        u = FakeManager([])
        CM._prefetch_from_previous(u)
        assert u.replayed == full                 # (the untagged live manager's log; not the tagged ones)
        CM._recent_managers.clear()
        CM.set_map_prefetch(False)                # off: nothing is replayed, nothing published
        CM._published_recipes.clear()
        CM.CoordinateManager.__del__(FakeManager(full))
        assert not CM._published_recipes
        f = FakeManager([])
        CM._prefetch_from_previous(f)
        assert f.replayed is None
    finally:
        CM._map_prefetch = old_flag
        CM._recent_managers.clear()
        CM._recent_managers.extend(old_recent)
        CM._published_recipes.clear()
        CM._published_recipes.update(old_pub)


def test_stem_policy_keeps_32_bit_table_offsets_in_range():
    """csrc/conv_stem.hip addresses the neighbour table with 32-bit byte offsets, look-ahead included (the walk counts up
    to offset volume + 42 before it clamps): the policy must refuse maps where (volume + 43) * n_tgt * 4 reaches 2^32 —
    a pure host function, no GPU needed"""
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    assert lib.me_conv_stem_use_bf16(200000, 125, 8, 32) == 1
    for volume in (8, 27, 125):
        limit = (1 << 30) // (volume + 43)
        assert lib.me_conv_stem_use_bf16(limit - 1, volume, 8, 32) == 1, volume
        assert lib.me_conv_stem_use_bf16(limit + 1, volume, 8, 32) == 0, volume
        assert (volume + 42) * (limit - 1) * 4 + (limit - 1) * 4 < (1 << 32)
    assert lib.me_conv_stem_use_bf16(200000, 125, 8, 128) == 0 and lib.me_conv_stem_use_bf16(200000, 125, 16, 32) == 0
    assert lib.me_conv_stem_tile_rows() == 256


def test_halo_plan_waits_for_the_second_launch_on_a_side(monkeypatch):
    """backend._halo_launch_cfg: under the policy (me_conv_halo_min_uses() == 2) the first LAUNCH on a kernel-map side
    stays on the tile-plan kernel, a recipe replay (count=False) never counts, the second launch gets the plan; forced
    mode (min uses 1) gets it at once.  Host logic only: the library is a stand-in, the plan is already in the store."""
    class FakeLib:
        def __init__(self, min_uses):
            self.min_uses, self.asked = min_uses, 0

        def me_conv_halo_use_bf16(self, *a):
            return 1

        def me_conv_halo_min_uses(self):
            return self.min_uses

        def me_conv_halo_config_bf16(self, n_tgt, volume, n_pairs, c_src, c_dst, t, cap):
            self.asked += 1
            t._obj.value, cap._obj.value = 128, 383
            return 1

    class FakeMap:
        volume, n_pairs, _recipe = 27, 10 ** 6, None

        def __init__(self):
            self._launch_cache, self._store = {}, {"halo_out_128_383": "PLAN"}

        def _name(self, kind, target):
            return kind + "_" + target

    lib = FakeLib(2)
    monkeypatch.setattr(MEB._lib, "load", lambda: lib)
    km = FakeMap()
    assert MEB._halo_launch_cfg(km, "out", 80000, 192, 128, count=False) is None      # a replay asks: no launch follows
    assert MEB._halo_launch_cfg(km, "out", 80000, 192, 128) is None                   # first launch: tile-plan kernel
    assert MEB._halo_launch_cfg(km, "out", 80000, 192, 128, count=False) is None
    assert lib.asked == 0
    assert MEB._halo_launch_cfg(km, "out", 80000, 192, 128) == "PLAN"                 # second launch: the plan
    assert MEB._halo_launch_cfg(km, "out", 80000, 192, 128, count=False) == "PLAN"    # ... and replays see it from now on
    assert MEB._halo_launch_cfg(km, "in", 80000, 192, 128) is None                    # the other side counts for itself
    lib1 = FakeLib(1)
    monkeypatch.setattr(MEB._lib, "load", lambda: lib1)
    km1 = FakeMap()
    assert MEB._halo_launch_cfg(km1, "out", 80000, 192, 128) == "PLAN"                # forced (ME_AMD_HALO=1): at once


def test_targeted_invalidation_touches_only_the_named_weights():
    """backend.invalidate_packed_weights(ptrs): the images of the weights at those storage addresses go stale, the others —
    a frozen / teacher network — stay valid; without an argument the global epoch moves (every image stale)"""
    class Entry:
        def __init__(self, ptr):
            self.ptr, self.epoch = ptr, MEB._PACK_EPOCH[0]

    class Packer:
        def __init__(self, ptrs):
            self.entries = {(p, 0, False): Entry(p) for p in ptrs}

    saved = dict(MEB._PACKERS)
    try:
        MEB._PACKERS.clear()
        MEB._PACKERS[0] = Packer([100, 200, 300])
        MEB.invalidate_packed_weights({200, 999})
        ep = {e.ptr: e.epoch for e in MEB._PACKERS[0].entries.values()}
        assert ep[200] == -1 and ep[100] == ep[300] == MEB._PACK_EPOCH[0]
        before = MEB._PACK_EPOCH[0]
        MEB.invalidate_packed_weights()
        assert MEB._PACK_EPOCH[0] == before + 1 and all(e.epoch != MEB._PACK_EPOCH[0] for e in MEB._PACKERS[0].entries.values())
    finally:
        MEB._PACKERS.clear()
        MEB._PACKERS.update(saved)


def test_invalidation_by_storage_range_and_master_copy_fallback():
    """host.invalidate_packed_weights(params) (the optimizer-step hook's call): an image whose weight is an OFFSET
    VIEW of a stepping parameter is matched by its address range; an optimizer that holds none of the cached weights — fp32
    master copies written back through `.data.copy_` — makes EVERY image stale (global epoch), as before targeted
    invalidation existed."""
    import torch
    from minkowskiengine_amd import host

    class Entry:
        def __init__(self, ptr):
            self.ptr, self.epoch = ptr, MEB._PACK_EPOCH[0]

    class Packer:
        def __init__(self, ptrs):
            self.entries = {(p, 0, False): Entry(p) for p in ptrs}

    saved = dict(MEB._PACKERS)
    try:
        flat = torch.zeros(64)
        view = flat[16:32]                                   # the model's kernel: a view into a flat parameter
        other = torch.zeros(8)
        MEB._PACKERS.clear()
        MEB._PACKERS[0] = Packer([view.data_ptr(), other.data_ptr()])
        before = MEB._PACK_EPOCH[0]
        host.invalidate_packed_weights([flat])               # the optimizer steps `flat`
        ep = {e.ptr: e.epoch for e in MEB._PACKERS[0].entries.values()}
        assert ep[view.data_ptr()] == -1 and ep[other.data_ptr()] == before and MEB._PACK_EPOCH[0] == before
        # master copies: nothing cached lives in the optimizer's tensors -> everything stale
        for e in MEB._PACKERS[0].entries.values():
            e.epoch = MEB._PACK_EPOCH[0]
        master = torch.zeros(64)
        host.invalidate_packed_weights([master])
        assert MEB._PACK_EPOCH[0] == before + 1
        # an empty cache: nothing to do, no epoch churn
        MEB._PACKERS.clear()
        e0 = MEB._PACK_EPOCH[0]
        host.invalidate_packed_weights([master])
        n = host.native_module()
        if n is None:
            assert MEB._PACK_EPOCH[0] == e0
    finally:
        MEB._PACKERS.clear()
        MEB._PACKERS.update(saved)
