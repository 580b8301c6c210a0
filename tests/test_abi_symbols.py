"""The C-ABI library loads (no GPU needed) and exports every symbol include/me_amd.h declares;
the ctypes prototypes cover exactly that set."""
import ctypes
import os
import re

from minkowskiengine_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(path=("include", "me_amd.h")):
    text = open(os.path.join(ROOT, *path)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(me_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ("me_coords_insert_and_map", "me_kernel_map_probe", "me_kernel_map_compact", "me_plan_build",
              "me_conv_target_f32", "me_conv_wgrad_f32"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/me_amd.h but not exported"


def test_ctypes_prototypes_match_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.me_version() >= 100


def test_debug_hooks_are_not_part_of_the_public_header():
    """me_debug_* (process-global kernel-selection switches for tests/ and scripts/) live in csrc/me_amd_debug.h; the
    drop-in boundary does not declare them, and the default build refuses result-invalidating variants."""
    assert not [s for s in declared_symbols() if s.startswith("me_debug")]
    assert sorted(_lib.DEBUG_SIGNATURES) == declared_symbols(("minkowskiengine_amd", "csrc", "me_amd_debug.h"))
    lib = _lib.load()
    if not lib.me_debug_variants_compiled():
        for v in (16, 33, 38, 256, 257, 3000):
            assert lib.me_debug_set_conv_variant(v) != 0
        assert lib.me_debug_set_conv_variant(0) == 0


def test_host_only_entry_points():
    lib = _lib.load()
    assert lib.me_hash_capacity(0) == 64
    assert lib.me_hash_capacity(100000) == 262144
    assert lib.me_plan_num_tiles(257, 128) == 3 and lib.me_plan_num_tiles(257, 131) == 2
    rg = _lib.make_region(4, 0, [3, 3, 3], [1, 1, 1], [1, 1, 1])
    assert lib.me_region_volume(ctypes.byref(rg)) == 27
    rg = _lib.make_region(4, 1, [3, 3, 5], [1, 1, 1], [1, 1, 1])
    assert lib.me_region_volume(ctypes.byref(rg)) == 9
    rg = _lib.make_region(5, 0, [3, 3, 3, 3], [1] * 4, [1] * 4)
    assert lib.me_region_volume(ctypes.byref(rg)) == 81
    # groups: one per 16 pairs + one partial group per non-empty item + 1, + the 4 groups (64 slots) of the index
    # window that the convolution kernels read past the last batch without clamping
    assert lib.me_plan_max_groups(1000, 27, 8000, 128) == 8000 // 16 + 8 * 27 + 1 + 4
    assert lib.me_plan_max_groups(1000, 27, 50, 128) == 50 // 16 + 50 + 1 + 4
    koffs = (ctypes.c_int64 * 4)(0, 10, 10, 5000)
    # (ranges + volume) slots of one 64 x 16 register image each: 5000 pairs -> 78 ranges of >= 64 pairs
    assert lib.me_conv_wgrad_workspace_bytes(koffs, 3, 8, 16) == (5000 // 64 + 3) * 64 * 16 * 4


def test_plan_config_fills_the_chip():
    """Host-only heuristic: tiles x slabs should sit just below a multiple of the resident slots and the
    workgroup must fit the 160 KiB LDS."""
    from minkowskiengine_amd import _lib
    lib = _lib.load()

    def cfg(*args):
        t, g = ctypes.c_int32(0), ctypes.c_int32(0)
        assert lib.me_conv_plan_config(*args, ctypes.byref(t), ctypes.byref(g)) == 0
        return t.value, g.value

    t, g = cfg(100000, 27, 834914, 64, 128)
    assert 16 <= t <= 256 and 1 <= g <= 4
    lds = (t + 1) * (64 + 4) * 4 + g * 16 * (68 * 4 + 4)
    occ = min(3, (160 * 1024) // lds)
    assert occ >= 1
    items = -(-100000 // t) * 2
    slots = 256 * occ
    assert items % slots == 0 or items % slots > 0.9 * slots, (t, items)
    for args in [(1, 27, 1, 4, 4), (4977, 27, 52353, 256, 256), (200000, 125, 889332, 3, 32),
                 (400000, 81, 1837616, 32, 64), (100000, 27, 834914, 128, 64), (20000, 8, 20000, 96, 96)]:
        t, g = cfg(*args)
        assert 16 <= t <= 256 and 1 <= g <= 4
