"""GPU parity of the coordinate path (hash insert / find / stride / kernel map / tile plan) against the
oracle, called through the C ABI (via the backend module).  Index maps must be bit-exact."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from helpers import make_cloud, row_mapping

pytestmark = pytest.mark.gpu


def _mgr():
    """(operator module, manager) of the host layer in charge: the Python twin unless a test takes `host_layer`"""
    from minkowskiengine_amd import host
    B = host.backend()
    return B, B.CoordinateMapManagerGPU_c10()


@pytest.mark.parametrize("n,extent,D,dup", [(1, 4, 3, 0), (5000, 12, 3, 900), (20000, 40, 3, 0), (3000, 6, 4, 500),
                                             (2000, 30, 2, 100), (1000, 200, 1, 300), (4000, 6, 5, 100)])
def test_insert_and_map_bit_exact(device, host_layer, n, extent, D, dup):
    coords = make_cloud(n, extent, D, seed=n + D, batch=2, dup=dup, negative=True)
    MEB, mgr = _mgr()
    key, (um, inv) = mgr.insert_and_map(coords.to(device), [1] * D, "")
    um_o, inv_o = O.insert_and_map(coords.numpy())
    assert um.dtype == torch.int64 and inv.dtype == torch.int64
    assert np.array_equal(um.cpu().numpy(), um_o), "unique_map differs from the oracle"
    assert np.array_equal(inv.cpu().numpy(), inv_o), "inverse_map differs from the oracle"
    got = mgr.get_coordinates(key).cpu().numpy()
    assert np.array_equal(got, coords.numpy()[um_o]), "stored coordinates differ / wrong row order"
    assert mgr.size(key) == len(um_o)
    # round trip of the reference test (tests/python/coordinate_manager.py:33-58)
    assert np.array_equal(coords.numpy(), got[inv.cpu().numpy()])


@pytest.mark.parametrize("n,extent,D,dup", [(1, 4, 3, 0), (70, 6, 3, 40), (5000, 12, 3, 900), (60000, 60, 3, 30000),
                                             (300000, 90, 3, 5000), (9000, 8, 4, 4000), (4097, 40, 2, 0)])
def test_fused_insert_equals_the_resolve_scan_finalize_pipeline(device, n, extent, D, dup):
    """Round 6: insert = fill + k_insert + k_insert_flags + k_insert_emit (ranks rebuilt from ballots, group prefixes and
    block counts; bounding box from per-block boxes) instead of resolve + one-block scan + finalize + bbox.  Same maps,
    same table, same bounding box, bit for bit — one block, many blocks, a partial last block, heavy duplication."""
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    coords = make_cloud(n, extent, D, seed=n + D, batch=2, dup=dup, negative=True).to(device)
    res = []
    for fused in (1, 0):
        lib.me_debug_set_insert_fused(fused)
        try:
            MEB, mgr = _mgr()
            key, (um, inv) = mgr.insert_and_map(coords, [1] * D, "")
            cm = mgr._maps[mgr._k(key)]
            # (the slot a key lands in depends on the race of the claims: the table is compared through its lookups)
            rows = torch.empty(coords.shape[0], dtype=torch.int32, device=device)
            with torch.cuda.device(device):
                _lib.check(lib.me_coords_find(cm.table.data_ptr(), cm.capacity, cm.coords.data_ptr(), D + 1,
                                              coords.data_ptr(), coords.shape[0], rows.data_ptr(), None))
            torch.cuda.synchronize()
            res.append((um.clone(), inv.clone(), mgr.get_coordinates(key).clone(), rows, tuple(cm.bbox or ())))
        finally:
            lib.me_debug_set_insert_fused(1)
    for a, b in zip(res[0][:4], res[1][:4]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][3].long(), res[0][1]), "a lookup of every input row must give its inverse-map entry"
    assert res[0][4] == res[1][4] and len(res[0][4]) == 2 * (D + 1)
    um_o, inv_o = O.insert_and_map(coords.cpu().numpy())
    assert np.array_equal(res[0][0].cpu().numpy(), um_o) and np.array_equal(res[0][1].cpu().numpy(), inv_o)


@pytest.mark.parametrize("n,extent,D,batch,ts,with_bbox", [(1, 4, 3, 1, 1, True), (5000, 12, 3, 1, 1, True), (60000, 70, 3, 2, 1, True),
                                                            (20000, 300, 3, 3, 2, True), (9000, 8, 4, 2, 1, True),
                                                            (4097, 1000, 2, 1, 4, True), (7000, 40, 3, 2, 1, False),
                                                            (3000, 2000, 1, 1, 1, True)])
def test_zorder_equals_the_stable_argsort_of_the_spatial_keys(device, n, extent, D, batch, ts, with_bbox):
    """me_coords_zorder (round 6: the library's own LSD radix sort, only over the key bytes that can differ inside the
    bounding box) = numpy's STABLE argsort of me_coords_spatial_keys' 64-bit keys, bit for bit — one pass, several,
    negative coordinates, several scenes (the batch byte), tensor strides, no bounding box (all eight bytes)."""
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    coords = make_cloud(n, extent, D, seed=n + D, batch=batch, negative=True)
    coords[:, 1:] *= ts
    c = coords.to(device)
    m = c.shape[0]
    tsa = (ctypes.c_int32 * D)(*([ts] * D))
    keys = torch.empty(m, dtype=torch.int64, device=device)
    order = torch.empty(m, dtype=torch.int32, device=device)
    ws = torch.empty(int(lib.me_coords_zorder_workspace_bytes(m)), dtype=torch.uint8, device=device)
    cn = coords.numpy()
    bb = None
    if with_bbox:
        bb = (ctypes.c_int32 * (2 * (D + 1)))(*([int(v) for v in cn.min(0)] + [int(v) for v in cn.max(0)]))
    with torch.cuda.device(device):
        _lib.check(lib.me_coords_spatial_keys(c.data_ptr(), m, D + 1, tsa, keys.data_ptr(), None))
        _lib.check(lib.me_coords_zorder(c.data_ptr(), m, D + 1, tsa, bb, order.data_ptr(), ws.data_ptr(), ws.numel(), None))
    torch.cuda.synchronize()
    want = np.argsort(keys.cpu().numpy(), kind="stable")
    assert np.array_equal(order.cpu().numpy(), want.astype(np.int32))


def test_insert_all_duplicates_and_empty(device, host_layer):
    MEB, mgr = _mgr()
    coords = torch.IntTensor([[0, 1, 2, 3]] * 257).to(device)
    key, (um, inv) = mgr.insert_and_map(coords, [1, 1, 1], "")
    assert um.tolist() == [0] and inv.tolist() == [0] * 257 and mgr.size(key) == 1
    key2, (um2, inv2) = mgr.insert_and_map(torch.zeros((0, 4), dtype=torch.int32, device=device), [1, 1, 1], "e")
    assert um2.numel() == 0 and inv2.numel() == 0 and mgr.size(key2) == 0


def test_key_collision_gets_random_suffix(device, host_layer):
    MEB, mgr = _mgr()
    c = make_cloud(100, 8, 3).to(device)
    k1, _ = mgr.insert_and_map(c, [1, 1, 1], "")
    k2, _ = mgr.insert_and_map(c, [1, 1, 1], "")
    assert k1 != k2 and k1.get_key()[1] == "" and len(k2.get_key()[1]) == 5


def test_find(device):
    from minkowskiengine_amd import _lib
    MEB, mgr = _mgr()
    coords = make_cloud(3000, 14, 3, seed=5, negative=True)
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    cmap = mgr._get(key)
    q = make_cloud(4000, 18, 3, seed=6, negative=True).to(device)
    rows = torch.empty(q.shape[0], dtype=torch.int32, device=device)
    lib = _lib.load()
    _lib.check(lib.me_coords_find(ctypes.c_void_p(cmap.table.data_ptr()), cmap.capacity,
                                  ctypes.c_void_p(cmap.coords.data_ptr()), 4, ctypes.c_void_p(q.data_ptr()),
                                  q.shape[0], ctypes.c_void_p(rows.data_ptr()),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert np.array_equal(rows.cpu().numpy(), O.find(coords.numpy(), q.cpu().numpy()))


@pytest.mark.parametrize("D,stride", [(3, 2), (3, [2, 1, 4]), (2, 3), (4, 2), (1, 2)])
def test_stride_map(device, host_layer, D, stride):
    coords = make_cloud(4000, 20, D, seed=11 + D, batch=2, negative=True)
    MEB, mgr = _mgr()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    st = [stride] * D if isinstance(stride, int) else stride
    okey = mgr.stride(key, st)
    assert okey.get_tensor_stride() == st
    got = mgr.get_coordinates(okey).cpu().numpy()
    exp, _ = O.stride_map(coords.numpy(), st)
    # first-occurrence order in input-row order: identical to the oracle's (deterministic) order
    assert np.array_equal(got, exp)
    assert mgr.stride(key, st) == okey                            # cached, not rebuilt
    if hasattr(mgr, "_maps"):
        assert len(mgr._maps) == 2
    assert mgr.stride(key, [1] * D) == key                        # all-ones stride reuses the map


def test_negative_coordinate_stride(device, host_layer):  # tests/python/coordinate_manager.py:183-200
    MEB, mgr = _mgr()
    coords = torch.IntTensor([[0, -3], [0, -2], [0, -1], [0, 0], [0, 1], [0, 2], [0, 3]]).to(device)
    key, _ = mgr.insert_and_map(coords, [1], "")
    out = mgr.get_coordinates(mgr.stride(key, [2])).cpu().numpy()
    assert sorted(out[:, 1].tolist()) == [-4, -2, 0, 2]


KMAP_CASES = [
    # n, extent, D, kernel_size, stride, dilation, region
    (6000, 20, 3, 3, 1, 1, 0),
    (6000, 60, 3, 3, 1, 1, 0),          # sparse
    (5000, 18, 3, 2, 2, 1, 0),          # MinkUNet down conv
    (4000, 16, 3, 3, 2, 1, 0),
    (3000, 14, 3, 5, 1, 1, 0),          # K = 125
    (3000, 14, 3, 3, 1, 2, 0),          # dilation
    (3000, 8, 4, 3, 1, 1, 0),           # 4-D, K = 81
    (3000, 30, 2, [3, 2], 1, 1, 0),     # mixed odd/even
    (3000, 14, 3, [3, 2, 2], 1, 1, 0),
    (3000, 14, 3, 3, 1, 1, 1),          # HYPER_CROSS
    (130, 6, 3, 3, 1, 1, 0),            # barely more than one tile
    (1, 2, 3, 3, 1, 1, 0),
]


@pytest.mark.parametrize("n,extent,D,ks,stride,dil,region", KMAP_CASES)
def test_kernel_map_of_the_manager_interface_vs_oracle(device, host_layer, n, extent, D, ks, stride, dil, region):
    """The same cases through the reference's manager interface alone (insert_and_map, stride, get_coordinates,
    kernel_map -> {k: int32 [2, n_k]}; pybind/extern.hpp:767-806) on BOTH host layers — what a user of the shipped
    native module sees — against the oracle: pair sets identical per offset."""
    coords = make_cloud(n, extent, D, seed=n + D, batch=2 if n > 100 else 1, negative=True)
    B, mgr = _mgr()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    ksl = [ks] * D if isinstance(ks, int) else ks
    okey = mgr.stride(key, [stride] * D, "")
    out_c = mgr.get_coordinates(okey).cpu().numpy()
    assert np.array_equal(out_c, O.stride_map(coords.numpy(), [stride] * D)[0] if stride != 1 else coords.numpy())
    _, km_o = O.kernel_map(coords.numpy(), out_c, O.make_region(D, ksl, dil, 1, region))
    d = mgr.kernel_map(key, okey, ksl, [stride] * D, [dil] * D, B.RegionType(region), torch.empty(0, dtype=torch.int32),
                       False, False)
    for k, v in d.items():
        assert v.dtype == torch.int32 and v.shape[0] == 2
    O.assert_same_kernel_map(d, km_o)
    assert mgr.size(okey) == len(out_c)


@pytest.mark.parametrize("n,extent,D,ks,stride,dil,region", KMAP_CASES)
def test_kernel_map_pair_sets_identical(device, n, extent, D, ks, stride, dil, region):
    coords = make_cloud(n, extent, D, seed=n + D, batch=2 if n > 100 else 1, negative=True)
    MEB, mgr = _mgr()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    ksl = [ks] * D if isinstance(ks, int) else ks
    okey = mgr.stride(key, [stride] * D)
    km = mgr._kernel_map(key, okey, ksl, [stride] * D, [dil] * D, MEB.RegionType(region), None, False, False)
    in_c = coords.numpy()
    out_c = mgr.get_coordinates(okey).cpu().numpy()
    nbr_o, km_o = O.kernel_map(in_c, out_c, O.make_region(D, ksl, dil, 1, region))
    # dict API, reference format (int32 [2, n_k], only non-empty offsets)
    d = mgr.kernel_map(key, okey, ksl, [stride] * D, [dil] * D, MEB.RegionType(region), None, False, False)
    for k, v in d.items():
        assert v.dtype == torch.int32 and v.shape[0] == 2
    O.assert_same_kernel_map(d, km_o)
    # flat-table maps list the pairs of an offset sorted by output row like the oracle: the lists are identical;
    # LDS-bucketed maps list them in supercell order of the output rows (still deterministic, same sets)
    if km._store.get("order_out") is None:
        for k in km_o:
            assert np.array_equal(d[k].cpu().numpy(), km_o[k])
    else:
        pos = km._store["pos_out"].cpu().numpy()
        for k in km_o:
            out_rows = d[k][1].cpu().numpy()
            assert np.all(np.diff(pos[out_rows]) > 0), "pairs of an offset are ordered by output position"
    assert km.n_pairs == sum(v.shape[1] for v in km_o.values())
    # dense neighbour tables
    assert np.array_equal(km.table("out").cpu().numpy()[:, :len(out_c)], nbr_o)
    nbrT = km.table("in").cpu().numpy()
    expT = np.full((km.volume, len(in_c)), -1, np.int32)
    for k, io in km_o.items():
        expT[k, io[0]] = io[1]
    assert np.array_equal(nbrT[:, :len(in_c)], expT)
    # cache hit returns the same object
    assert mgr._kernel_map(key, okey, ksl, [stride] * D, [dil] * D, MEB.RegionType(region), None, False, False) is km


@pytest.mark.parametrize("tile_order", ["rows", "spatial"])
@pytest.mark.parametrize("stride", [2, 1])      # 2: flat-table map (row-space tables); 1: LDS-bucketed map (positions)
@pytest.mark.parametrize("target,T,CAP", [("out", 128, 4), ("in", 128, 4), ("out", 131, 3), ("in", 37, 1),
                                          ("out", 16, 2), ("out", 256, 4)])
def test_tile_plan_covers_every_pair_once(device, target, T, CAP, stride, tile_order, monkeypatch):
    from minkowskiengine_amd import _lib
    from minkowskiengine_amd import backend as MEB0
    monkeypatch.setattr(MEB0, "_SPATIAL_MAPS", True)      # (the automatic choice takes the flat table at this size)
    monkeypatch.setattr(MEB0, "_TILE_ORDER", tile_order)
    coords = make_cloud(5000, 16, 3, seed=21, batch=2, negative=True)
    MEB, mgr = _mgr()
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    okey = mgr.stride(key, [stride] * 3)
    km = mgr._kernel_map(key, okey, [3, 3, 3], [stride] * 3, [1, 1, 1], MEB.RegionType.HYPER_CUBE, None, False, False)
    assert (km._store.get("order_out") is not None) == (stride == 1)
    plan_src, plan_dst, batch_desc, tile_bptr, item_gptr = [t.cpu().numpy() for t in km.plan(target, T, CAP)]
    tbl = km.table(target).cpu().numpy()
    n_tgt = km.n_out if target == "out" else km.n_in
    order = km.order(target)
    order = order.cpu().numpy() if order is not None else np.arange(n_tgt)
    assert sorted(order.tolist()) == list(range(n_tgt)), "the tile order is a permutation of the target rows"
    G = _lib.ME_GROUP_ROWS
    K = km.volume
    n_tiles = (n_tgt + T - 1) // T
    assert tile_bptr[0] == 0 and np.all(np.diff(tile_bptr[:n_tiles + 1]) >= 0)
    # behind the batch pointers: the dispatch order, a permutation of the tiles with (binned) non-increasing work
    perm = tile_bptr[n_tiles + 1:2 * n_tiles + 1]
    assert sorted(perm.tolist()) == list(range(n_tiles))
    work = np.array([item_gptr[(t + 1) * km.volume] - item_gptr[t * km.volume] for t in range(n_tiles)])
    span = max(int(work.max() - work.min()), 1)
    assert np.all(np.diff(work[perm]) <= span / 255 + 1), "heaviest tiles first, up to one bin of the counting sort"
    assert item_gptr[0] == 0 and np.all(np.diff(item_gptr[:n_tiles * K + 1]) >= 0)
    seen = set()
    next_group = 0
    for t in range(n_tiles):
        last_k = -1
        for b in range(tile_bptr[t], tile_bptr[t + 1]):
            g0, y = int(batch_desc[2 * b]), int(batch_desc[2 * b + 1])
            ng, k = y & 255, y >> 8
            assert 1 <= ng <= CAP and k >= last_k, "batches of a tile are sorted by offset, one offset each"
            assert g0 == next_group, "batches tile the group list without gaps"
            assert item_gptr[t * K + k] <= g0 and g0 + ng <= item_gptr[t * K + k + 1]
            next_group = g0 + ng
            last_k = k
            for g in range(g0, g0 + ng):
                for j in range(G):
                    s, d = int(plan_src[g * G + j]), int(plan_dst[g * G + j])
                    if s < 0:
                        assert d == T, "padding slots must point at the dummy row"
                        continue
                    assert 0 <= d < T and t * T + d < n_tgt
                    row = int(order[t * T + d])
                    assert tbl[k, row] == s
                    assert (k, row) not in seen
                    seen.add((k, row))
    assert next_group == item_gptr[n_tiles * K]
    assert len(seen) == int((tbl[:, :n_tgt] >= 0).sum()) == km.n_pairs


def test_plan_build_multi_writes_the_arrays_of_plan_build(device):
    """me_plan_build_multi (ABI 1.4): all plans of a scene in four launches.  A table of jobs over two kernel maps —
    forward and transposed tables, row tiles and a permuted tile order, tile heights 16 ... 256, batches of 1 ... 4 groups,
    K = 27 and K = 8, and a job without target rows in the middle — must leave every job's five arrays exactly as
    me_plan_build leaves them."""
    import ctypes
    from minkowskiengine_amd import _lib
    from minkowskiengine_amd import backend as MEB
    lib = _lib.load()
    coords = make_cloud(5000, 16, 3, seed=5, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    okey = mgr.stride(key, [2, 2, 2])
    km1 = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    km2 = mgr._kernel_map(key, okey, [2] * 3, [2] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(3)
    specs = []   # (table, n_tgt, volume, order, tile_rows, batch_groups, pairs)
    for km, target, T, cap, permuted in ((km1, "out", 128, 4, False), (km1, "in", 37, 1, True), (km2, "out", 16, 2, False),
                                         (None, None, 64, 4, False), (km2, "in", 256, 3, True), (km1, "out", 131, 3, True)):
        if km is None:
            specs.append((torch.zeros(27, 1, dtype=torch.int32, device=device), 0, 27, None, T, cap, 0))
            continue
        tbl = km.table(target).contiguous()
        n_tgt = km.n_out if target == "out" else km.n_in
        assert tbl.shape == (km.volume, n_tgt)
        order = torch.randperm(n_tgt, generator=g).to(torch.int32).to(device) if permuted else None
        specs.append((tbl, n_tgt, km.volume, order, T, cap, km.n_pairs))

    def arrays(n_tgt, volume, pairs, T):
        mg = int(lib.me_plan_max_groups(n_tgt, volume, pairs, T))
        n_tiles = int(lib.me_plan_num_tiles(n_tgt, T))
        mk = lambda n: torch.full((n,), -7, dtype=torch.int32, device=device)
        return [mk(mg * 16), mk(mg * 16), mk(2 * mg), mk(int(lib.me_plan_tile_bptr_elems(n_tgt, T))), mk(n_tiles * volume + 1)]
    stream = torch.cuda.current_stream().cuda_stream
    single, multi = [], []
    jobs = (_lib.MePlanJob * len(specs))()
    for i, (tbl, n_tgt, volume, order, T, cap, pairs) in enumerate(specs):
        a = arrays(n_tgt, volume, pairs, T)
        ws = torch.empty(int(lib.me_plan_workspace_bytes(n_tgt, volume, T)), dtype=torch.uint8, device=device)
        _lib.check(lib.me_plan_build(tbl.data_ptr(), order.data_ptr() if order is not None else None, n_tgt, volume, T, cap,
                                     *[t.data_ptr() for t in a], ws.data_ptr(), ws.numel(), stream))
        single.append(a)
        b = arrays(n_tgt, volume, pairs, T)
        multi.append(b)
        j = jobs[i]
        j.tbl, j.order, j.n_tgt, j.volume = tbl.data_ptr(), (order.data_ptr() if order is not None else None), n_tgt, volume
        j.tile_rows, j.batch_groups = T, cap
        j.plan_src, j.plan_dst, j.batch_desc, j.tile_bptr, j.item_gptr = [t.data_ptr() for t in b]
    total = int(lib.me_plan_jobs_init(ctypes.byref(jobs), len(specs)))
    assert total == sum(-(-s[1] // s[4]) * s[2] for s in specs) and jobs[3].n_items == 0 and jobs[4].item_base == jobs[3].item_base
    raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(device)
    ws = torch.empty(int(lib.me_plan_multi_workspace_bytes(total)), dtype=torch.uint8, device=device)
    _lib.check(lib.me_plan_build_multi(ctypes.byref(jobs), raw.data_ptr(), len(specs), ws.data_ptr(), ws.numel(), stream))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(single, multi)):
        n_tgt, volume, T = specs[i][1], specs[i][2], specs[i][4]
        n_tiles = -(-n_tgt // T)
        groups = int(a[4][n_tiles * volume])            # total groups of the plan
        assert int(b[4][n_tiles * volume]) == groups and int(a[3][n_tiles]) == int(b[3][n_tiles])
        used = [(groups + 4) * 16 if n_tgt else 0, (groups + 4) * 16 if n_tgt else 0, 2 * int(a[3][n_tiles]),
                n_tiles + 1, n_tiles * volume + 1]
        for name, u, v, n in zip(("plan_src", "plan_dst", "batch_desc", "tile_bptr", "item_gptr"), a, b, used):
            assert torch.equal(u[:n], v[:n]), (i, name)
        # the dispatch order behind the batch pointers: a counting sort by binned work whose order inside a bin is
        # arbitrary (LDS atomics) — the same sequence of WORK BINS, both permutations of the tiles
        if n_tiles:
            ig = a[4].cpu().numpy()
            work = np.array([ig[(t + 1) * volume] - ig[t * volume] for t in range(n_tiles)], np.int64)
            span = max(int(work.max() - work.min()), 1)
            bins = 255 - (work - work.min()) * 255 // span
            pa, pb = a[3][n_tiles + 1:2 * n_tiles + 1].cpu().numpy(), b[3][n_tiles + 1:2 * n_tiles + 1].cpu().numpy()
            assert sorted(pa.tolist()) == sorted(pb.tolist()) == list(range(n_tiles)), i
            assert np.array_equal(bins[pa], bins[pb]), i
    # an uninitialised table is refused
    jobs[1].item_base = 12345
    assert lib.me_plan_build_multi(ctypes.byref(jobs), raw.data_ptr(), len(specs), ws.data_ptr(), ws.numel(), stream) != 0


def test_transposed_map_reuses_forward_map(device):  # src/coordinate_map_manager.cpp:763-774
    coords = make_cloud(3000, 14, 3, seed=31, negative=True)
    MEB, mgr = _mgr()
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    okey = mgr.stride(key, [2, 2, 2])
    args = ([2, 2, 2], [2, 2, 2], [1, 1, 1], MEB.RegionType.HYPER_CUBE, None)
    fwd = mgr._kernel_map(key, okey, *args, False, False)
    tr = mgr._kernel_map(okey, key, *args, True, False)
    assert tr.in_pairs.data_ptr() == fwd.out_pairs.data_ptr() and tr.n_in == fwd.n_out
    # built directly (no cached forward map) it must be the same set
    MEB2, mgr2 = _mgr()
    key2, _ = mgr2.insert_and_map(coords.to(device), [1, 1, 1], "")
    okey2 = mgr2.stride(key2, [2, 2, 2])
    d2 = mgr2.kernel_map(okey2, key2, *args, True, False)
    O.assert_same_kernel_map(d2, tr.to_dict())
    fwd_o = O.kernel_map(coords.numpy(), mgr.get_coordinates(okey).cpu().numpy(), O.make_region(3, 2))[1]
    O.assert_same_kernel_map(d2, {k: np.stack((v[1], v[0])) for k, v in fwd_o.items()})


def test_kernel_map_100k_properties(device):
    """Full config-2 size: size-independent properties + oracle pair sets (the C oracle takes < 1 s)."""
    coords = make_cloud(100000, 70, 3, seed=0)
    MEB, mgr = _mgr()
    key, (um, inv) = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    assert um.numel() == 100000 and torch.equal(inv.cpu(), torch.arange(100000))
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    d = km.to_dict()
    _, km_o = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
    O.assert_same_kernel_map(d, km_o)
    # centre offset is the identity; offset k and 26-k are mirror images (stride-1 symmetry)
    c = d[13].cpu().numpy()
    assert np.array_equal(np.sort(c[0]), np.arange(100000)) and np.array_equal(c[1], c[0])
    for k in range(13):
        a, b = d[k].cpu().numpy(), d[26 - k].cpu().numpy()
        pa = a.T[np.lexsort((a[1], a[0]))]
        pb = b[::-1].T[np.lexsort((b[0], b[1]))]
        assert np.array_equal(pa, pb)
    # determinism: a second build gives identical lists
    MEB2, mgr2 = _mgr()
    key2, _ = mgr2.insert_and_map(coords.to(device), [1, 1, 1], "")
    d2 = mgr2.kernel_map(key2, key2, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    for k in d:
        assert torch.equal(d[k], d2[k])


@pytest.mark.parametrize("n,extent,D,batch,ts", [(6000, 20, 3, 2, 1), (5000, 300, 3, 3, 1), (3000, 9, 4, 2, 1),
                                                 (4000, 50, 2, 1, 1), (2000, 40, 3, 2, 4), (700, 5000, 1, 2, 1),
                                                 (1, 3, 3, 1, 1)])
def test_spatial_index(device, n, extent, D, batch, ts):
    """me_spatial_index_build: `order` is a stable sort of the rows by supercell (batch-major, then the axes), the
    directory holds the first position of every supercell, coords_sorted / pos_of_row are consistent."""
    from minkowskiengine_amd import backend as MEB
    coords = make_cloud(n, extent, D, seed=7 * n + D, batch=batch, negative=True) * torch.tensor([1] + [ts] * D).int()
    cmap, _, _ = MEB._insert(coords.contiguous().to(device), [ts] * D)
    sp = cmap.spatial()
    assert sp is not None
    c = coords.numpy().astype(np.int64)
    order = sp.order.cpu().numpy()
    assert sorted(order.tolist()) == list(range(len(c)))
    assert np.array_equal(sp.pos_of_row.cpu().numpy()[order], np.arange(len(c)))
    assert np.array_equal(sp.coords_sorted.cpu().numpy(), coords.numpy()[order])
    g = sp.grid
    key = c[:, 0] - g.sc_min[0]
    for d in range(D):
        sc = ((c[:, 1 + d] // ts) >> g.shift[d]) - g.sc_min[1 + d]
        assert sc.min() >= 0 and sc.max() < g.sc_dim[1 + d]
        key = key * g.sc_dim[1 + d] + sc
    assert np.array_equal(order, np.argsort(key, kind="stable")), "stable sort by supercell key"
    ds = sp.dir_start.cpu().numpy().astype(np.int64)
    assert ds[0] == 0 and ds[-1] == len(c) and len(ds) == sp.m + 1
    assert np.array_equal(np.diff(ds), np.bincount(key, minlength=sp.m))


LDS_CASES = [
    # n, extent, D, kernel_size, dilation, region, batch, tensor stride
    (6000, 20, 3, 3, 1, 0, 2, 1),
    (6000, 90, 3, 3, 1, 0, 3, 1),           # sparse, several supercells per axis
    (4000, 14, 3, 5, 1, 0, 2, 1),           # K = 125, halo 2
    (4000, 30, 3, 3, 3, 0, 1, 1),           # dilation 3: halo 3
    (3000, 9, 4, 3, 1, 0, 2, 1),            # 4-D, K = 81
    (3000, 40, 2, [3, 2], 1, 0, 2, 1),      # mixed odd / even kernel
    (3000, 14, 3, 3, 1, 1, 2, 1),           # HYPER_CROSS
    (3000, 40, 3, 3, 1, 0, 2, 2),           # maps of tensor stride 2 (a layer below a strided conv)
    (2000, 3000, 1, 3, 1, 0, 2, 1),         # 1-D
    (1, 2, 3, 3, 1, 0, 1, 1),
]


@pytest.mark.parametrize("n,extent,D,ks,dil,region,batch,ts", LDS_CASES)
def test_lds_bucketed_kernel_map_equals_the_flat_table_build(device, n, extent, D, ks, dil, region, batch, ts):
    """The LDS-bucketed build (spatial index + k_kmap_probe_lds, position-space tables, no host sync) and the
    flat-table build (k_kmap_probe) produce the same row-space neighbour tables and the same pair sets, and both
    match the oracle."""
    from minkowskiengine_amd import backend as MEB
    coords = make_cloud(n, extent, D, seed=3 * n + D, batch=batch, negative=True) * torch.tensor([1] + [ts] * D).int()
    coords = coords.contiguous()
    ksl = [ks] * D if isinstance(ks, int) else ks
    kms = []
    for spatial in (True, False):
        old = MEB._SPATIAL_MAPS
        MEB._SPATIAL_MAPS = spatial
        try:
            mgr = MEB.CoordinateMapManagerGPU_c10()
            key, _ = mgr.insert_and_map(coords.to(device), [ts] * D, "")
            km = mgr._kernel_map(key, key, ksl, [1] * D, [dil] * D, MEB.RegionType(region), None, False, False)
            kms.append((km, km.table("out").cpu().numpy(), km.table("in").cpu().numpy(), km.to_dict(), km.n_pairs))
        finally:
            MEB._SPATIAL_MAPS = old
    (a, a_out, a_in, a_d, a_n), (b, b_out, b_in, b_d, b_n) = kms
    assert a._store.get("order_out") is not None, "the LDS-bucketed path must have been taken"
    assert b._store.get("order_out") is None
    assert a_n == b_n and np.array_equal(a_out, b_out) and np.array_equal(a_in, b_in)
    O.assert_same_kernel_map(a_d, b_d)
    nbr_o, km_o = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(D, ksl, dil, ts, region))
    assert np.array_equal(a_out, nbr_o)
    O.assert_same_kernel_map(a_d, km_o)


def test_lds_bucketed_build_falls_back(device):
    """A kernel whose reach exceeds one supercell (dilation 20 > 16 cells) and maps of different tensor stride take
    the flat-table build."""
    from minkowskiengine_amd import backend as MEB
    coords = make_cloud(3000, 60, 3, seed=5, batch=2)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [20] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    assert km._store.get("order_out") is None
    _, km_o = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3, 20, 1))
    O.assert_same_kernel_map(km.to_dict(), km_o)
    okey = mgr.stride(key, [2, 2, 2])
    km2 = mgr._kernel_map(key, okey, [2] * 3, [2] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    assert km2._store.get("order_out") is None


@pytest.mark.parametrize("host_readback", [True, False])
def test_kernel_map_count_recounts_an_existing_table(device, host_readback):
    """me_kernel_map_count: the per-offset pair prefix of a neighbour table that already exists (e.g. one produced by
    the LDS-bucketed probe of another library instance, or edited by pruning) — with the host read-back and without it
    (NULL host pointer: no synchronisation, the prefix stays on the device)."""
    import ctypes
    from minkowskiengine_amd import backend as MEB, _lib
    lib = _lib.load()
    coords = make_cloud(5000, 16, 3, seed=11, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    nbr = km.table("out").contiguous()
    volume, n = nbr.shape
    ws = torch.empty(int(lib.me_kernel_map_workspace_bytes(n, volume)), dtype=torch.uint8, device=device)
    koffs_dev = torch.full((volume + 1,), -1, dtype=torch.int64, device=device)
    host = (ctypes.c_int64 * (volume + 1))() if host_readback else None
    _lib.check(lib.me_kernel_map_count(nbr.data_ptr(), n, volume, host, koffs_dev.data_ptr(), ws.data_ptr(), ws.numel(),
                                       torch.cuda.current_stream().cuda_stream))
    want = [0]
    for k in range(volume):
        want.append(want[-1] + int((nbr[k] >= 0).sum()))
    assert koffs_dev.cpu().tolist() == want == list(km.k_offsets)
    if host_readback:
        assert [int(v) for v in host] == want
