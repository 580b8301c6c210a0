"""GPU parity of the convolution feature kernels (forward, dgrad, wgrad; strided and transposed maps)
against the oracle and against the fixtures produced by the reference itself.  Tolerance: PER ELEMENT
|got - want| <= 1e-4 + 1e-4 |want| (BASELINE.json's north_star bar for fp32; helpers.assert_close); typical
error is ~3e-7."""
import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from helpers import assert_close, golden_cases, golden_kmap, make_cloud, row_mapping

pytestmark = pytest.mark.gpu


def _run_layer(device, coords, cin, cout, ks, stride=1, dil=1, seed=0, bias=False):
    import minkowskiengine_amd as ME
    D = coords.shape[1] - 1
    g = torch.Generator().manual_seed(seed)
    feats = torch.rand(coords.shape[0], cin, generator=g)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, dilation=dil, bias=bias, dimension=D)
    with torch.no_grad():
        conv.kernel.copy_(torch.rand(conv.kernel.shape, generator=g) - 0.5)
    conv = conv.to(device)
    x = ME.SparseTensor(feats.to(device), coords.to(device), requires_grad=True)
    y = conv(x)
    gy = torch.rand(y.F.shape, generator=g)
    y.F.backward(gy.to(device))
    return conv, x, y, feats, gy


CONV_CASES = [
    # n, extent, D, cin, cout, ks, stride, dil
    (3000, 14, 3, 16, 32, 3, 1, 1),
    (3000, 14, 3, 64, 128, 3, 1, 1),      # config-2 channel shape
    (3000, 40, 3, 64, 128, 3, 1, 1),      # sparse: most (tile, offset) groups nearly empty
    (2500, 14, 3, 3, 32, 5, 1, 1),        # MinkUNet stem: cin = 3, K = 125
    (2500, 14, 3, 32, 96, 3, 1, 1),       # cout not a multiple of 64
    (2500, 14, 3, 96, 32, 3, 1, 1),       # cin = 96 (chunks of 32)
    (2500, 14, 3, 20, 24, 3, 1, 1),       # odd channel counts (scalar gather path off: 20 % 4 == 0)
    (2500, 14, 3, 5, 7, 3, 1, 1),         # channels not multiples of 4 (scalar paths)
    (2500, 14, 3, 32, 32, 2, 2, 1),       # down conv k=2 s=2
    (2500, 14, 3, 16, 16, 3, 2, 1),
    (2500, 14, 3, 8, 8, 3, 1, 2),
    (2000, 8, 4, 32, 64, 3, 1, 1),        # 4-D, K = 81 (config-5 channel shape)
    (300, 6, 3, 128, 256, 3, 1, 1),       # few rows, many channels
    (1, 2, 3, 4, 4, 3, 1, 1),
]


@pytest.mark.parametrize("pipe", ["bf16x6", "fp32_mfma"])
@pytest.mark.parametrize("n,extent,D,cin,cout,ks,stride,dil", CONV_CASES)
def test_conv_forward_backward_vs_oracle(device, monkeypatch, pipe, n, extent, D, cin, cout, ks, stride, dil):
    """Both fp32 forward / dgrad kernels: the exactly-split operands on the bf16 matrix pipe (default where
    c_src % 8 == 0) and the fp32-MFMA kernel (ME_AMD_F32_SPLIT=0, and the fall-back for other channel counts)."""
    from minkowskiengine_amd import backend as MEB, _lib
    monkeypatch.setattr(MEB, "_F32_SPLIT", pipe == "bf16x6")
    # the weight gradient follows: k_wgrad_f32x3 wherever the rows allow (-4) / never (-3)
    lib = _lib.load()
    lib.me_debug_set_wgrad_config(-4 if pipe == "bf16x6" else -3, 0)
    try:
        _conv_case(device, n, extent, D, cin, cout, ks, stride, dil)
    finally:
        lib.me_debug_set_wgrad_config(0, 0)


@pytest.mark.parametrize("n,extent,D,cin,cout,ks,stride,dil", CONV_CASES)
def test_conv_vs_oracle_on_the_native_host(device, n, extent, D, cin, cout, ks, stride, dil):
    """The same oracle comparison on the shipped default: the native C++ host layer (csrc_host/) with its own
    kernel-selection policy (split pipe where c_src * c_dst >= 8192, fp32 MFMA below, multi-offset batches by density)."""
    import minkowskiengine_amd as ME
    prev = ME.get_host()
    ME.set_host("native")
    try:
        _conv_case(device, n, extent, D, cin, cout, ks, stride, dil, expect_native=True)
    finally:
        ME.set_host(prev)


def _conv_case(device, n, extent, D, cin, cout, ks, stride, dil, expect_native=False):
    coords = make_cloud(n, extent, D, seed=n + cin, batch=2 if n > 100 else 1, negative=True)
    conv, x, y, feats, gy = _run_layer(device, coords, cin, cout, ks, stride, dil)
    assert bool(x.coordinate_manager._native) == expect_native
    in_c = coords.numpy()
    out_c = y.C.cpu().numpy()
    if stride == 1:
        assert np.array_equal(out_c, in_c)
    else:
        assert np.array_equal(out_c, O.stride_map(in_c, [stride] * D)[0])
    _, km = O.kernel_map(in_c, out_c, O.make_region(D, ks, dil, 1))
    w = conv.kernel.detach().cpu().numpy()
    ref = O.conv_forward(feats.numpy(), w, km, len(out_c))
    assert_close(y.F.detach().cpu().numpy(), ref)
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
    assert_close(x.F.grad.cpu().numpy(), gi)
    assert_close(conv.kernel.grad.cpu().numpy(), gw)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("n,extent,D,cin,cout,ks,stride,dil", [CONV_CASES[1], CONV_CASES[5], CONV_CASES[7]])
def test_wide_address_kernels_match_the_default(device, monkeypatch, dtype, n, extent, D, cin, cout, ks, stride, dil):
    """Feature matrices of 4 GiB and more (or 2^24 rows) take kernels with 64-bit gather addresses instead of the
    32-bit offsets every other test exercises; debug variant 6 forces them.  Same plan, same summation order:
    forward, grad_in and grad_kernel must be bit-identical to the default kernels.  (The multi-offset instantiation of
    the fp32-MFMA kernel exists for 32-bit offsets only — a >= 4 GiB matrix runs the plain batches, whose partial sums
    are grouped differently — so it is switched off for this comparison.)"""
    from minkowskiengine_amd import _lib, backend as MEB
    import minkowskiengine_amd as ME
    lib = _lib.load()
    monkeypatch.setattr(MEB, "_F32_FUSE", False)
    coords = make_cloud(n, extent, D, seed=n + cin, batch=2, negative=True)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    g = torch.Generator().manual_seed(3)
    feats = torch.rand(coords.shape[0], cin, generator=g)
    kernel = torch.rand(ks ** D, cin, cout, generator=g) - 0.5
    results = []
    for variant in (0, 6):
        _lib.check(lib.me_debug_set_conv_variant(variant))
        try:
            conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, dilation=dil, dimension=D)
            with torch.no_grad():
                conv.kernel.copy_(kernel)
            conv = conv.to(device)
            x = ME.SparseTensor(feats.to(device).to(tdt), coords.to(device), requires_grad=True)
            y = conv(x)
            y.F.backward(torch.ones_like(y.F))
            torch.cuda.synchronize()
            results.append((y.F.detach().clone(), x.F.grad.clone(), conv.kernel.grad.clone()))
        finally:
            lib.me_debug_set_conv_variant(0)
    for a, b in zip(*results):
        assert torch.equal(a, b)


def test_mfma_and_naive_kernels_agree(device):
    from minkowskiengine_amd import backend as MEB
    coords = make_cloud(3000, 14, 3, seed=4, negative=True).to(device)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords, [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(3000, 24, generator=g).to(device)
    w = (torch.rand(27, 24, 40, generator=g) - 0.5).to(device)
    gy = torch.rand(3000, 40, generator=g).to(device)
    y1, y0 = MEB._conv_forward(x, w, km, "mfma"), MEB._conv_forward(x, w, km, "naive")
    assert_close(y1, y0, 1e-5, 1e-5)
    a1, b1 = MEB._conv_backward(x, gy, w, km, "mfma")
    a0, b0 = MEB._conv_backward(x, gy, w, km, "naive")
    assert_close(a1, a0, 1e-5, 1e-5)
    assert_close(b1, b0, 2e-5, 2e-5)


def test_bitwise_reproducible(device, host_layer):
    coords = make_cloud(4000, 14, 3, seed=9)
    r1 = _run_layer(device, coords, 32, 64, 3)
    r2 = _run_layer(device, coords, 32, 64, 3)
    assert torch.equal(r1[2].F, r2[2].F) and torch.equal(r1[1].F.grad, r2[1].F.grad)
    assert torch.equal(r1[0].kernel.grad, r2[0].kernel.grad)


def test_non_default_stream(device, host_layer):
    """Every launch goes to torch's CURRENT stream (DistributedDataParallel overlaps its collectives with the
    backward pass on side streams; users wrap steps in torch.cuda.stream): coordinate insertion, kernel map, tile
    plans, forward, dgrad, wgrad and pooling issued on a side stream — with the default stream kept busy by an
    unrelated long kernel queue — give bit-identical results to the default-stream run."""
    import minkowskiengine_amd as ME
    coords = make_cloud(20000, 30, 3, seed=21)
    ref = _run_layer(device, coords, 32, 64, 3)
    side = torch.cuda.Stream(device)
    busy = torch.rand(4096, 4096, device=device)
    torch.cuda.synchronize()
    for _ in range(20):                       # ~tens of ms of work queued on the DEFAULT stream
        busy = (busy @ busy) * 2.4e-4      # stays O(1)
    with torch.cuda.stream(side):
        got = _run_layer(device, coords, 32, 64, 3)
        pool = ME.MinkowskiMaxPooling(kernel_size=2, stride=2, dimension=3)(got[1])
        pooled = pool.F.clone()
    side.synchronize()                        # NOT a device-wide sync: only the side stream's work is awaited
    assert torch.equal(got[2].F, ref[2].F) and torch.equal(got[1].F.grad, ref[1].F.grad)
    assert torch.equal(got[0].kernel.grad, ref[0].kernel.grad)
    torch.cuda.synchronize()
    pool_ref = ME.MinkowskiMaxPooling(kernel_size=2, stride=2, dimension=3)(ref[1])
    assert torch.equal(pooled, pool_ref.F)


@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 64), (64, 64), (128, 128)])
@pytest.mark.parametrize("variant,T,cap", [(3000, 96, 4), (3064, 80, 3), (3100, 96, 4), (3100, 64, 2)])
@pytest.mark.parametrize("spatial,tile_order", [(False, "rows"), (True, "rows"), (True, "spatial")])
def test_lds_dma_tile_kernel_matches_the_oracle(device, monkeypatch, cin, cout, variant, T, cap, spatial, tile_order):
    """k_conv_tile_dma_f32 (LDS-DMA gather, accumulators initialised from the LDS tile; debug variants 3000 / 3064 /
    3100 select its instantiations: eight waves x 16 columns, four waves x 16 columns, four waves with 32 columns or
    two source chunks per batch) on flat-table maps, LDS-bucketed maps with row tiles and with spatial tiles:
    forward and dgrad against the oracle, per element."""
    from minkowskiengine_amd import _lib
    from minkowskiengine_amd import backend as MEB
    lib = _lib.load()
    monkeypatch.setattr(MEB, "_F32_SPLIT", False)
    monkeypatch.setattr(MEB, "_SPATIAL_MAPS", spatial)
    monkeypatch.setattr(MEB, "_TILE_ORDER", tile_order)
    monkeypatch.setattr(MEB, "_TILE_ROWS", T)
    monkeypatch.setattr(MEB, "_BATCH_GROUPS", cap)
    if not lib.me_debug_variants_compiled():
        pytest.skip("the LDS-DMA experiment family is only in a -DME_DEBUG_VARIANTS build (scripts/build_debug.sh)")
    coords = make_cloud(5000, 18, 3, seed=cin + cout, batch=2, negative=True)
    _lib.check(lib.me_debug_set_conv_variant(variant))
    try:
        conv, x, y, feats, gy = _run_layer(device, coords, cin, cout, 3)
    finally:
        lib.me_debug_set_conv_variant(0)
    km_gpu = x.coordinate_manager._manager._kernel_map(x.coordinate_map_key, y.coordinate_map_key, [3] * 3, [1] * 3,
                                                       [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    assert (km_gpu._store.get("order_out") is not None) == spatial
    _, km = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
    w = conv.kernel.detach().cpu().numpy()
    assert_close(y.F, O.conv_forward(feats.numpy(), w, km, len(coords)))
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
    assert_close(x.F.grad, gi)
    assert_close(conv.kernel.grad, gw)


@pytest.mark.parametrize("cin,cout,scale", [(64, 128, 1.0), (32, 32, 1e-3), (96, 96, 300.0), (256, 64, 1.0)])
def test_split_bf16_pipe_is_fp32_grade(device, monkeypatch, cin, cout, scale):
    """k_conv_tile_f32x3 rebuilds every fp32 product from six bf16 MFMAs (a = a1 + a2 + a3 exactly; the dropped terms
    are < 2^-23 |a b|).  Against float64 ground truth its error must be of fp32 order: per element
    <= 2e-6 * sum_k |x| |w| (the fp32-MFMA kernel measures ~2e-7 of that sum; bf16 arithmetic would be ~4e-3), for
    operands across magnitudes, signs and with values whose low mantissa bits are all set; and it must not be worse
    than 4x the fp32-MFMA kernel's own error."""
    from minkowskiengine_amd import backend as MEB
    coords = make_cloud(4000, 12, 3, seed=cin, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(cin + cout)
    x = ((torch.rand(4000, cin, generator=g) - 0.5) * scale)
    x[::7] = torch.nextafter(x[::7], torch.full_like(x[::7], 1e30))        # odd low bits
    x[5] = 0.0
    x[6, :] = torch.tensor(1.0 + 2.0 ** -23) * scale                           # 1 + ulp: a2 = 0, a3 carries the bit
    w = (torch.rand(27, cin, cout, generator=g) - 0.5) / scale
    _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
    x64, w64 = x.double().numpy(), w.double().numpy()
    truth = O.conv_forward(x64, w64, okm, len(coords))
    bound = O.conv_forward(np.abs(x64), np.abs(w64), okm, len(coords))        # sum |x| |w| per output element
    errs = {}
    for split in (True, False):
        monkeypatch.setattr(MEB, "_F32_SPLIT", split)
        y = MEB._conv_forward(x.to(device), w.to(device), km, "mfma").double().cpu().numpy()
        errs[split] = float(np.max(np.abs(y - truth) / np.maximum(bound, 1e-300)))
    print(f"relative to sum|x||w|: bf16x6 {errs[True]:.2e}, fp32 MFMA {errs[False]:.2e}")
    assert errs[True] <= 2e-6, errs
    assert errs[True] <= 4 * errs[False] + 1e-7, errs


@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 128), (32, 64)])
def test_wgrad_range_order_and_split_kernel(device, cin, cout):
    """(a) The XCD-aware range order of the weight-gradient kernels only changes which workgroup takes which range:
    results are bit-identical to launch order.  (b) k_wgrad_f32x3 (fp32 rows split into three bf16 terms, six bf16
    MFMAs) against float64 ground truth: error of fp32 order, not worse than 4x the fp32-MFMA kernel's."""
    from minkowskiengine_amd import backend as MEB, _lib
    lib = _lib.load()
    coords = make_cloud(30000, 30, 3, seed=cin, negative=True)
    n = coords.shape[0]
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(n, cin, generator=g) - 0.5
    gy = torch.rand(n, cout, generator=g) - 0.5
    w = torch.rand(27, cin, cout, generator=g).to(device)
    _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
    _, truth = O.conv_backward(x.double().numpy(), gy.double().numpy(), w.double().cpu().numpy(), okm)
    _, bound = O.conv_backward(x.abs().double().numpy(), gy.abs().double().numpy(), w.double().cpu().numpy(), okm)
    got = {}
    try:
        for name, depth, order in (("split", -4, 0), ("split_launch_order", -4, -1), ("mfma", -3, 0),
                                   ("mfma_launch_order", -3, -1)):
            lib.me_debug_set_wgrad_config(depth, 0)
            lib.me_debug_set_wgrad_order(order)
            got[name] = MEB._conv_backward(x.to(device), gy.to(device), w, km, "mfma")[1].clone()
    finally:
        lib.me_debug_set_wgrad_config(0, 0)
        lib.me_debug_set_wgrad_order(0)
    assert torch.equal(got["split"], got["split_launch_order"])
    assert torch.equal(got["mfma"], got["mfma_launch_order"])
    err = {k: float(np.max(np.abs(v.double().cpu().numpy() - truth) / np.maximum(bound, 1e-300)))
           for k, v in got.items()}
    print(f"wgrad {cin}->{cout}: error relative to sum |x||dy|: bf16x6 {err['split']:.2e}, fp32 MFMA {err['mfma']:.2e}")
    assert err["split"] <= 2e-6 and err["split"] <= 4 * err["mfma"] + 1e-7, err


@pytest.mark.parametrize("n,extent,cin,cout", [(6000, 18, 64, 128), (6000, 18, 128, 64), (5000, 40, 64, 64),
                                               (4000, 16, 96, 128), (4000, 16, 128, 128), (3000, 14, 72, 120)])
def test_wave_specialised_split_kernel_is_bit_identical(device, monkeypatch, n, extent, cin, cout):
    """k_conv_tile_f32x3_ws (four multiplier waves + four producer waves; the default for 64- and 128-column slabs)
    runs the same plan and the same MFMA sequence per output element as the ping-pong kernel k_conv_tile_f32x3
    (debug variant 30): forward and input gradient must be bit-identical, and both match the oracle."""
    from minkowskiengine_amd import backend as MEB, _lib
    lib = _lib.load()
    monkeypatch.setattr(MEB, "_F32_SPLIT", True)
    coords = make_cloud(n, extent, 3, seed=cin + cout, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(coords.shape[0], cin, generator=g) - 0.5)
    gy = (torch.rand(coords.shape[0], cout, generator=g) - 0.5)
    w = (torch.rand(27, cin, cout, generator=g) - 0.5)
    res = {}
    try:
        # shipped, four multipliers, ping-pong; in a tuning build also the round-6 experiments that were measured and not
        # adopted (profiles/r06_ring_persistent_ab.md): ring kernel with three / two stage slots, persistent kernel
        extra = (40, 41, 43) if lib.me_debug_variants_compiled() else ()
        for variant in (0, 31, 30) + extra:
            _lib.check(lib.me_debug_set_conv_variant(variant))
            y = MEB._conv_forward(x.to(device), w.to(device), km, "mfma")
            gi = MEB._conv_target(gy.to(device), w.to(device), km, "in", km.n_in, name="d", transposed=True)
            res[variant] = (y.clone(), gi.clone())
    finally:
        lib.me_debug_set_conv_variant(0)
    for v in (31, 30) + extra:
        assert torch.equal(res[0][0], res[v][0]) and torch.equal(res[0][1], res[v][1]), v
    _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
    assert_close(res[0][0], O.conv_forward(x.numpy(), w.numpy(), okm, coords.shape[0]))
    assert_close(res[0][1], O.conv_backward(x.numpy(), gy.numpy(), w.numpy(), okm)[0])


@pytest.mark.parametrize("spatial_maps", [False, True])
@pytest.mark.parametrize("n,extent,cin,cout", [(6000, 18, 64, 128), (5000, 40, 128, 64), (3000, 14, 72, 120)])
def test_matrix_bound_kernels_take_spatial_tiles(device, monkeypatch, n, extent, cin, cout, spatial_maps):
    """The fp32 kernels on the bf16 matrix pipe take spatially compact tiles by default (supercell order of the
    target map: the map's own position space, or — flat-table maps — the spatial index of the coordinate map,
    KernelMapGPU._flat_order).  The tile order only regroups target rows: the sum of a row runs over the offsets in
    the same order, so forward and input gradient are bit-identical to row tiles, and both match the oracle."""
    from minkowskiengine_amd import backend as MEB
    monkeypatch.setattr(MEB, "_F32_SPLIT", True)
    monkeypatch.setattr(MEB, "_SPATIAL_MAPS", spatial_maps)
    coords = make_cloud(n, extent, 3, seed=cin + cout, batch=2, negative=True)
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(coords.shape[0], cin, generator=g) - 0.5)
    gy = (torch.rand(coords.shape[0], cout, generator=g) - 0.5)
    w = (torch.rand(27, cin, cout, generator=g) - 0.5)
    res = {}
    for order in ("auto", "rows"):
        monkeypatch.setattr(MEB, "_TILE_ORDER", order)
        mgr = MEB.CoordinateMapManagerGPU_c10()
        key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
        km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
        y = MEB._conv_forward(x.to(device), w.to(device), km, "mfma")
        gi = MEB._conv_target(gy.to(device), w.to(device), km, "in", km.n_in, name="d", transposed=True)
        _, cfg = MEB._conv_launch_cfg(km, "out", km.n_out, cin, cout, False)
        res[order] = (y.clone(), gi.clone(), cfg[6])
    assert res["auto"][2] is not None and res["rows"][2] is None      # a tile permutation / none
    perm = res["auto"][2].long().cpu()
    assert torch.equal(torch.sort(perm).values, torch.arange(coords.shape[0]))
    assert torch.equal(res["auto"][0], res["rows"][0]) and torch.equal(res["auto"][1], res["rows"][1])
    _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
    assert_close(res["auto"][0], O.conv_forward(x.numpy(), w.numpy(), okm, coords.shape[0]))
    assert_close(res["auto"][1], O.conv_backward(x.numpy(), gy.numpy(), w.numpy(), okm)[0])


def test_bias_and_use_mm(device, host_layer):
    import minkowskiengine_amd as ME
    coords = make_cloud(500, 10, 3, seed=2).to(device)
    x = ME.SparseTensor(torch.rand(500, 8).to(device), coords)
    conv1 = ME.MinkowskiConvolution(8, 6, kernel_size=1, bias=True, dimension=3).to(device)
    y = conv1(x)
    assert torch.allclose(y.F, x.F @ conv1.kernel + conv1.bias, atol=1e-6)
    assert y.coordinate_map_key == x.coordinate_map_key


@pytest.mark.parametrize("path", golden_cases())
def test_against_reference_fixtures(device, host_layer, path):
    """The fixtures hold the REFERENCE's outputs (tests/golden/make_golden.py).  Row order of strided
    maps is implementation-defined in the reference, so output rows are relabelled by coordinate."""
    import minkowskiengine_amd as ME
    z = np.load(path)
    D = z["coords"].shape[1] - 1
    ks, st, dl = z["kernel_size"].tolist(), z["stride"].tolist(), z["dilation"].tolist()
    cin, cout = z["kernel"].shape[1:]
    coords = torch.from_numpy(z["coords"]).to(device)
    mgr = ME.CoordinateManager(D=D)
    key, (um, inv) = mgr.insert_and_map(coords, [1] * D, "")
    assert np.array_equal(um.cpu().numpy(), z["unique_map"]) and np.array_equal(inv.cpu().numpy(), z["inverse_map"])
    x = ME.SparseTensor(torch.from_numpy(z["feats"]).to(device), coordinate_map_key=key, coordinate_manager=mgr,
                        requires_grad=True)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=st, dilation=dl, dimension=D)
    with torch.no_grad():
        conv.kernel.copy_(torch.from_numpy(z["kernel"]))
    conv = conv.to(device)
    y = conv(x)
    assert y.tensor_stride == z["out_tensor_stride"].tolist()
    m = row_mapping(y.C.cpu().numpy(), z["out_coords"])      # our out row -> reference out row
    d = mgr.kernel_map(key, y.coordinate_map_key, stride=st, kernel_size=ks, dilation=dl)
    O.assert_same_kernel_map({k: np.stack((v[0].cpu().numpy().astype(np.int64), m[v[1].cpu().numpy()]))
                              for k, v in d.items()}, golden_kmap(z))
    assert_close(y.F.detach().cpu().numpy(), z["out"][m])
    y.F.backward(torch.from_numpy(z["grad_out"][m]).to(device))
    assert_close(x.F.grad.cpu().numpy(), z["grad_in"])
    assert_close(conv.kernel.grad.cpu().numpy(), z["grad_kernel"])
    if "up" in z.files:
        convt = ME.MinkowskiConvolutionTranspose(cout, cin, kernel_size=ks, stride=st, dilation=dl, dimension=D)
        with torch.no_grad():
            convt.kernel.copy_(torch.from_numpy(z["kernel_t"]))
        convt = convt.to(device)
        xin = ME.SparseTensor(torch.from_numpy(z["out"][m]).to(device), coordinate_map_key=y.coordinate_map_key,
                              coordinate_manager=mgr, requires_grad=True)
        up = convt(xin)
        assert up.coordinate_map_key == key
        assert_close(up.F.detach().cpu().numpy(), z["up"])
        up.F.backward(torch.from_numpy(z["up_grad_out"]).to(device))
        assert_close(xin.F.grad.cpu().numpy(), z["up_grad_in"][m])
        assert_close(convt.kernel.grad.cpu().numpy(), z["up_grad_kernel"])


def test_config2_full_size(device, host_layer):
    """BASELINE config 2 at full size (100k voxels, 64 -> 128, k = 3): oracle values (numpy, a few
    seconds) + linearity as a size-independent property."""
    coords = make_cloud(100000, 70, 3, seed=0)
    conv, x, y, feats, gy = _run_layer(device, coords, 64, 128, 3)
    _, km = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
    w = conv.kernel.detach().cpu().numpy()
    ref = O.conv_forward(feats.numpy(), w, km, 100000, dtype=np.float32)
    assert_close(y.F.detach().cpu().numpy(), ref)
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km, dtype=np.float32)
    assert_close(x.F.grad.cpu().numpy(), gi)
    assert_close(conv.kernel.grad.cpu().numpy(), gw)
    # linearity: conv(2a - 3b) == 2 conv(a) - 3 conv(b)
    import minkowskiengine_amd as ME
    a, b = torch.rand_like(x.F.detach()), torch.rand_like(x.F.detach())
    mk = lambda f: ME.SparseTensor(f, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
    with torch.no_grad():
        lhs = conv(mk(2 * a - 3 * b)).F
        rhs = 2 * conv(mk(a)).F - 3 * conv(mk(b)).F
    assert_close(lhs, rhs, 1e-4, 1e-4, "linearity")


def test_config5_full_size(device, host_layer):
    """BASELINE config 5 at full size: 4-D (3-D + time) k = 3 (K = 81), 8 frames x 50k voxels in
    [0,100)^3 x [0,8), 32 -> 64 — the high-D coordinate-hash stress.  Kernel map pair sets identical to the
    oracle's per offset, features within 1e-4, plus mirror symmetry of the pair counts
    (|map[k]| == |map[K-1-k]| for a stride-1 map onto itself)."""
    import minkowskiengine_amd as ME
    g = torch.Generator().manual_seed(5)
    n = 400000
    pts = torch.stack([torch.randint(0, e, (int(1.6 * n),), generator=g) for e in (100, 100, 100, 8)], 1)
    pts = torch.unique(pts, dim=0)
    pts = pts[torch.randperm(pts.shape[0], generator=g)][:n]
    coords = torch.cat([torch.zeros(n, 1, dtype=torch.long), pts], 1).int().contiguous()
    conv, x, y, feats, gy = _run_layer(device, coords, 32, 64, 3)
    assert np.array_equal(y.C.cpu().numpy(), coords.numpy())
    _, km = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(4, 3))
    d = x.coordinate_manager.kernel_map(x.coordinate_map_key, y.coordinate_map_key, kernel_size=3)
    O.assert_same_kernel_map({k: v.cpu().numpy().astype(np.int64) for k, v in d.items()}, km)
    sizes = {k: v.shape[1] for k, v in d.items()}
    assert all(sizes.get(k, 0) == sizes.get(80 - k, 0) for k in range(81)) and sizes[40] == n
    w = conv.kernel.detach().cpu().numpy()
    ref = O.conv_forward(feats.numpy(), w, km, n, dtype=np.float32)
    assert_close(y.F.detach().cpu().numpy(), ref)
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km, dtype=np.float32)
    assert_close(x.F.grad.cpu().numpy(), gi)
    assert_close(conv.kernel.grad.cpu().numpy(), gw)


def _same_nonfinite(got, want, what):
    """Element-wise class agreement for non-finite results: NaN where the reference is NaN, the same infinity where
    it is infinite; finite elements within the 1e-4 bar."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN positions differ"
    inf = np.isinf(want)
    assert np.array_equal(np.isinf(got), inf), f"{what}: infinity positions differ"
    assert np.array_equal(got[inf], want[inf]), f"{what}: signs of infinities differ"
    fin = np.isfinite(want)
    assert_close(np.where(fin, got, 0.0), np.where(fin, want, 0.0), what=what)


@pytest.mark.parametrize("pipe", ["bf16x6", "fp32_mfma"])
@pytest.mark.parametrize("cin,cout", [(64, 128), (32, 64)])
def test_nonfinite_and_tiny_inputs_follow_fp32_semantics(device, monkeypatch, pipe, cin, cout):
    """+-inf / NaN / 1e-38 in the features and in the upstream gradient: the reference multiplies in fp32 (sgemm,
    src/math_functions_cpu.cpp:29-46), where inf * w is +-inf (NaN for w = 0) and sums of opposite infinities are NaN.
    The split kernels keep that (conv_common.hpp split3: a non-finite row value travels in the third bf16 plane alone),
    including weights that are exact in bf16 (second / third plane zero) and exact zeros; the one documented divergence
    — a non-finite WEIGHT-side value (weights; dy in the weight gradient) gives NaN where fp32 keeps an infinity's
    sign — is pinned as 'non-finite wherever the reference is non-finite'."""
    from minkowskiengine_amd import backend as MEB, _lib
    lib = _lib.load()
    monkeypatch.setattr(MEB, "_F32_SPLIT", pipe == "bf16x6")
    lib.me_debug_set_wgrad_config(-4 if pipe == "bf16x6" else -3, 0)
    try:
        coords = make_cloud(3000, 14, 3, seed=cin, negative=True)
        n = coords.shape[0]
        mgr = MEB.CoordinateMapManagerGPU_c10()
        key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
        km = mgr._kernel_map(key, key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
        g = torch.Generator().manual_seed(7)
        x = torch.rand(n, cin, generator=g) - 0.5
        w = torch.rand(27, cin, cout, generator=g) - 0.5
        w[:, 3, :] = 1.0                 # exact in bf16: planes 2 and 3 of the split are zero
        w[:, 4, ::2] = 0.0               # inf * 0 = NaN
        w[:, 5, :] = -0.5
        gy = torch.rand(n, cout, generator=g) - 0.5
        inf = float("inf")
        x[10, 3], x[11, 3], x[12, 4], x[13, 5], x[14, 7] = inf, -inf, inf, inf, float("nan")
        x[15, 3], x[15, 5] = inf, inf    # +inf * 1 and +inf * -0.5 in one sum: NaN
        x[20, :] = 1e-38                 # below 2^-100: the third plane may lose bits (absolute error < 2^-133)
        x[21, 0] = 1e-38
        gy[30, 2], gy[31, 9], gy[32, 1] = inf, -inf, float("nan")
        _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
        with np.errstate(invalid="ignore", over="ignore"):
            want_y = O.conv_forward(x.numpy(), w.numpy(), okm, n)
            want_gi, want_gw = O.conv_backward(x.numpy(), gy.numpy(), w.numpy(), okm)
        y = MEB._conv_forward(x.to(device), w.to(device), km, "mfma")
        gi, gw = MEB._conv_backward(x.to(device), gy.to(device), w.to(device), km, "mfma")
        assert np.isnan(want_y).any() and np.isinf(want_y).any() and np.isfinite(want_y).mean() > 0.9
        _same_nonfinite(y.cpu().numpy(), want_y, "forward")
        _same_nonfinite(gi.cpu().numpy(), want_gi, "grad_in")
        # weight gradient: x (row side) and dy (weight side) both carry non-finite values here
        gw, want_gw = gw.cpu().numpy().astype(np.float64), np.asarray(want_gw, np.float64)
        bad = ~np.isfinite(want_gw)
        assert bad.any() and np.array_equal(~np.isfinite(gw), bad), "grad_kernel: non-finite positions differ"
        assert_close(np.where(bad, 0.0, gw), np.where(bad, 0.0, want_gw), what="grad_kernel")
        # a non-finite weight: every output it reaches is non-finite (NaN in the split kernels, +-inf / NaN in fp32)
        w2 = w.clone()
        w2[13, 6, 5] = inf
        with np.errstate(invalid="ignore", over="ignore"):
            want2 = O.conv_forward(x.numpy(), w2.numpy(), okm, n)
        y2 = MEB._conv_forward(x.to(device), w2.to(device), km, "mfma").cpu().numpy()
        assert np.array_equal(~np.isfinite(y2), ~np.isfinite(want2))
    finally:
        lib.me_debug_set_wgrad_config(0, 0)


@pytest.mark.parametrize("n,extent,D,cin,cout", [(3000, 9, 4, 32, 64), (3000, 9, 4, 64, 32), (6000, 40, 3, 32, 32),
                                                  (5000, 40, 3, 16, 48), (5000, 40, 3, 20, 24), (4000, 30, 3, 64, 64),
                                                  (3000, 14, 3, 32, 64)])
def test_multi_offset_batches_of_the_fp32_kernel_match_the_oracle(device, monkeypatch, n, extent, D, cin, cout):
    """me_conv_target_f32_fused: runs of single-group batches of consecutive offsets (a sparse map's plan) staged and
    multiplied together, each group with its own offset's weights.  Forward and input gradient against the oracle
    per element (1e-4 + 1e-4 |ref|), bitwise reproducible, and the unfused kernel agrees to fp32 rounding."""
    from minkowskiengine_amd import backend as MEB
    monkeypatch.setattr(MEB, "_F32_SPLIT", False)
    coords = make_cloud(n, extent, D, seed=cin + cout, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    km = mgr._kernel_map(key, key, [3] * D, [1] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(coords.shape[0], cin, generator=g) - 0.5
    gy = torch.rand(coords.shape[0], cout, generator=g) - 0.5
    w = torch.rand(3 ** D, cin, cout, generator=g) - 0.5
    res = {}
    for fuse in ("1", "1", "0"):
        monkeypatch.setattr(MEB, "_BF16_FUSE", fuse)
        km._launch_cache.clear()
        y = MEB._conv_forward(x.to(device), w.to(device), km, "mfma")
        gi = MEB._conv_target(gy.to(device), w.to(device), km, "in", km.n_in, name="d", transposed=True)
        res.setdefault(fuse, []).append((y.clone(), gi.clone()))
    assert torch.equal(res["1"][0][0], res["1"][1][0]) and torch.equal(res["1"][0][1], res["1"][1][1])
    _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(D, 3))
    want_y = O.conv_forward(x.numpy(), w.numpy(), okm, len(coords))
    want_gi = O.conv_backward(x.numpy(), gy.numpy(), w.numpy(), okm)[0]
    for fuse in ("1", "0"):
        assert_close(res[fuse][0][0], want_y, what=f"forward fuse={fuse}")
        assert_close(res[fuse][0][1], want_gi, what=f"grad_in fuse={fuse}")
    assert_close(res["1"][0][0], res["0"][0][0].cpu().numpy(), 5e-5, 5e-5, what="fused vs unfused")


@pytest.mark.parametrize("n,extent,D,cin,cout", [(3000, 9, 4, 32, 64), (3000, 9, 4, 64, 32), (6000, 40, 3, 32, 32),
                                                  (4000, 30, 3, 64, 64), (5000, 40, 3, 64, 128), (3000, 14, 3, 32, 64),
                                                  (40, 10, 3, 32, 32)])
def test_sparse_fp32_launches_on_the_bf16_pipe(device, monkeypatch, n, extent, D, cin, cout):
    """k_conv_tile_f32x3_fused (round 4): the fp32 multi-offset launch of sparse maps with exactly split operands on the
    bf16 matrix pipe, behind me_conv_target_f32_fused (same packed image, same plan).  Forward and input gradient against
    the oracle per element, bitwise reproducible; against float64 ground truth its error is of fp32 order — at most 4x
    the fp32-MFMA kernel's own (me_debug_set_f32_fused_split(0)) and <= 2e-6 of sum |x| |w|."""
    from minkowskiengine_amd import _lib, backend as MEB
    lib = _lib.load()
    if not lib.me_debug_variants_compiled():
        pytest.skip("2x slower than k_conv_tile_f32's fused launch on config 5: in a -DME_DEBUG_VARIANTS build only")
    monkeypatch.setattr(MEB, "_F32_SPLIT", False)        # (the channel-count policy would send 64 -> 128 to the split kernels)
    monkeypatch.setattr(MEB, "_BF16_FUSE", "1")
    coords = make_cloud(n, extent, D, seed=cin + cout, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    km = mgr._kernel_map(key, key, [3] * D, [1] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(coords.shape[0], cin, generator=g) - 0.5
    x[::7] = torch.nextafter(x[::7], torch.full_like(x[::7], 1e30))        # odd low bits
    gy = torch.rand(coords.shape[0], cout, generator=g) - 0.5
    w = torch.rand(3 ** D, cin, cout, generator=g) - 0.5
    res = {}
    try:
        for mode in (1, 1, 0):
            lib.me_debug_set_f32_fused_split(mode)
            y = MEB._conv_forward(x.to(device), w.to(device), km, "mfma")
            gi = MEB._conv_target(gy.to(device), w.to(device), km, "in", km.n_in, name="d", transposed=True)
            res.setdefault(mode, []).append((y.cpu().numpy().astype(np.float64), gi.cpu().numpy().astype(np.float64)))
    finally:
        lib.me_debug_set_f32_fused_split(0)
    assert np.array_equal(res[1][0][0], res[1][1][0]) and np.array_equal(res[1][0][1], res[1][1][1])
    _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(D, 3))
    x64, w64, gy64 = x.numpy().astype(np.float64), w.numpy().astype(np.float64), gy.numpy().astype(np.float64)
    want_y = O.conv_forward(x64, w64, okm, len(coords))
    want_gi = O.conv_backward(x64, gy64, w64, okm)[0]
    mag_y = O.conv_forward(np.abs(x64), np.abs(w64), okm, len(coords))
    mag_gi = O.conv_backward(np.abs(x64), np.abs(gy64), np.abs(w64), okm)[0]
    for got, want, mag, what in ((0, want_y, mag_y, "forward"), (1, want_gi, mag_gi, "grad_in")):
        assert_close(torch.from_numpy(res[1][0][got]), want, what=what)
        err_split = np.abs(res[1][0][got] - want).max() / max(mag.max(), 1e-30)
        err_mfma = np.abs(res[0][0][got] - want).max() / max(mag.max(), 1e-30)
        assert err_split <= 2e-6 and err_split <= 4 * max(err_mfma, 1e-7), (what, err_split, err_mfma)
