"""GPU parity of the bf16 feature path (BASELINE configs[2]: MinkUNet34C in bf16).

The reference has no reduced-precision path, so the oracle is the fp32 reference algorithm applied to the
bf16-ROUNDED operands (features, weights, upstream gradient): products of bf16 values are exact in fp32, so
the only differences are the fp32 summation order and ONE final rounding to bf16 (relative 2^-9 per
element).  Tolerances written below: element-wise |err| <= 2^-8 |ref| + 1e-3 max|ref| for bf16 outputs
(one ulp of slack on top of the half-ulp rounding), 1e-4 relative for the fp32 weight gradient."""
import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from helpers import assert_close, make_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture
def no_split_k():
    """The bit-identity tests compare schedules of the UNSPLIT kernel (slab width, deep pipeline, batch fusion): a
    split-K launch (small maps, round 4) adds its offset groups in another order.  Off for the test, policy after."""
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    lib.me_debug_set_bf16_splitk(0)
    lib.me_debug_set_bf16_ws(0)        # (and they compare schedules of k_conv_tile_bf16, not of the wave-specialised kernel)
    try:
        yield
    finally:
        lib.me_debug_set_bf16_splitk(-1)
        lib.me_debug_set_bf16_ws(-1)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def assert_bf16_close(got, ref, what):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    tol = 2.0 ** -8 * np.abs(ref) + 1e-3 * max(1.0, np.abs(ref).max())
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} elements off, max err {np.abs(got - ref).max():.4g}"


def _run_layer(device, coords, cin, cout, ks, stride=1, dil=1, seed=0, kernel_dtype=torch.float32, transpose=False):
    import minkowskiengine_amd as ME
    D = coords.shape[1] - 1
    g = torch.Generator().manual_seed(seed)
    feats = bf16_round(torch.rand(coords.shape[0], cin, generator=g) - 0.3)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, dilation=dil, dimension=D)
    with torch.no_grad():
        conv.kernel.copy_(bf16_round(torch.rand(conv.kernel.shape, generator=g) - 0.5))
    conv = conv.to(device)
    if kernel_dtype != torch.float32:
        conv = conv.to(kernel_dtype)
    x = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device), requires_grad=True)
    y = conv(x)
    assert y.F.dtype == torch.bfloat16
    gy = bf16_round(torch.rand(y.F.shape, generator=g) - 0.5)
    y.F.backward(gy.to(device).to(torch.bfloat16))
    return conv, x, y, feats, gy


BF16_CASES = [
    # n, extent, D, cin, cout, ks, stride, dil
    (3000, 14, 3, 64, 128, 3, 1, 1),      # config-2 channel shape: KC = 64, two 64-column slabs
    (3000, 40, 3, 64, 128, 3, 1, 1),      # sparse map: nearly empty groups
    (3000, 14, 3, 32, 32, 3, 1, 1),       # MinkUNet full-resolution layers: KC = 32, NC = 32
    (2500, 14, 3, 3, 32, 5, 1, 1),        # MinkUNet stem: cin = 3 (scalar gather), K = 125
    (2500, 14, 3, 32, 96, 3, 1, 1),       # cout = 64 + 32
    (2500, 14, 3, 96, 32, 3, 1, 1),       # cin = 96 inside one 128-channel chunk
    (2000, 12, 3, 192, 128, 3, 1, 1),     # cin = 128 + 64: two chunks, the second half empty
    (1500, 10, 3, 256, 256, 3, 1, 1),     # two full 128-channel chunks, four slabs
    (2500, 14, 3, 20, 24, 3, 1, 1),       # channels not multiples of 8
    (2500, 14, 3, 5, 7, 3, 1, 1),         # odd channel counts: scalar loads and stores
    (2500, 14, 3, 32, 32, 2, 2, 1),       # down conv k=2 s=2
    (2500, 14, 3, 16, 16, 3, 2, 1),
    (2000, 8, 4, 32, 64, 3, 1, 1),        # 4-D, K = 81
    (1, 2, 3, 8, 8, 3, 1, 1),
]


@pytest.mark.parametrize("gather", [True, False], ids=["gather", "plan"])
@pytest.mark.parametrize("n,extent,D,cin,cout,ks,stride,dil", BF16_CASES)
def test_bf16_conv_forward_backward_vs_oracle(device, monkeypatch, n, extent, D, cin, cout, ks, stride, dil, gather):
    """gather: the output-stationary kernel (k_conv_gather_bf16) where the channel counts are multiples of 32, the
    plan kernel elsewhere; plan: the target-stationary plan kernel (k_conv_tile_bf16) for every shape."""
    from minkowskiengine_amd import backend as MEB, _lib
    if gather and not _lib.load().me_debug_variants_compiled():
        pytest.skip("the output-stationary kernel is only in a -DME_DEBUG_VARIANTS build (scripts/build_debug.sh)")
    monkeypatch.setattr(MEB, "_BF16_GATHER", gather)
    coords = make_cloud(n, extent, D, seed=n + cin, batch=2 if n > 100 else 1, negative=True)
    conv, x, y, feats, gy = _run_layer(device, coords, cin, cout, ks, stride, dil)
    in_c = coords.numpy()
    out_c = y.C.cpu().numpy()
    _, km = O.kernel_map(in_c, out_c, O.make_region(D, ks, dil, 1))
    w = conv.kernel.detach().float().cpu().numpy()
    ref = O.conv_forward(feats.numpy(), w, km, len(out_c))
    assert_bf16_close(y.F.detach().float().cpu().numpy(), ref, "forward")
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
    assert x.F.grad.dtype == torch.bfloat16 and conv.kernel.grad.dtype == torch.float32
    assert_bf16_close(x.F.grad.float().cpu().numpy(), gi, "grad_in")
    assert_close(conv.kernel.grad.cpu().numpy(), gw)          # fp32 accumulation of exact products


@pytest.mark.parametrize("n,extent,D,cin,cout,ks,stride,dil", BF16_CASES)
def test_bf16_conv_vs_oracle_on_the_native_host(device, n, extent, D, cin, cout, ks, stride, dil):
    """The same oracle comparison on the shipped default: the native C++ host layer with its own policy (shape, fusion,
    deep pipeline, split-K decisions made in csrc_host/, not in backend.py)."""
    import minkowskiengine_amd as ME
    prev = ME.get_host()
    ME.set_host("native")
    try:
        coords = make_cloud(n, extent, D, seed=n + cin, batch=2 if n > 100 else 1, negative=True)
        conv, x, y, feats, gy = _run_layer(device, coords, cin, cout, ks, stride, dil)
        assert x.coordinate_manager._native
        in_c, out_c = coords.numpy(), y.C.cpu().numpy()
        _, km = O.kernel_map(in_c, out_c, O.make_region(D, ks, dil, 1))
        w = conv.kernel.detach().float().cpu().numpy()
        assert_bf16_close(y.F.detach().float().cpu().numpy(), O.conv_forward(feats.numpy(), w, km, len(out_c)), "forward")
        gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
        assert_bf16_close(x.F.grad.float().cpu().numpy(), gi, "grad_in")
        assert_close(conv.kernel.grad.cpu().numpy(), gw)
    finally:
        ME.set_host(prev)


def test_bf16_kernel_parameter(device, host_layer):
    """Weights stored in bf16 (net.to(torch.bfloat16)) give the same forward bits as fp32 master weights that
    hold bf16-representable values; the weight gradient comes back in the parameter's dtype."""
    coords = make_cloud(3000, 14, 3, seed=3)
    a = _run_layer(device, coords, 64, 64, 3, seed=5)
    b = _run_layer(device, coords, 64, 64, 3, seed=5, kernel_dtype=torch.bfloat16)
    assert torch.equal(a[2].F, b[2].F) and torch.equal(a[1].F.grad, b[1].F.grad)
    assert b[0].kernel.grad.dtype == torch.bfloat16
    assert torch.equal(a[0].kernel.grad.to(torch.bfloat16), b[0].kernel.grad)


def test_bf16_bitwise_reproducible(device, host_layer):
    coords = make_cloud(4000, 14, 3, seed=9)
    r1 = _run_layer(device, coords, 32, 64, 3)
    r2 = _run_layer(device, coords, 32, 64, 3)
    assert torch.equal(r1[2].F, r2[2].F) and torch.equal(r1[1].F.grad, r2[1].F.grad)
    assert torch.equal(r1[0].kernel.grad, r2[0].kernel.grad)


def test_bf16_transposed_conv_and_bias(device, host_layer):
    """Down conv then transposed conv back onto the input map, bias and the 1x1 `use_mm` path in bf16."""
    import minkowskiengine_amd as ME
    coords = make_cloud(3000, 20, 3, seed=11)
    g = torch.Generator().manual_seed(0)
    feats = bf16_round(torch.rand(3000, 32, generator=g))
    down = ME.MinkowskiConvolution(32, 64, kernel_size=2, stride=2, dimension=3)
    up = ME.MinkowskiConvolutionTranspose(64, 32, kernel_size=2, stride=2, dimension=3)
    head = ME.MinkowskiConvolution(32, 20, kernel_size=1, bias=True, dimension=3)
    with torch.no_grad():
        for m in (down, up):
            m.kernel.copy_(bf16_round(torch.rand(m.kernel.shape, generator=g) - 0.5))
    down, up, head = down.to(device), up.to(device), head.to(device)
    x = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device))
    d = down(x)
    u = up(d)
    h = head(u)
    assert u.coordinate_map_key == x.coordinate_map_key and h.F.dtype == torch.bfloat16 and h.F.shape == (3000, 20)
    in_c, mid_c = coords.numpy(), d.C.cpu().numpy()
    _, km = O.kernel_map(in_c, mid_c, O.make_region(3, 2, 1, 1))
    ref_d = O.conv_forward(feats.numpy(), down.kernel.detach().cpu().numpy(), km, len(mid_c))
    assert_bf16_close(d.F.detach().float().cpu().numpy(), ref_d, "down")
    # transposed conv = the same pair lists with the roles swapped (coordinate_map_manager.cpp:763-774)
    kmt = {k: v[::-1].copy() for k, v in km.items()}
    ref_u = O.conv_forward(d.F.detach().float().cpu().numpy(), up.kernel.detach().cpu().numpy(), kmt, len(in_c))
    assert_bf16_close(u.F.detach().float().cpu().numpy(), ref_u, "up")


def test_bf16_config2_full_size(device, host_layer):
    """BASELINE config 2 shape at full size in bf16 + linearity in exact arithmetic: features that are small
    integers keep every product and partial sum exactly representable, so conv(a + b) == conv(a) + conv(b)
    bit for bit once the outputs are integers below 2^8."""
    import minkowskiengine_amd as ME
    coords = make_cloud(100000, 70, 3, seed=0)
    conv, x, y, feats, gy = _run_layer(device, coords, 64, 128, 3)
    _, km = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
    w = conv.kernel.detach().cpu().numpy()
    ref = O.conv_forward(feats.numpy(), w, km, 100000, dtype=np.float32)
    assert_bf16_close(y.F.detach().float().cpu().numpy(), ref, "forward")
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km, dtype=np.float32)
    assert_bf16_close(x.F.grad.float().cpu().numpy(), gi, "grad_in")
    assert_close(conv.kernel.grad.cpu().numpy(), gw)
    g = torch.Generator().manual_seed(1)
    conv1 = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3)
    with torch.no_grad():   # weights in {-1, 0, 1}, sparse enough that |out| stays far below 256
        conv1.kernel.copy_((torch.rand(conv1.kernel.shape, generator=g) < 0.01).float()
                           * torch.sign(torch.rand(conv1.kernel.shape, generator=g) - 0.5))
    conv1 = conv1.to(device)
    a = torch.randint(0, 3, (100000, 64), generator=g).to(device).to(torch.bfloat16)
    b = torch.randint(0, 3, (100000, 64), generator=g).to(device).to(torch.bfloat16)
    mk = lambda f: ME.SparseTensor(f, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
    with torch.no_grad():
        lhs, ra, rb = conv1(mk(a + b)).F, conv1(mk(a)).F, conv1(mk(b)).F
    assert float(lhs.float().abs().max()) < 256
    assert torch.equal(lhs, ra + rb)


@pytest.mark.parametrize("n,extent,cin,cout,ks,D", [(6000, 40, 96, 96, 3, 3), (6000, 40, 32, 64, 3, 3), (4000, 14, 64, 128, 3, 3),
                                                     (3000, 10, 32, 64, 3, 4), (5000, 40, 128, 96, 3, 3)])
def test_batch_fusion_is_bit_identical(device, no_split_k, monkeypatch, n, extent, cin, cout, ks, D):
    """k_conv_tile_bf16 stages consecutive small batches of a tile together (up to four offsets per barrier pair on
    sparse maps); the sub-batches are multiplied and accumulated in the order of the unfused loop, so forward and
    input-gradient results must be BIT-identical with fusion (me_conv_target_bf16_fused) and without (me_conv_target_bf16)."""
    from minkowskiengine_amd import backend as MEB, _lib
    lib = _lib.load()
    coords = make_cloud(n, extent, D, seed=cin + cout, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    km = mgr._kernel_map(key, key, [ks] * D, [1] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(coords.shape[0], cin, generator=g) - 0.5).to(device).bfloat16()
    gy = (torch.rand(coords.shape[0], cout, generator=g) - 0.5).to(device).bfloat16()
    w = (torch.rand(ks ** D, cin, cout, generator=g) - 0.5).to(device)
    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setattr(MEB, "_BF16_FUSE", fuse)
        km._launch_cache.clear()
        y = MEB._conv_forward(x, w, km, "mfma")
        gi = MEB._conv_target(gy, w, km, "in", km.n_in, name="d", transposed=True)
        res[0 if fuse == "1" else 7] = (y.clone(), gi.clone())
    assert torch.equal(res[0][0], res[7][0]) and torch.equal(res[0][1], res[7][1])
    assert float(res[0][0].float().abs().max()) > 0


@pytest.mark.parametrize("n,extent,D,cin,cout,ks,stride,dil", [c for c in BF16_CASES if c[3] % 8 or c[4] % 8])
def test_bf16_odd_channels_without_padding(device, monkeypatch, n, extent, D, cin, cout, ks, stride, dil):
    """Channel counts that are not multiples of 8 are zero-padded around the operator by default (convolution.py
    _pad_channels); with the padding off the kernels' own scalar gather / store paths serve them: same results."""
    from minkowskiengine_amd import convolution as MC
    coords = make_cloud(n, extent, D, seed=n + cin, batch=2, negative=True)
    res = {}
    for pad in (True, False):
        monkeypatch.setattr(MC, "_PAD_CHANNELS", pad)
        conv, x, y, feats, gy = _run_layer(device, coords, cin, cout, ks, stride, dil)
        res[pad] = (y.F.detach().float().cpu().numpy(), x.F.grad.float().cpu().numpy(), conv.kernel.grad.cpu().numpy())
        assert y.F.shape[1] == cout and x.F.grad.shape[1] == cin and conv.kernel.grad.shape[1:] == (cin, cout)
    in_c = coords.numpy()
    out_c = y.C.cpu().numpy()
    _, km = O.kernel_map(in_c, out_c, O.make_region(D, ks, dil, 1))
    w = conv.kernel.detach().float().cpu().numpy()
    ref = O.conv_forward(feats.numpy(), w, km, len(out_c))
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
    for pad in (True, False):
        assert_bf16_close(res[pad][0], ref, f"forward pad={pad}")
        assert_bf16_close(res[pad][1], gi, f"grad_in pad={pad}")
        assert_close(res[pad][2], gw)


@pytest.mark.parametrize("cin,cout,shapes", [(256, 256, ((64, 128), (128, 128))), (128, 256, ((64, 128), (128, 128))),
                                              (64, 128, ((64, 64), (128, 64))), (384, 256, ((64, 128), (128, 128))),
                                              (96, 96, ((64, 96), (96, 96))), (128, 96, ((64, 128), (96, 128))),
                                              (256, 128, ((128, 128), (128, 256)))])
def test_bf16_slab_width_does_not_change_a_bit(device, no_split_k, cin, cout, shapes):
    """96- and 128-column workgroups (six / eight waves; the defaults where they tile the output channels) against
    64-column ones at the same source-channel chunk: the columns of a target row are independent sums in the same
    order, so forward and input gradient are bit-identical; a deeper chunk (256 against 128) regroups the fp32 partial
    sums and stays within the bf16 bar of the oracle, as does the policy's own choice."""
    from minkowskiengine_amd import _lib, backend as MEB
    lib = _lib.load()
    coords = make_cloud(2500, 12, 3, seed=cin + cout, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    g = torch.Generator().manual_seed(3)
    x = bf16_round(torch.rand(coords.shape[0], cin, generator=g) - 0.3)
    gy = bf16_round(torch.rand(coords.shape[0], cout, generator=g) - 0.5)
    w = bf16_round(torch.rand(27, cin, cout, generator=g) - 0.5)
    res = {}
    try:
        for nc, kc in tuple(shapes) + ((0, 0),):
            lib.me_debug_set_bf16_shape(nc, kc)
            # (plans and packed images depend on the shape: a fresh kernel map and fresh weight tensors per setting)
            km = MEB._build_kernel_map(mgr._get(key), mgr._get(key), _lib.make_region(4, 0, [3] * 3, [1] * 3, [1] * 3))
            wd = w.clone().to(device)
            y = MEB._conv_forward(x.to(device).to(torch.bfloat16), wd, km)
            gi = MEB._conv_target(gy.to(device).to(torch.bfloat16), wd, km, "in", km.n_in, name="d", transposed=True)
            res[(nc, kc)] = (y.clone(), gi.clone())
    finally:
        lib.me_debug_set_bf16_shape(0, 0)
    _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(3, 3))
    want_y = O.conv_forward(x.numpy(), w.numpy(), okm, len(coords))
    want_gi = O.conv_backward(x.numpy(), gy.numpy(), w.numpy(), okm)[0]
    a, b = shapes
    if a[1] == b[1]:      # same chunk depth: only the slab width differs (forward; the input gradient swaps the roles)
        assert torch.equal(res[a][0], res[b][0])
    for shp in res:
        assert_bf16_close(res[shp][0].float().cpu().numpy(), want_y, f"forward {shp}")
        assert_bf16_close(res[shp][1].float().cpu().numpy(), want_gi, f"grad_in {shp}")


@pytest.mark.parametrize("n,extent,cin,cout,ks,D", [(5000, 16, 256, 256, 3, 3), (300, 6, 256, 256, 3, 3), (4000, 40, 128, 128, 3, 3),
                                                     (2500, 12, 256, 128, 3, 3), (1500, 8, 384, 256, 3, 3), (70, 3, 128, 256, 3, 3),
                                                     (2000, 9, 256, 256, 2, 4), (900, 30, 512, 256, 3, 3),
                                                     (6000, 40, 96, 96, 3, 3), (4000, 14, 64, 64, 3, 3), (3000, 10, 32, 64, 3, 4),
                                                     (6000, 40, 128, 96, 3, 3), (5000, 30, 32, 32, 3, 3), (3000, 12, 192, 128, 3, 3)])
def test_bf16_deep_pipeline_is_bit_identical(device, no_split_k, n, extent, cin, cout, ks, D):
    """k_conv_tile_bf16<.., DEEP>: gathers and weights of the batch AFTER NEXT in flight (two register sets, loop
    unrolled by two).  Same batches, same MFMAs in the same order: forward and input gradient must be bit-identical with
    the pipeline forced on (1), off (0) and chosen by the policy (-1), on dense and sparse (batch-fused) maps, tiles of
    one batch, an odd and an even number of batches, several source-channel chunks, every slab width."""
    from minkowskiengine_amd import _lib, backend as MEB
    lib = _lib.load()
    coords = make_cloud(n, extent, D, seed=cin + cout + n, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    km = mgr._kernel_map(key, key, [ks] * D, [1] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(coords.shape[0], cin, generator=g) - 0.5).to(device).bfloat16()
    gy = (torch.rand(coords.shape[0], cout, generator=g) - 0.5).to(device).bfloat16()
    w = (torch.rand(ks ** D, cin, cout, generator=g) - 0.5).to(device)
    res = {}
    try:
        # (deep, two stage buffers): TWOBUF (round 4) = the deep pipeline with ONE barrier per batch
        for deep in (1, 0, -1, (1, 1)):
            two = 0
            if isinstance(deep, tuple):
                deep, two = deep
            lib.me_debug_set_bf16_deep(deep)
            lib.me_debug_set_bf16_twobuf(two)
            y = MEB._conv_forward(x, w, km, "mfma")
            gi = MEB._conv_target(gy, w, km, "in", km.n_in, name="d", transposed=True)
            res[(deep, two) if two else deep] = (y.clone(), gi.clone())
    finally:
        lib.me_debug_set_bf16_deep(-1)
        lib.me_debug_set_bf16_twobuf(-1)
    for deep in (1, -1, (1, 1)):
        assert torch.equal(res[deep][0], res[0][0]) and torch.equal(res[deep][1], res[0][1]), deep
    assert torch.isfinite(res[1][0].float()).all() and float(res[1][0].float().abs().max()) > 0


@pytest.mark.parametrize("n,extent,cin,cout,ks,D", [(6000, 30, 128, 128, 3, 3), (300, 6, 256, 256, 3, 3), (4000, 14, 128, 256, 3, 3),
                                                     (2500, 12, 256, 128, 3, 3), (1500, 8, 384, 256, 3, 3), (70, 3, 128, 128, 3, 3),
                                                     (3000, 9, 128, 128, 3, 4), (5000, 16, 192, 128, 2, 3), (12000, 30, 128, 128, 3, 3)])
def test_wgrad_bf16_128_channel_blocks_match_the_64_channel_ones(device, n, extent, cin, cout, ks, D):
    """k_wgrad_bf16<.., MB = 8>: 128 x 128 blocks of grad_w per workgroup (two thirds of the gather bytes per
    multiply-add) against the 64 x 128 blocks.  The pair ranges differ, so the fp32 partial sums regroup: equal to fp32
    rounding (2e-5 of the largest entry), and both match the oracle; grids of more workgroups than the chip holds at
    once (the case that exposed the in-flight-load hazard at the final flush, docs/HISTORY.md 10.9), ranges shorter than a step
    and channel counts that are no multiple of 128 included."""
    from minkowskiengine_amd import backend as MEB, _lib
    lib = _lib.load()
    coords = make_cloud(n, extent, D, seed=cin + cout, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    km = mgr._kernel_map(key, key, [ks] * D, [1] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(8)
    x = bf16_round(torch.rand(coords.shape[0], cin, generator=g) - 0.5)
    gy = bf16_round(torch.rand(coords.shape[0], cout, generator=g) - 0.5)
    w = bf16_round(torch.rand(ks ** D, cin, cout, generator=g) - 0.5)
    res = {}
    try:
        for mb in (4, 8):
            lib.me_debug_set_wgrad_mb(mb)
            km._launch_cache.clear()
            res[mb] = MEB._conv_backward(x.to(device).bfloat16(), gy.to(device).bfloat16(), w.to(device), km, "mfma",
                                         need_grad_in=False)[1].clone()
    finally:
        lib.me_debug_set_wgrad_mb(0)
    scale = float(res[4].abs().max())
    assert float((res[8] - res[4]).abs().max()) <= 2e-5 * scale
    _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(D, ks))
    want = O.conv_backward(x.numpy(), gy.numpy(), w.numpy(), okm)[1]
    assert_close(res[4], want)
    assert_close(res[8], want)


@pytest.mark.parametrize("n,extent,cin,cout,ks,D", [(1500, 10, 256, 256, 3, 3), (900, 8, 128, 256, 3, 3), (2500, 12, 256, 128, 3, 3),
                                                     (300, 6, 256, 256, 3, 3), (5000, 16, 128, 128, 3, 3), (1200, 9, 384, 256, 3, 3),
                                                     (2000, 9, 64, 128, 2, 4), (70, 3, 128, 256, 3, 3)])
def test_bf16_split_k_matches_the_oracle(device, n, extent, cin, cout, ks, D):
    """Split-K launches of k_conv_tile_bf16 (round 4): G offset groups per tile through an fp32 workspace, added in group
    order by k_conv_splitk_reduce.  For every G in 2 .. 8 (and the policy's choice): forward and input gradient within
    the bf16 bar of the oracle, bit-identical from run to run, the tile geometry really split (taller tiles than the
    unsplit plan), and — because groups only regroup fp32 partial sums — within fp32 rounding of the unsplit launch."""
    from minkowskiengine_amd import _lib, backend as MEB
    lib = _lib.load()
    coords = make_cloud(n, extent, D, seed=cin + cout + n, batch=2, negative=True)
    g = torch.Generator().manual_seed(13)
    x = bf16_round(torch.rand(coords.shape[0], cin, generator=g) - 0.5)
    gy = bf16_round(torch.rand(coords.shape[0], cout, generator=g) - 0.5)
    w = bf16_round(torch.rand(ks ** D, cin, cout, generator=g) - 0.5)
    _, okm = O.kernel_map(coords.numpy(), coords.numpy(), O.make_region(D, ks))
    want_y = O.conv_forward(x.numpy(), w.numpy(), okm, len(coords))
    want_gi = O.conv_backward(x.numpy(), gy.numpy(), w.numpy(), okm)[0]
    res, geo = {}, {}
    try:
        for G in (0, 2, 3, 4, 6, 8, -1):
            lib.me_debug_set_bf16_splitk(G)
            mgr = MEB.CoordinateMapManagerGPU_c10()          # (plans depend on the setting: a fresh kernel map each)
            key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
            km = mgr._kernel_map(key, key, [ks] * D, [1] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
            wd = w.clone().to(device)
            runs = []
            for _ in range(2):
                y = MEB._conv_forward(x.to(device).bfloat16(), wd, km, "mfma")
                gi = MEB._conv_target(gy.to(device).bfloat16(), wd, km, "in", km.n_in, name="d", transposed=True)
                runs.append((y.clone(), gi.clone()))
            assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), f"G={G}: not reproducible"
            res[G] = runs[0]
            geo[G] = MEB.plan_config(km.n_out, km.volume, km.n_pairs, cin, cout, True, False, with_split_k=True)
    finally:
        lib.me_debug_set_bf16_splitk(-1)
    assert geo[0][2] == 1
    vol = ks ** D
    for G in (2, 3, 4, 6, 8):
        if cout % 128 == 0:                     # eligible shapes: 128-column slabs
            assert geo[G][2] == min(G, vol, 8) and geo[G][0] > geo[0][0], (G, geo[G], geo[0])
        else:
            assert geo[G][2] == 1
    for G, (y, gi) in res.items():
        assert_bf16_close(y.float().cpu().numpy(), want_y, f"forward G={G}")
        assert_bf16_close(gi.float().cpu().numpy(), want_gi, f"grad_in G={G}")
        # regrouped fp32 sums, one rounding: at most one bf16 ulp from the unsplit launch, and only on a few elements
        d = (y.float() - res[0][0].float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(res[0][0].float().abs().max())
        assert float((d > 0).float().mean()) < 0.05, f"G={G}: too many elements differ from the unsplit launch"


def test_bf16_split_k_policy_and_statistics(device, host_layer, monkeypatch):
    """The policy splits the coarse MinkUNet levels (256 channels on a few thousand voxels) and nothing large; a
    convolution -> batch-norm pair on a split launch takes the tiles' statistics from the REDUCE kernel and equals the
    pair with the epilogue statistics switched off (a pass over the output), on both host layers."""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import backend as MEB, host as H
    t_small = MEB.plan_config(4977, 27, 66569, 256, 256, True, False, with_split_k=True)
    t_large = MEB.plan_config(79572, 27, 718104, 128, 128, True, False, with_split_k=True)
    t_mid = MEB.plan_config(4977, 27, 66569, 128, 256, True, False, with_split_k=True)      # 128-channel chunks: no gain
    assert t_small[2] >= 2 and t_small[0] > 60 and t_large[2] == 1 and t_mid[2] == 1, (t_small, t_large, t_mid)
    coords = make_cloud(2500, 12, 3, seed=21, batch=2).to(device)
    assert MEB.plan_config(coords.shape[0], 27, 60000, 256, 256, True, False, with_split_k=True)[2] >= 2
    f = (torch.rand(coords.shape[0], 256, generator=torch.Generator().manual_seed(3)) - 0.5).to(device).bfloat16()
    outs = {}
    try:
        for stats in (True, False):
            monkeypatch.setattr(MEB, "_CONV_BN_STATS", stats)
            if ME.is_native():
                H.native_module().set_conv_bn_stats(1 if stats else 0)
            torch.manual_seed(0)
            conv = ME.MinkowskiConvolution(256, 256, kernel_size=3, dimension=3).to(device)
            bn = ME.MinkowskiBatchNorm(256).to(device)
            y = bn(conv(ME.SparseTensor(f, coords)))
            outs[stats] = (y.F.detach().float().clone(), bn.bn.running_mean.clone(), bn.bn.running_var.clone())
    finally:
        if ME.is_native():
            H.native_module().set_conv_bn_stats(-1)
    a, b = outs[True], outs[False]
    scale = float(b[0].abs().max())
    assert float((a[0] - b[0]).abs().max()) <= 2.0 ** -7 * scale
    assert float((a[1] - b[1]).abs().max()) <= 1e-5 * scale + 1e-6
    assert float(((a[2] - b[2]) / b[2]).abs().max()) <= 1e-4


@pytest.mark.parametrize("n,extent,cin,cout,ks,stride,D", [
    (5000, 16, 128, 128, 3, 1, 3), (6000, 40, 96, 96, 3, 1, 3), (4000, 14, 64, 64, 3, 1, 3), (5000, 30, 32, 32, 3, 1, 3),
    (3000, 12, 192, 128, 3, 1, 3), (2500, 12, 256, 256, 3, 1, 3), (1500, 8, 384, 256, 3, 1, 3), (6000, 40, 128, 96, 3, 1, 3),
    (3000, 10, 32, 64, 3, 1, 4), (4000, 14, 64, 128, 2, 2, 3), (70, 3, 128, 256, 3, 1, 3), (1, 2, 32, 32, 3, 1, 3),
    (9000, 60, 64, 64, 3, 1, 3)])
def test_offset_synchronous_kernel_is_bit_identical(device, no_split_k, n, extent, cin, cout, ks, stride, D):
    """k_conv_off_bf16 (round 4): the waves of a workgroup split an item by rows (their 16 gathered rows go straight into
    the MFMA operand, the offset's weight slice sits in LDS, one barrier per offset) instead of by columns.  The same
    additions in the same order as k_conv_tile_bf16 on the same plan: forward and input gradient are BIT-IDENTICAL in
    both wave shapes, on dense, sparse, strided, 4-D, several-chunk and single-row maps — and both match the oracle."""
    from minkowskiengine_amd import _lib, backend as MEB
    lib = _lib.load()
    if not lib.me_debug_variants_compiled():
        pytest.skip("k_conv_off_bf16 measured slower than the column-split kernel (profiles/r04_offsync_schedule_sweep.log): "
                    "it is only in a -DME_DEBUG_VARIANTS build (scripts/build_debug.sh)")
    coords = make_cloud(n, extent, D, seed=cin + cout + n, batch=2 if n > 100 else 1, negative=True)
    g = torch.Generator().manual_seed(17)
    x = bf16_round(torch.rand(coords.shape[0], cin, generator=g) - 0.5)
    w = bf16_round(torch.rand(ks ** D, cin, cout, generator=g) - 0.5)
    res = {}
    try:
        for mode in (0, 1, 2):
            lib.me_debug_set_bf16_offsync(mode)
            # the offset-synchronous kernel keeps 256-channel layers in 128-channel chunks: the column-split kernel is
            # held to the same chunking (another chunk depth regroups the fp32 sums)
            lib.me_debug_set_bf16_shape(0, 128 if (mode == 0 and cin % 256 == 0) or (mode == 0 and cout % 256 == 0) else 0)
            mgr = MEB.CoordinateMapManagerGPU_c10()
            key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
            okey = mgr.stride(key, [stride] * D, "")
            km = mgr._kernel_map(key, okey, [ks] * D, [stride] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
            wd = w.clone().to(device)
            gy = bf16_round(torch.rand(km.n_out, cout, generator=torch.Generator().manual_seed(5)) - 0.5)
            y = MEB._conv_forward(x.to(device).bfloat16(), wd, km, "mfma")
            gi = MEB._conv_target(gy.to(device).bfloat16(), wd, km, "in", km.n_in, name="d", transposed=True)
            res[mode] = (y.clone(), gi.clone(), mgr.get_coordinates(okey).cpu().numpy(), gy)
    finally:
        lib.me_debug_set_bf16_offsync(0)
        lib.me_debug_set_bf16_shape(0, 0)
    for mode in (1, 2):
        assert torch.equal(res[mode][0], res[0][0]), f"forward, wave shape {mode}"
        assert torch.equal(res[mode][1], res[0][1]), f"input gradient, wave shape {mode}"
    in_c, out_c = coords.numpy(), res[0][2]
    _, okm = O.kernel_map(in_c, out_c, O.make_region(D, ks, 1, 1))
    assert_bf16_close(res[1][0].float().cpu().numpy(), O.conv_forward(x.numpy(), w.numpy(), okm, len(out_c)), "forward")
    assert_bf16_close(res[1][1].float().cpu().numpy(), O.conv_backward(x.numpy(), res[0][3].numpy(), w.numpy(), okm)[0],
                      "grad_in")


@pytest.mark.parametrize("n,extent,cin,cout", [(6000, 30, 96, 96), (5000, 16, 128, 96), (4000, 14, 64, 128), (2500, 12, 192, 128)])
def test_bf16_tile_order_and_dispatch_do_not_change_a_bit(device, no_split_k, host_layer, monkeypatch, n, extent, cin, cout):
    """Round 4: bf16 launches whose source matrix is larger than the L2s take spatially compact tiles
    (ME_AMD_TILE_SPATIAL_SRC_MB; forced here with a threshold of one byte), and the tiles of a plan can be dealt to the
    XCDs in contiguous chunks (me_debug_set_tile_dispatch).  Both only regroup target rows into tiles / permute the
    dispatch: forward and both gradients of a layer are bit-identical to row tiles in heaviest-first order, on both
    hosts — and the dispatch table stays a permutation of the tiles."""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import _lib, backend as MEB, host as H
    lib = _lib.load()
    native = H.native_module() if host_layer == "native" else None
    coords = make_cloud(n, extent, 3, seed=n + cin, batch=2, negative=True)
    res = {}

    def policy(src_bytes):
        monkeypatch.setattr(MEB, "_TILE_SPATIAL_MIN_SRC_BYTES", src_bytes)
        if native is not None:
            native.set_policy("tile_spatial_src_bytes", src_bytes)
    try:
        for name, src_bytes, dispatch in (("rows", 0, 0), ("spatial", 1, 0), ("rows_xcd", 0, 1), ("spatial_xcd", 1, 1)):
            policy(src_bytes)
            lib.me_debug_set_tile_dispatch(dispatch)
            conv, x, y, feats, gy = _run_layer(device, coords, cin, cout, 3, seed=3)
            res[name] = (y.F.clone(), x.F.grad.clone(), conv.kernel.grad.clone())
            if host_layer == "python":
                km = x.coordinate_manager._manager._kernel_map(x.coordinate_map_key, y.coordinate_map_key, [3] * 3, [1] * 3,
                                                               [1] * 3, MEB.RegionType.HYPER_CUBE, None, False, False)
                _, cfg = MEB._conv_launch_cfg(km, "out", km.n_out, cin, cout, True)
                assert (cfg[6] is not None) == (src_bytes > 0)          # an order tensor <=> spatial tiles on a flat map
                bptr = cfg[5].cpu().numpy()
                n_tiles = -(-km.n_out // cfg[0])
                assert sorted(bptr[n_tiles + 1:2 * n_tiles + 1].tolist()) == list(range(n_tiles))
    finally:
        lib.me_debug_set_tile_dispatch(0)
        if native is not None:
            native.set_policy("tile_spatial_src_bytes", 28 << 20)
    for name in ("spatial", "rows_xcd", "spatial_xcd"):
        for got, want, what in zip(res[name], res["rows"], ("forward", "grad_in", "grad_kernel")):
            assert torch.equal(got, want), (name, what)
    assert float(res["rows"][0].float().abs().max()) > 0


WS_CASES = [
    # n, extent, cin, cout, ks, D, stride
    (6000, 16, 96, 96, 3, 3, 1),       # MinkUNet's widest layers on its largest maps: 96-channel chunks, 64 + 32 columns
    (5000, 14, 64, 128, 3, 3, 1),      # config 2: one 128-column slab
    (5000, 14, 128, 64, 3, 3, 1),      # its input gradient shape
    (4000, 12, 128, 128, 3, 3, 1),
    (4000, 12, 64, 64, 3, 3, 1),
    (3000, 12, 192, 128, 3, 3, 1),     # two 96-channel chunks
    (3000, 12, 384, 256, 3, 3, 1),     # three 128-channel chunks, two slabs
    (4000, 14, 32, 64, 3, 3, 1),       # one 32-channel step
    (4000, 14, 128, 96, 3, 3, 1),
    (3000, 12, 64, 128, 2, 3, 2),      # strided map: targets != sources
    (2000, 8, 64, 64, 3, 4, 1),        # 4-D
    (70, 3, 128, 128, 3, 3, 1),        # one tile, a handful of batches
    (1, 2, 64, 64, 3, 3, 1),
]


@pytest.mark.parametrize("n,extent,cin,cout,ks,D,stride", WS_CASES)
@pytest.mark.parametrize("spatial", [False, True], ids=["rows", "spatial"])
def test_bf16_wave_specialised_kernel_is_bit_identical(device, monkeypatch, n, extent, cin, cout, ks, D, stride, spatial):
    """k_conv_tile_bf16_ws (round 4: producer waves / multiplier waves, two stage buffers, one barrier per batch, counted
    LDS waits) walks the same plan with the same packed weights and adds in the same order as k_conv_tile_bf16: forward,
    input gradient and the tile statistics' consumers must see the same bits — on row tiles and spatial tiles, one to
    three source chunks, one and two column slabs, tiles of one batch."""
    from minkowskiengine_amd import _lib, backend as MEB
    lib = _lib.load()
    monkeypatch.setattr(MEB, "_BF16_FUSE", "0")             # (multi-offset batches stay with k_conv_tile_bf16)
    monkeypatch.setattr(MEB, "_TILE_ORDER", "spatial" if spatial else "rows")
    coords = make_cloud(n, extent, D, seed=cin + cout + n, batch=2, negative=True)
    g = torch.Generator().manual_seed(5)
    w = (torch.rand(ks ** D, cin, cout, generator=g) - 0.5).to(device)
    res = {}
    try:
        for mode in (0, 1, 8):                      # lock-step kernel | four multiplier waves | eight (128-column slabs)
            lib.me_debug_set_bf16_ws(min(mode, 1))
            lib.me_debug_set_bf16_ws_ncw(8 if mode == 8 else 4)
            lib.me_debug_set_bf16_splitk(0)
            mgr = MEB.CoordinateMapManagerGPU_c10()
            key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
            okey = mgr.stride(key, [stride] * D) if stride > 1 else key
            km = mgr._kernel_map(key, okey, [ks] * D, [stride] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
            gx = torch.Generator().manual_seed(6)
            x = (torch.rand(km.n_in, cin, generator=gx) - 0.5).to(device).bfloat16()
            gy = (torch.rand(km.n_out, cout, generator=gx) - 0.5).to(device).bfloat16()
            y = MEB._conv_forward(x, w, km, "mfma")
            gi = MEB._conv_target(gy, w, km, "in", km.n_in, name="d", transposed=True)
            res[mode] = (y.clone(), gi.clone())
    finally:
        lib.me_debug_set_bf16_ws(-1)
        lib.me_debug_set_bf16_ws_ncw(0)
        lib.me_debug_set_bf16_splitk(-1)
    for mode in (1, 8):
        assert torch.equal(res[0][0], res[mode][0]), ("forward", mode)
        assert torch.equal(res[0][1], res[mode][1]), ("input gradient", mode)
    assert torch.isfinite(res[1][0].float()).all() and (n < 10 or float(res[1][0].float().abs().max()) > 0)


def test_bf16_wave_specialised_kernel_statistics(device, host_layer):
    """conv -> batch norm through the module path with the wave-specialised kernel's statistics epilogue: the batch
    norm's output equals the one computed from a pass over the convolution's output (1e-5 relative: regrouped fp32
    partial sums), and the convolution's output does not depend on the epilogue."""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    coords = make_cloud(6000, 16, 3, seed=9, batch=2).to(device)
    feats = (torch.rand(coords.shape[0], 64, generator=torch.Generator().manual_seed(1)) - 0.5).to(device).bfloat16()
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).to(device)
    bn = ME.MinkowskiBatchNorm(128).to(device)
    out = {}
    try:
        for mode in (0, 1):
            lib.me_debug_set_bf16_ws(mode)
            bn.bn.reset_running_stats()
            x = ME.SparseTensor(feats, coords)
            y = conv(x)
            z = bn(y)
            out[mode] = (y.F.clone(), z.F.clone(), bn.bn.running_var.clone())
    finally:
        lib.me_debug_set_bf16_ws(-1)
    assert torch.equal(out[0][0], out[1][0])
    assert torch.allclose(out[0][1].float(), out[1][1].float(), rtol=2e-2, atol=2e-2)     # bf16 outputs of the norm
    assert torch.allclose(out[0][2], out[1][2], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("n,extent,cin,cout,ks,D", [(6000, 40, 96, 96, 3, 3), (6000, 40, 128, 96, 3, 3), (5000, 30, 64, 64, 3, 3),
                                                     (4000, 40, 64, 128, 3, 3), (4000, 30, 32, 64, 3, 3), (3000, 10, 64, 64, 3, 4),
                                                     (3000, 24, 192, 128, 3, 3), (5000, 16, 96, 96, 3, 3), (40, 12, 64, 64, 3, 3)])
@pytest.mark.parametrize("spatial", [False, True], ids=["rows", "spatial"])
def test_bf16_wave_specialised_batch_fusion_is_bit_identical(device, monkeypatch, n, extent, cin, cout, ks, D, spatial):
    """k_conv_tile_bf16_ws<.., FUSE>: on sparse maps runs of single-group batches of consecutive offsets are staged and
    multiplied together (up to four offsets per barrier, each group with its own weights), producers and multipliers
    walking the same super-batch sequence.  Forward and input gradient must equal k_conv_tile_bf16's fused launch bit for
    bit — sparse and dense maps (where nothing fuses), K = 27 and 81, one and two chunks, one and two slabs."""
    from minkowskiengine_amd import _lib, backend as MEB
    lib = _lib.load()
    if not lib.me_debug_variants_compiled():
        pytest.skip("measured slower than k_conv_tile_bf16's fused launch: instantiated in a -DME_DEBUG_VARIANTS build only")
    monkeypatch.setattr(MEB, "_BF16_FUSE", "1")
    monkeypatch.setattr(MEB, "_TILE_ORDER", "spatial" if spatial else "rows")
    coords = make_cloud(n, extent, D, seed=cin + cout + n, batch=2, negative=True)
    g = torch.Generator().manual_seed(5)
    w = (torch.rand(ks ** D, cin, cout, generator=g) - 0.5).to(device)
    res = {}
    try:
        for mode in (0, 1):
            lib.me_debug_set_bf16_ws_fuse(mode)
            lib.me_debug_set_bf16_splitk(0)
            mgr = MEB.CoordinateMapManagerGPU_c10()
            key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
            km = mgr._kernel_map(key, key, [ks] * D, [1] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
            gx = torch.Generator().manual_seed(6)
            x = (torch.rand(km.n_in, cin, generator=gx) - 0.5).to(device).bfloat16()
            gy = (torch.rand(km.n_out, cout, generator=gx) - 0.5).to(device).bfloat16()
            y = MEB._conv_forward(x, w, km, "mfma")
            gi = MEB._conv_target(gy, w, km, "in", km.n_in, name="d", transposed=True)
            res[mode] = (y.clone(), gi.clone())
    finally:
        lib.me_debug_set_bf16_ws_fuse(0)
        lib.me_debug_set_bf16_splitk(-1)
    assert torch.equal(res[0][0], res[1][0]), "forward"
    assert torch.equal(res[0][1], res[1][1]), "input gradient"
    assert torch.isfinite(res[1][0].float()).all() and float(res[1][0].float().abs().max()) > 0


@pytest.mark.parametrize("n,extent,cin,cout,ks,D,stride", [(6000, 16, 96, 96, 3, 3, 1), (5000, 14, 64, 128, 3, 3, 1), (4000, 12, 128, 128, 3, 3, 1),
                                                            (3000, 12, 256, 256, 3, 3, 1), (3000, 12, 384, 256, 3, 3, 1), (4000, 30, 32, 32, 3, 3, 1),
                                                            (4000, 14, 64, 64, 2, 3, 2), (2000, 8, 32, 64, 3, 4, 1), (300, 6, 256, 128, 3, 3, 1),
                                                            (20000, 40, 64, 64, 3, 3, 1), (1, 2, 64, 64, 3, 3, 1)])
def test_bf16_wave_specialised_weight_gradient_is_bit_identical(device, n, extent, cin, cout, ks, D, stride):
    """k_wgrad_bf16_ws (round 4: producer waves stage the rows of the next steps while multiplier waves run the MFMAs of
    this one) keeps the pair ranges, the slots and the order of additions of k_wgrad_bf16: grad_w must be the same bits —
    64- and 128-channel blocks, one and two output blocks per wave, ranges that end inside an offset, more ranges than
    pairs / 64, a grid larger than the chip, and both producer depths."""
    from minkowskiengine_amd import _lib, backend as MEB
    lib = _lib.load()
    if not lib.me_debug_variants_compiled():
        pytest.skip("not faster than k_wgrad_bf16 (profiles/r04_wgrad_ws_sweep.log): in a -DME_DEBUG_VARIANTS build only")
    coords = make_cloud(n, extent, D, seed=cin + cout + n, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    okey = mgr.stride(key, [stride] * D) if stride > 1 else key
    km = mgr._kernel_map(key, okey, [ks] * D, [stride] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(km.n_in, cin, generator=g) - 0.5).to(device).bfloat16()
    gy = (torch.rand(km.n_out, cout, generator=g) - 0.5).to(device).bfloat16()
    w = (torch.rand(ks ** D, cin, cout, generator=g) - 0.5).to(device)
    res = {}
    try:
        for mode in (0, 1, 2):
            lib.me_debug_set_wgrad_ws(mode)
            _, gw = MEB._conv_backward(x, gy, w, km, "mfma", need_grad_in=False)
            res[mode] = gw.clone()
    finally:
        lib.me_debug_set_wgrad_ws(0)
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
    assert torch.isfinite(res[1]).all() and (n < 10 or float(res[1].abs().max()) > 0)
