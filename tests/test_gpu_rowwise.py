"""GPU parity of the row-wise launches (round 6, csrc/conv_rowwise.hip): kernel-map sides with exactly one pair per target
row — K = 1 layers (the reference's `input.F.mm(kernel)`, MinkowskiEngine/MinkowskiConvolution.py:304-308) and the fine
side of kernel_size == stride maps (forward of the transposed layer, input gradient of the strided one).

Oracle: the fp32 reference algorithm (oracle/me_oracle.py, pinned to the compiled reference) on the bf16-ROUNDED operands;
tolerance as tests/test_gpu_bf16.py: |err| <= 2^-8 |ref| + 1e-3 max|ref| per element for bf16 outputs (one rounding of an
fp32 sum), 1e-4 relative for the fp32 weight gradient.  Every test runs on both host layers and checks that the launch
side really took the row-wise kernel (a side on the tile-plan kernels would pass the oracle too)."""
import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from helpers import assert_close, make_cloud
from test_gpu_bf16 import assert_bf16_close, bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def rowwise_also_with_wanted_statistics():
    """The layers of these tests are in training mode (a batch norm MAY follow: statistics wanted); the policy would keep
    small K = 1 forward launches on the tile-plan kernel.  Here the row-wise kernel is the subject: threshold 1 row."""
    from minkowskiengine_amd import backend as MEB, host as H
    prev = MEB._ROWWISE_MIN_ROWS_WITH_STATS
    MEB._ROWWISE_MIN_ROWS_WITH_STATS = 1
    if H.native_module() is not None:
        H.native_module().set_policy("rowwise_min_rows_with_stats", 1)
    try:
        yield
    finally:
        MEB._ROWWISE_MIN_ROWS_WITH_STATS = prev
        if H.native_module() is not None:
            H.native_module().set_policy("rowwise_min_rows_with_stats", 100000)


def _cfg_is_rowwise(x_in, y_out, ks, stride, target, c_src, c_dst, transpose=False):
    """does the (kernel map side, shape) of this launch run on the row-wise kernel under the host in charge?"""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import backend as MEB
    mgr = x_in.coordinate_manager._manager
    D = x_in.D
    if ME.is_native():
        t, g = mgr._conv_cfg(x_in.coordinate_map_key, y_out.coordinate_map_key, [ks] * D, [stride] * D, [1] * D, 0, transpose,
                             target, c_src, c_dst, True)
        return (t, g) == (0, 0)
    km = mgr._kernel_map(x_in.coordinate_map_key, y_out.coordinate_map_key, [ks] * D, [stride] * D, [1] * D,
                         ME.RegionType.HYPER_CUBE, None, transpose, False)
    n_tgt = km.n_out if target == "out" else km.n_in
    return MEB._rowwise_cfg(km, target, n_tgt, c_src, c_dst) is not None


K1_CASES = [
    # n, extent, cin, cout
    (3000, 14, 32, 64),       # one 32-channel step, 64 columns
    (3000, 14, 64, 128),      # eight column blocks
    (2500, 14, 128, 96),      # MinkUNet decoder: six column blocks
    (2500, 14, 96, 24),       # the segmentation head (20 classes padded to 24): two blocks, the second half empty
    (2000, 12, 192, 128),     # 128-channel chunks, the second half full of padding
    (1500, 10, 384, 256),     # twelve steps, two 128-column slabs
    (1500, 10, 128, 256),
    (70, 4, 32, 32),          # a single, partial item
    (1, 2, 8, 8),
]


@pytest.mark.parametrize("n,extent,cin,cout", K1_CASES)
def test_rowwise_k1_layer_vs_oracle(device, host_layer, n, extent, cin, cout):
    """kernel_size = 1, stride 1: out = F @ W through the row-wise kernel, forward and input gradient; the weight gradient
    (pair-list kernel, unchanged) rides along"""
    import minkowskiengine_amd as ME
    coords = make_cloud(n, extent, 3, seed=n + cin, batch=2 if n > 100 else 1, negative=True)
    g = torch.Generator().manual_seed(cin + cout)
    feats = bf16_round(torch.rand(coords.shape[0], cin, generator=g) - 0.3)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=1, dimension=3)
    with torch.no_grad():
        conv.kernel.copy_(bf16_round(torch.rand(conv.kernel.shape, generator=g) - 0.5))
    conv = conv.to(device)
    x = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device), requires_grad=True)
    y = conv(x)
    gy = bf16_round(torch.rand(y.F.shape, generator=g) - 0.5)
    y.F.backward(gy.to(device).to(torch.bfloat16))
    assert y.coordinate_map_key == x.coordinate_map_key and y.F.dtype == torch.bfloat16
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    # (a shape whose W[k] slab exceeds the kernel's LDS policy — 384 -> 256 — stays on the tile-plan kernels: same oracle)
    assert _cfg_is_rowwise(x, y, 1, 1, "out", cin, cout) == bool(lib.me_conv_rowwise_supported_bf16(1, cin, cout)), \
        "forward: row-wise kernel not taken / taken for an unsupported shape"
    assert _cfg_is_rowwise(x, y, 1, 1, "in", cout, cin) == bool(lib.me_conv_rowwise_supported_bf16(1, cout, cin)), \
        "input gradient: row-wise kernel not taken / taken for an unsupported shape"
    w = conv.kernel.detach().float().cpu().numpy()
    f64 = feats.numpy().astype(np.float64)
    assert_bf16_close(y.F.detach().float().cpu().numpy(), f64 @ w.astype(np.float64), "forward")
    assert_bf16_close(x.F.grad.float().cpu().numpy(), gy.numpy().astype(np.float64) @ w.astype(np.float64).T, "grad_in")
    assert_close(conv.kernel.grad.cpu().numpy(), (f64.T @ gy.numpy().astype(np.float64)))


S2_CASES = [
    # n, extent, c_fine, c_coarse
    (3000, 20, 32, 32),       # MinkUNet conv1 / conv2: 32 -> 32 down
    (3000, 20, 32, 64),
    (2500, 16, 96, 128),      # decoder: 128 -> 96 up
    (2000, 14, 128, 256),     # 256 -> 128 up, two slabs on the way down
    (1500, 12, 256, 256),
    (300, 40, 64, 64),        # sparse: one child per parent, offsets with a handful of pairs
]


@pytest.mark.parametrize("n,extent,c_fine,c_coarse", S2_CASES)
def test_rowwise_stride2_down_dgrad_and_up_forward_vs_oracle(device, host_layer, n, extent, c_fine, c_coarse):
    """k = 2, s = 2 down convolution followed by the transposed convolution back onto the input map: the FINE side of
    the shared kernel map has one pair per row — the up layer's forward launch and the down layer's input gradient run
    row-wise; the coarse side (down forward, up input gradient) stays on the tile-plan kernels.  All four against the
    oracle, plus both weight gradients."""
    import minkowskiengine_amd as ME
    coords = make_cloud(n, extent, 3, seed=n + c_fine, batch=2, negative=True)
    g = torch.Generator().manual_seed(c_fine * 3 + c_coarse)
    feats = bf16_round(torch.rand(coords.shape[0], c_fine, generator=g) - 0.4)
    down = ME.MinkowskiConvolution(c_fine, c_coarse, kernel_size=2, stride=2, dimension=3)
    up = ME.MinkowskiConvolutionTranspose(c_coarse, c_fine, kernel_size=2, stride=2, dimension=3)
    with torch.no_grad():
        for m in (down, up):
            m.kernel.copy_(bf16_round((torch.rand(m.kernel.shape, generator=g) - 0.5) * 0.5))
    down, up = down.to(device), up.to(device)
    x = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device), requires_grad=True)
    d = down(x)
    d_leaf = ME.SparseTensor(d.F.detach().requires_grad_(True), coordinate_map_key=d.coordinate_map_key,
                             coordinate_manager=d.coordinate_manager)
    u = up(d_leaf)
    assert u.coordinate_map_key == x.coordinate_map_key
    gu = bf16_round(torch.rand(u.F.shape, generator=g) - 0.5)
    u.F.backward(gu.to(device).to(torch.bfloat16))
    gd = bf16_round(torch.rand(d.F.shape, generator=g) - 0.5)
    d.F.backward(gd.to(device).to(torch.bfloat16))
    from minkowskiengine_amd import _lib
    sup = bool(_lib.load().me_conv_rowwise_supported_bf16(8, c_coarse, c_fine))
    assert _cfg_is_rowwise(x, d, 2, 2, "in", c_coarse, c_fine) == sup, "down input gradient: row-wise kernel not taken"
    assert not _cfg_is_rowwise(x, d, 2, 2, "out", c_fine, c_coarse)
    assert _cfg_is_rowwise(d, u, 2, 2, "out", c_coarse, c_fine, transpose=True) == sup, "up forward: row-wise kernel not taken"
    in_c, mid_c = coords.numpy(), d.C.cpu().numpy()
    _, km = O.kernel_map(in_c, mid_c, O.make_region(3, 2, 1, 1))
    wd = down.kernel.detach().float().cpu().numpy()
    wu = up.kernel.detach().float().cpu().numpy()
    assert_bf16_close(d.F.detach().float().cpu().numpy(), O.conv_forward(feats.numpy(), wd, km, len(mid_c)), "down forward")
    gi, gw = O.conv_backward(feats.numpy(), gd.numpy(), wd, km)
    assert_bf16_close(x.F.grad.float().cpu().numpy(), gi, "down grad_in (row-wise)")
    assert_close(down.kernel.grad.cpu().numpy(), gw)
    kmt = {k: v[::-1].copy() for k, v in km.items()}      # transposed = the same pair lists, roles swapped
    dF = d.F.detach().float().cpu().numpy()
    assert_bf16_close(u.F.detach().float().cpu().numpy(), O.conv_forward(dF, wu, kmt, len(in_c)), "up forward (row-wise)")
    gi_u, gw_u = O.conv_backward(dF, gu.numpy(), wu, kmt)
    assert_bf16_close(d_leaf.F.grad.float().cpu().numpy(), gi_u, "up grad_in")
    assert_close(up.kernel.grad.cpu().numpy(), gw_u)


def test_rowwise_large_map_two_group_items_and_reproducible(device, host_layer):
    """140k rows: 128-pair items (two row groups per wave) — against the oracle, bitwise run to run, and equal to the
    64-pair-item kernel's bits on the same rows (the per-row sum does not depend on the item shape)"""
    import minkowskiengine_amd as ME
    n, cin, cout = 140000, 128, 96
    coords = make_cloud(n, 80, 3, seed=5)
    g = torch.Generator().manual_seed(1)
    feats = bf16_round(torch.rand(n, cin, generator=g) - 0.3)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=1, dimension=3)
    with torch.no_grad():
        conv.kernel.copy_(bf16_round(torch.rand(conv.kernel.shape, generator=g) - 0.5))
    conv = conv.to(device)
    outs = []
    for _ in range(2):
        x = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device))
        outs.append(conv(x).F.detach().clone())
    assert torch.equal(outs[0], outs[1])
    ref = feats.numpy().astype(np.float64) @ conv.kernel.detach().float().cpu().numpy().astype(np.float64)
    assert_bf16_close(outs[0].float().cpu().numpy(), ref, "forward 140k")
    small = ME.SparseTensor(feats[:3000].to(device).to(torch.bfloat16), coords[:3000].to(device))
    assert torch.equal(conv(small).F.detach(), outs[0][:3000])


def test_rowwise_switch_and_fallbacks(device, host_layer, monkeypatch):
    """ME_AMD_ROWWISE=0 keeps such sides on the tile-plan kernels (same oracle, other bits allowed); a 3^3 map is never
    row-wise; a pruned coarse map (fine rows WITHOUT a parent: fewer pairs than rows) falls back to the tile plan and still
    writes zeros into the orphaned rows"""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import backend as MEB, host
    coords = make_cloud(2500, 14, 3, seed=21, batch=2)
    g = torch.Generator().manual_seed(3)
    feats = bf16_round(torch.rand(coords.shape[0], 64, generator=g) - 0.3)
    conv = ME.MinkowskiConvolution(64, 64, kernel_size=1, dimension=3)
    conv3 = ME.MinkowskiConvolution(64, 64, kernel_size=3, dimension=3)
    with torch.no_grad():
        conv.kernel.copy_(bf16_round(torch.rand(conv.kernel.shape, generator=g) - 0.5))
    conv, conv3 = conv.to(device), conv3.to(device)
    x = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device))
    y_on = conv(x).F.detach().clone()
    assert not _cfg_is_rowwise(x, conv3(x), 3, 1, "out", 64, 64)
    # off: policy switch of either host
    if host_layer == "native":
        host.native_module().set_policy("rowwise", 0)
    else:
        monkeypatch.setattr(MEB, "_ROWWISE", False)
    try:
        x2 = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device))
        y_off = conv(x2).F.detach().clone()
        assert not _cfg_is_rowwise(x2, x2, 1, 1, "out", 64, 64)
    finally:
        if host_layer == "native":
            host.native_module().set_policy("rowwise", 1)
    ref = feats.numpy().astype(np.float64) @ conv.kernel.detach().float().cpu().numpy().astype(np.float64)
    assert_bf16_close(y_on.float().cpu().numpy(), ref, "row-wise")
    assert_bf16_close(y_off.float().cpu().numpy(), ref, "tile plan")
    # orphaned fine rows: up-sample from a PRUNED coarse map onto the full fine map
    down = ME.MinkowskiConvolution(64, 64, kernel_size=2, stride=2, dimension=3).to(device)
    up = ME.MinkowskiConvolutionTranspose(64, 64, kernel_size=2, stride=2, dimension=3).to(device)
    x3 = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device))
    d = down(x3)
    keep = torch.ones(len(d.C), dtype=torch.bool, device=device)
    keep[::3] = False
    dp = ME.MinkowskiPruning()(d, keep)
    u = up(dp, x3.coordinate_map_key)
    assert not _cfg_is_rowwise(dp, u, 2, 2, "out", 64, 64, transpose=True)
    _, km = O.kernel_map(coords.numpy(), dp.C.cpu().numpy(), O.make_region(3, 2, 1, 1))
    kmt = {k: v[::-1].copy() for k, v in km.items()}
    ref_u = O.conv_forward(dp.F.detach().float().cpu().numpy(), up.kernel.detach().float().cpu().numpy(), kmt, len(coords))
    assert_bf16_close(u.F.detach().float().cpu().numpy(), ref_u, "up from a pruned map")
    assert (np.abs(ref_u).sum(1) == 0).any(), "the case must contain orphaned rows"


def test_rowwise_k1_forward_with_wanted_statistics_policy(device, host_layer):
    """A K = 1 forward launch in TRAINING mode (a batch norm may follow: its statistics are wanted) keeps the tile-plan
    kernel — whose epilogue leaves them behind — on small maps and takes the row-wise kernel (the batch norm then reads
    the output once more) from Policy.rowwise_min_rows_with_stats rows on; the conv -> batch-norm pair gives the same
    result either way up to the bf16 rounding of the normalised output."""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import backend as MEB, host as H
    coords = make_cloud(6000, 30, 3, seed=5, batch=2).to(device)
    g = torch.Generator().manual_seed(9)
    f0 = torch.rand(coords.shape[0], 64, generator=g).to(device).bfloat16()
    res = {}
    for thr in (1, 10 ** 9):          # row-wise + pass over the output | tile plan + epilogue statistics
        if host_layer == "native":
            H.native_module().set_policy("rowwise_min_rows_with_stats", thr)
        else:
            MEB._ROWWISE_MIN_ROWS_WITH_STATS = thr
        try:
            torch.manual_seed(3)
            conv = ME.MinkowskiConvolution(64, 128, kernel_size=1, dimension=3).to(device)
            bn = ME.MinkowskiBatchNorm(128).to(device)
            f = f0.clone().requires_grad_(True)
            y = bn(conv(ME.SparseTensor(f, coords)))
            y.F.float().square().mean().backward()
            res[thr] = (y.F.detach().float(), f.grad.float(), bn.bn.running_mean.clone(), bn.bn.running_var.clone())
        finally:
            if host_layer == "native":
                H.native_module().set_policy("rowwise_min_rows_with_stats", 100000)
            else:
                MEB._ROWWISE_MIN_ROWS_WITH_STATS = 100000
    a, b = res[1], res[10 ** 9]
    scale = float(b[0].abs().max())
    assert float((a[0] - b[0]).abs().max()) <= 2.0 ** -7 * scale        # one bf16 ulp of the largest output
    assert float((a[1] - b[1]).abs().max()) <= 0.02 * float(b[1].abs().max()) + 1e-6
    assert float((a[2] - b[2]).abs().max()) <= 1e-5 * scale + 1e-6
    assert float(((a[3] - b[3]) / b[3]).abs().max()) <= 1e-4
