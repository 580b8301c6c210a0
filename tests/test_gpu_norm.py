"""MinkowskiBatchNorm on the HIP kernels (csrc/norm.hip) against torch.nn.BatchNorm1d evaluated in float64 on
the CPU — the reference's MinkowskiBatchNorm IS BatchNorm1d on the feature matrix
(MinkowskiNormalization.py:35-82).  Tolerance 1e-5 (abs + rel) in fp32; for bf16 rows the output is compared
after one bf16 rounding (2^-8 relative + 1e-3 of the range)."""
import numpy as np
import pytest
import torch

from helpers import make_cloud

pytestmark = pytest.mark.gpu


def _reference(x64, w, b, gy64, eps=1e-5, momentum=0.1):
    bn = torch.nn.BatchNorm1d(x64.shape[1], eps=eps, momentum=momentum).double()
    with torch.no_grad():
        bn.weight.copy_(w.double())
        bn.bias.copy_(b.double())
    x = x64.clone().requires_grad_(True)
    y = bn(x)
    y.backward(gy64)
    return y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var


def _ours(device, x, w, b, gy, coords):
    import minkowskiengine_amd as ME
    bn = ME.MinkowskiBatchNorm(x.shape[1]).to(device)
    with torch.no_grad():
        bn.bn.weight.copy_(w)
        bn.bn.bias.copy_(b)
    f = x.to(device).requires_grad_(True)
    st = ME.SparseTensor(f, coords.to(device))
    y = bn(st)
    y.F.backward(gy.to(device))
    return y.F.detach(), f.grad, bn.bn.weight.grad, bn.bn.bias.grad, bn.bn.running_mean, bn.bn.running_var, bn


def close(a, b, tol=1e-5):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()) <= tol * (1.0 + float(b.abs().max()))


@pytest.mark.parametrize("n,c,offset", [(5000, 32, 0.0), (1000, 96, 0.0), (3000, 7, 0.0), (2, 4, 0.0),
                                        (4000, 256, 0.0), (3000, 384, 0.0), (20000, 64, 300.0)])
def test_batch_norm_matches_torch_float64(device, host_layer, n, c, offset):
    g = torch.Generator().manual_seed(n + c)
    coords = make_cloud(n, 40, 3, seed=c)
    x = torch.randn(n, c, generator=g) * (1.0 + torch.arange(c) % 5) + offset   # offset: mean >> std
    w, b = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5
    gy = torch.randn(n, c, generator=g)
    ry, rdx, rdw, rdb, rrm, rrv = _reference(x.double(), w, b, gy.double())
    y, dx, dw, db, rm, rv, _ = _ours(device, x, w, b, gy, coords)
    tol = 1e-5 if offset == 0.0 else 2e-4      # fp32 input with mean 300, std ~1: 1e-7 * 300 relative to std
    assert close(y, ry, tol) and close(dx, rdx, tol) and close(dw, rdw, tol * 10) and close(db, rdb, tol * 10)
    assert close(rm, rrm, 1e-5) and close(rv, rrv, 1e-4)


def test_batch_norm_bf16_rows(device, host_layer):
    n, c = 50000, 64
    g = torch.Generator().manual_seed(1)
    coords = make_cloud(n, 60, 3, seed=1)
    x = (torch.randn(n, c, generator=g) * 2 + 1).bfloat16()
    w, b = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5
    gy = torch.randn(n, c, generator=g).bfloat16()
    ry, rdx, rdw, rdb, _, _ = _reference(x.double(), w, b, gy.double())
    y, dx, dw, db, _, _, _ = _ours(device, x, w, b, gy, coords)
    assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16 and dw.dtype == torch.float32
    for got, ref in ((y, ry), (dx, rdx)):
        err = (got.double().cpu() - ref).abs()
        assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-3 * ref.abs().max()).all())
    assert close(dw, rdw, 1e-4) and close(db, rdb, 1e-4)


def test_batch_norm_reproducible_and_eval(device, host_layer):
    import minkowskiengine_amd as ME
    n, c = 30000, 32
    g = torch.Generator().manual_seed(2)
    coords = make_cloud(n, 50, 3, seed=2)
    x, gy = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
    w, b = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g)
    a = _ours(device, x, w, b, gy, coords)
    b2 = _ours(device, x, w, b, gy, coords)
    assert all(torch.equal(p, q) for p, q in zip(a[:6], b2[:6]))
    bn = a[6].eval()
    st = ME.SparseTensor(x.to(device), coords.to(device))
    ref = torch.nn.functional.batch_norm(x.double(), bn.bn.running_mean.double().cpu(), bn.bn.running_var.double().cpu(),
                                         w.double(), b.double(), False, 0.1, 1e-5)
    assert close(bn(st).F, ref, 1e-5)
    assert int(bn.bn.num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_batch_norm_relu(device, host_layer, dtype):
    """MinkowskiBatchNorm(fuse_relu) + MinkowskiReLU = the two separate layers (forward clamp, backward mask
    recomputed from x), against relu(BatchNorm1d(x)) in float64."""
    import minkowskiengine_amd as ME
    n, c = 20000, 48
    g = torch.Generator().manual_seed(5)
    coords = make_cloud(n, 50, 3, seed=5)
    x = torch.randn(n, c, generator=g).to(dtype)
    w, b = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5
    gy = torch.randn(n, c, generator=g).to(dtype)
    ref_bn = torch.nn.BatchNorm1d(c).double()
    with torch.no_grad():
        ref_bn.weight.copy_(w.double())
        ref_bn.bias.copy_(b.double())
    xr = x.double().clone().requires_grad_(True)
    yr = torch.relu(ref_bn(xr))
    yr.backward(gy.double())
    bn, relu = ME.MinkowskiBatchNorm(c).to(device), ME.MinkowskiReLU()
    bn.fuse_relu = True
    with torch.no_grad():
        bn.bn.weight.copy_(w)
        bn.bn.bias.copy_(b)
    f = x.to(device).requires_grad_(True)
    mid = bn(ME.SparseTensor(f, coords.to(device)))
    y = relu(mid)
    assert y is mid and float(y.F.detach().min()) >= 0.0        # the ReLU module passed the rectified tensor through
    y.F.backward(gy.to(device))
    if dtype == torch.float32:
        assert close(y.F.detach(), yr.detach(), 1e-5) and close(f.grad, xr.grad, 1e-5)
        assert close(bn.bn.weight.grad, ref_bn.weight.grad, 1e-4) and close(bn.bn.bias.grad, ref_bn.bias.grad, 1e-4)
    else:
        for got, ref in ((y.F.detach(), yr.detach()), (f.grad, xr.grad)):
            err = (got.double().cpu() - ref).abs()
            # (an element within rounding of the ReLU threshold may flip its mask: allow a handful)
            assert int((err > 2.0 ** -7 * ref.abs() + 2e-3 * ref.abs().max()).sum()) <= 5
        assert close(bn.bn.weight.grad, ref_bn.weight.grad, 2e-3) and close(bn.bn.bias.grad, ref_bn.bias.grad, 2e-3)
    # evaluation mode: not fused, the ReLU module does its work
    bn.eval()
    out = relu(bn(ME.SparseTensor(x.to(device), coords.to(device))))
    assert float(out.F.min()) >= 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,c", [(5000, 32), (3001, 96), (700, 256), (2500, 20)])
def test_fused_norm_add_relu_is_bit_identical_to_the_three_operators(device, monkeypatch, dtype, n, c):
    """relu(bn(x) + skip) — the tail of a ResNet block — runs as ONE kernel per direction
    (MinkowskiBatchNorm.forward_residual, csrc/norm.hip k_bn_apply / k_bn_bwd_* with the residual arguments).  Output,
    both input gradients, the parameter gradients and the running statistics must equal those of batch norm, addition
    and ReLU run as separate operators, bit for bit (same arithmetic, same roundings, same summation order)."""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import layers as L
    g = torch.Generator().manual_seed(n + c)
    coords = make_cloud(n, 40, 3, seed=c).to(device)
    x0 = (torch.randn(n, c, generator=g) * 2).to(dtype)
    s0 = torch.randn(n, c, generator=g).to(dtype)
    gy = torch.randn(n, c, generator=g).to(dtype).to(device)
    w, b = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(L, "_FUSE_RESIDUAL", fused)
        bn = ME.MinkowskiBatchNorm(c).to(device)
        with torch.no_grad():
            bn.bn.weight.copy_(w)
            bn.bn.bias.copy_(b)
        x = x0.to(device).requires_grad_(True)
        s = s0.to(device).requires_grad_(True)
        y = bn.forward_residual(ME.SparseTensor(x, coords), ME.SparseTensor(s, coords), relu=True)
        y.F.backward(gy)
        res[fused] = (y.F.detach(), x.grad, s.grad, bn.bn.weight.grad, bn.bn.bias.grad, bn.bn.running_mean.clone(),
                      bn.bn.running_var.clone())
    assert float((res[True][0] == 0).float().mean()) > 0.2          # the ReLU does mask something
    for a, b_ in zip(res[True], res[False]):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("n,extent,D,cin,cout,ks", [(6000, 40, 3, 96, 96, 3), (4000, 14, 3, 64, 128, 3), (300, 6, 3, 256, 256, 3),
                                                    (3000, 9, 4, 32, 64, 3), (5000, 16, 3, 128, 96, 2), (70, 3, 3, 64, 64, 3),
                                                    (5000, 30, 3, 32, 32, 3), (2500, 12, 3, 256, 128, 3), (20000, 60, 3, 16, 32, 3)])
def test_convolution_epilogue_statistics_equal_a_pass_over_the_output(device, n, extent, D, cin, cout, ks):
    """me_conv_target_bf16_stats: the bf16 forward launch leaves the (mean, M2) of every tile and
    output channel behind, and me_bn_stats_from_tiles merges them into the batch norm's mean / rstd / running statistics
    without reading the matrix.  The convolution's output is bit-identical with and without the epilogue; the statistics
    equal those of k_bn_partial's pass over that output up to the regrouping of fp32 partial sums (1e-5 relative + 1e-6
    on the mean, 1e-4 on rstd), for every slab width, the fused and the deep instantiation, one-tile and many-tile maps."""
    from minkowskiengine_amd import backend as MEB
    coords = make_cloud(n, extent, D, seed=cin + cout + n, batch=2, negative=True)
    mgr = MEB.CoordinateMapManagerGPU_c10()
    key, _ = mgr.insert_and_map(coords.to(device), [1] * D, "")
    km = mgr._kernel_map(key, key, [ks] * D, [1] * D, [1] * D, MEB.RegionType.HYPER_CUBE, None, False, False)
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(coords.shape[0], cin, generator=g) - 0.3).to(device).bfloat16()
    w = (torch.rand(ks ** D, cin, cout, generator=g) - 0.5).to(device)
    MEB._BN_PARTIALS.clear()
    try:
        MEB.conv_bn_stats_hint(False)
        y0 = MEB._conv_forward(x, w, km, "mfma")
        assert not MEB._BN_PARTIALS
        MEB.conv_bn_stats_hint(True)
        y1 = MEB._conv_forward(x, w, km, "mfma")
    finally:
        MEB.conv_bn_stats_hint(False)
    assert torch.equal(y0, y1)
    assert y1.data_ptr() in MEB._BN_PARTIALS
    rm0, rv0 = torch.zeros(cout, device=device), torch.ones(cout, device=device)
    rm1, rv1 = rm0.clone(), rv0.clone()
    nb0, nb1 = torch.zeros((), dtype=torch.int64, device=device), torch.zeros((), dtype=torch.int64, device=device)
    mean1, rstd1 = MEB.bn_stats(y1, 1e-5, 0.1, rm1, rv1, nb1)          # from the tiles (the entry is consumed)
    assert not MEB._BN_PARTIALS
    mean0, rstd0 = MEB.bn_stats(y0, 1e-5, 0.1, rm0, rv0, nb0)          # a pass over the matrix
    ref = y0.double()
    assert close(mean0, ref.mean(0), 1e-5) and close(mean1, ref.mean(0), 1e-5)
    scale = float(ref.abs().max())
    assert float((mean1 - mean0).abs().max()) <= 1e-5 * scale + 1e-6
    assert float(((rstd1 - rstd0) / rstd0).abs().max()) <= 1e-4
    assert float((rm1 - rm0).abs().max()) <= 1e-5 * scale + 1e-6 and float(((rv1 - rv0) / rv0).abs().max()) <= 1e-4
    assert int(nb1) == 1 and int(nb0) == 1
    # the statistics belong to that tensor at that version: a modified, sliced or different tensor reads the matrix
    MEB.conv_bn_stats_hint(True)
    try:
        y2 = MEB._conv_forward(x, w, km, "mfma")
        y2.add_(1.0)
        assert MEB._bn_partials_take(y2) is None
        y3 = MEB._conv_forward(x, w, km, "mfma")
        assert MEB._bn_partials_take(y3[:, : cout // 2].contiguous()) is None
        assert MEB._bn_partials_take(y3) is not None and MEB._bn_partials_take(y3) is None
        y4 = MEB._conv_forward(x, w, km, "mfma")
        assert len(MEB._BN_PARTIALS) == 1
        del y4                                    # never normalised: its entry goes with the tensor
        assert not MEB._BN_PARTIALS
    finally:
        MEB.conv_bn_stats_hint(False)
        MEB._BN_PARTIALS.clear()


@pytest.mark.parametrize("native", [False, True])
def test_conv_batchnorm_pair_uses_the_epilogue_statistics(device, native, monkeypatch):
    """MinkowskiConvolution -> MinkowskiBatchNorm in training mode, bf16, on both host layers: with the epilogue
    statistics (default) the pair runs without k_bn_partial and agrees with the two-pass pair to bf16 rounding of the
    normalised output; the gradient flows as before."""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import backend as MEB, host as H
    if native and H.native_module() is None:
        pytest.skip("native host layer not built")
    prev = H.get_host()
    H.set_host("native" if native else "python")
    try:
        MEB._BN_PARTIALS.clear()
        coords = make_cloud(8000, 30, 3, seed=5, batch=2).to(device)
        g = torch.Generator().manual_seed(9)
        f0 = torch.rand(coords.shape[0], 64, generator=g).to(device).bfloat16()
        res = {}
        for fused in (True, False):
            monkeypatch.setattr(MEB, "_CONV_BN_STATS", fused)
            if native:
                H.native_module().set_conv_bn_stats(1 if fused else 0)
            torch.manual_seed(3)
            conv = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).to(device)
            bn = ME.MinkowskiBatchNorm(128).to(device)
            f = f0.clone().requires_grad_(True)
            y = bn(conv(ME.SparseTensor(f, coords)))
            y.F.float().square().mean().backward()
            res[fused] = (y.F.detach().float(), f.grad.float(), bn.bn.running_mean.clone(), bn.bn.running_var.clone())
            if fused and not native:
                assert not MEB._BN_PARTIALS, "the batch norm consumed the convolution's statistics"
        a, b = res[True], res[False]
        scale = float(b[0].abs().max())
        assert float((a[0] - b[0]).abs().max()) <= 2.0 ** -7 * scale        # one bf16 ulp of the largest output
        assert float((a[1] - b[1]).abs().max()) <= 0.02 * float(b[1].abs().max()) + 1e-6
        assert float((a[2] - b[2]).abs().max()) <= 1e-5 * scale + 1e-6
        assert float(((a[3] - b[3]) / b[3]).abs().max()) <= 1e-4
    finally:
        if native:
            H.native_module().set_conv_bn_stats(-1)
        H.set_host(prev)


def test_batch_norm_buffers_of_another_dtype_are_not_overrun(device, host_layer):
    """after `model.bfloat16()` the running statistics are bf16 — c floats written into a 2c-byte buffer
    would corrupt its neighbour.  They go through float32 temporaries; the guard rows stay intact and the statistics
    equal the float32 module's (rounded to bf16)."""
    import minkowskiengine_amd as ME
    from helpers import make_cloud
    c = 32
    coords = make_cloud(3000, 14, 3, seed=2).to(device)
    f = torch.rand(coords.shape[0], c, generator=torch.Generator().manual_seed(4)).to(device)
    ref = ME.MinkowskiBatchNorm(c).to(device)
    ref.train()
    x32 = ME.SparseTensor(f, coords)
    y32 = ref(x32)
    bn = ME.MinkowskiBatchNorm(c).to(device).bfloat16()
    assert bn.bn.running_mean.dtype == torch.bfloat16
    # the two running buffers carved out of ONE allocation with guard values around them
    slab = torch.full((4 * c,), 7.0, dtype=torch.bfloat16, device=device)
    slab[c:2 * c] = 0.0
    slab[2 * c:3 * c] = 1.0
    bn.bn.running_mean = slab[c:2 * c]
    bn.bn.running_var = slab[2 * c:3 * c]
    bn.train()
    xb = ME.SparseTensor(f.bfloat16(), coordinate_map_key=x32.coordinate_map_key, coordinate_manager=x32.coordinate_manager)
    yb = bn(xb)
    torch.cuda.synchronize()
    assert bool((slab[:c] == 7.0).all()) and bool((slab[3 * c:] == 7.0).all()), "wrote past the running statistics"
    assert torch.allclose(bn.bn.running_mean.float(), ref.bn.running_mean, atol=1e-2)
    assert torch.allclose(bn.bn.running_var.float(), ref.bn.running_var, atol=1e-2)
    assert int(bn.bn.num_batches_tracked) == 1
    assert torch.allclose(yb.F.float(), y32.F, atol=0.06)
    bn.eval()
    ye = bn(xb)                                                # evaluation with bf16 buffers: float32 vectors inside
    assert ye.F.dtype == torch.bfloat16 and bool(torch.isfinite(ye.F.float()).all())
