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
def test_batch_norm_matches_torch_float64(device, n, c, offset):
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


def test_batch_norm_bf16_rows(device):
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


def test_batch_norm_reproducible_and_eval(device):
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
def test_fused_batch_norm_relu(device, dtype):
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
