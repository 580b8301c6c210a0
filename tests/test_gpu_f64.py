"""float64 features on the GPU (round 4): the reference dispatches every feature operator for float AND double
(AT_DISPATCH_FLOATING_TYPES, src/convolution_gpu.cu:137-155) and its own tests are float64 `gradcheck`s of the autograd
Functions (tests/python/convolution.py:148-168, 201-218, pool.py, global.py, broadcast.py:79-160 through
MinkowskiEngine/utils/gradcheck.py:34-57).  The same checks here, on both host layers: torch.autograd.gradcheck of the
Functions (eps 1e-6, atol 1e-5, rtol 1e-3 — the reference's own settings), and values against the float64 oracle to
1e-12 (the kernels of csrc/f64.hip are plain double FMAs in the oracle's loop order)."""
import numpy as np
import pytest
import torch
from torch.autograd import gradcheck

from oracle import me_oracle as O
from helpers import make_cloud

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("host_layer")]

GC = dict(eps=1e-6, atol=1e-5, rtol=1e-3)     # MinkowskiEngine/utils/gradcheck.py:37-39


def _data_loader(device, nchannel=2, seed=0):
    """the reference's tiny fixture (tests/python/common.py:57-78): two batches of a few 2-D points"""
    import minkowskiengine_amd as ME
    c0 = [[0, 0], [0, 1], [1, 0], [1, 1], [2, 1], [3, 2], [0, 3]]
    c1 = [[1, 0], [0, 2], [2, 2], [3, 0], [1, 3]]
    coords = ME.utils.batched_coordinates([torch.IntTensor(c0), torch.IntTensor(c1)])
    g = torch.Generator().manual_seed(seed)
    feats = torch.rand(coords.shape[0], nchannel, generator=g, dtype=torch.float64)
    return coords.to(device), feats.to(device)


@pytest.mark.parametrize("ks,stride,transposed", [(3, 1, False), (2, 2, False), (3, 2, False), (2, 2, True)])
def test_convolution_gradcheck(device, ks, stride, transposed):
    """tests/python/convolution.py:148-168 (stride, kernel) and convolution_transpose: gradcheck of the autograd
    Function in float64"""
    import minkowskiengine_amd as ME
    coords, feats = _data_loader(device)
    feats.requires_grad_()
    x = ME.SparseTensor(feats, coords)
    if transposed:
        down = ME.MinkowskiConvolution(2, 2, kernel_size=2, stride=2, dimension=2).double().to(device)
        x = down(x)
        xf = x.F.detach().clone().requires_grad_()
        x = ME.SparseTensor(xf, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
        conv = ME.MinkowskiConvolutionTranspose(2, 3, kernel_size=ks, stride=stride, dimension=2).double().to(device)
        fn = ME.MinkowskiConvolutionTransposeFunction
    else:
        conv = ME.MinkowskiConvolution(2, 3, kernel_size=ks, stride=stride, dimension=2).double().to(device)
        fn = ME.MinkowskiConvolutionFunction
    y = conv(x)
    assert y.F.dtype == torch.float64
    assert gradcheck(lambda f, w: fn.apply(f, w, conv.kernel_generator, conv.convolution_mode, x.coordinate_map_key,
                                           y.coordinate_map_key, x.coordinate_manager), (x.F, conv.kernel), **GC)
    # the module path (on the native host: its C++ autograd function)
    assert gradcheck(lambda f: conv(ME.SparseTensor(f, coordinate_map_key=x.coordinate_map_key,
                                                    coordinate_manager=x.coordinate_manager)).F, (x.F,), **GC)


@pytest.mark.parametrize("n,extent,D,cin,cout,ks,stride", [(3000, 14, 3, 16, 32, 3, 1), (2500, 14, 3, 5, 7, 3, 1),
                                                            (2500, 14, 3, 8, 8, 2, 2), (1500, 8, 4, 6, 10, 3, 1)])
def test_convolution_float64_matches_the_oracle(device, n, extent, D, cin, cout, ks, stride):
    import minkowskiengine_amd as ME
    coords = make_cloud(n, extent, D, seed=n + cin, batch=2, negative=True)
    g = torch.Generator().manual_seed(1)
    feats = torch.rand(coords.shape[0], cin, generator=g, dtype=torch.float64)
    conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, dimension=D).double()
    with torch.no_grad():
        conv.kernel.copy_(torch.rand(conv.kernel.shape, generator=g, dtype=torch.float64) - 0.5)
    conv = conv.to(device)
    x = ME.SparseTensor(feats.to(device), coords.to(device), requires_grad=True)
    y = conv(x)
    gy = torch.rand(y.F.shape, generator=g, dtype=torch.float64)
    y.F.backward(gy.to(device))
    in_c, out_c = coords.numpy(), y.C.cpu().numpy()
    _, km = O.kernel_map(in_c, out_c, O.make_region(D, ks, 1, 1))
    w = conv.kernel.detach().cpu().numpy()
    ref = O.conv_forward(feats.numpy(), w, km, len(out_c))
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
    for got, want, what in ((y.F, ref, "forward"), (x.F.grad, gi, "grad_in"), (conv.kernel.grad, gw, "grad_kernel")):
        got = got.detach().cpu().numpy()
        assert got.dtype == np.float64
        err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        assert err < 1e-12, f"{what}: {err}"


@pytest.mark.parametrize("mode", ["sum", "avg", "max"])
def test_local_pooling_gradcheck(device, mode):
    """tests/python/pool.py: MinkowskiLocalPoolingFunction in float64"""
    import minkowskiengine_amd as ME
    coords, feats = _data_loader(device)
    feats.requires_grad_()
    x = ME.SparseTensor(feats, coords)
    pool = {"sum": ME.MinkowskiSumPooling, "avg": ME.MinkowskiAvgPooling, "max": ME.MinkowskiMaxPooling}[mode](
        kernel_size=2, stride=2, dimension=2)
    y = pool(x)
    assert y.F.dtype == torch.float64
    assert gradcheck(lambda f: pool(ME.SparseTensor(f, coordinate_map_key=x.coordinate_map_key,
                                                    coordinate_manager=x.coordinate_manager)).F, (x.F,), **GC)
    up = ME.MinkowskiPoolingTranspose(kernel_size=2, stride=2, dimension=2)
    yf = y.F.detach().clone().requires_grad_()
    assert gradcheck(lambda f: up(ME.SparseTensor(f, coordinate_map_key=y.coordinate_map_key,
                                                  coordinate_manager=y.coordinate_manager)).F, (yf,), **GC)


@pytest.mark.parametrize("mode", ["sum", "avg", "max"])
def test_global_pooling_and_broadcast_gradcheck(device, mode):
    """tests/python/global.py and broadcast.py:79-160 in float64"""
    import minkowskiengine_amd as ME
    coords, feats = _data_loader(device, nchannel=3)
    feats.requires_grad_()
    x = ME.SparseTensor(feats, coords)
    gp = {"sum": ME.MinkowskiGlobalSumPooling, "avg": ME.MinkowskiGlobalAvgPooling,
          "max": ME.MinkowskiGlobalMaxPooling}[mode]()
    mk = lambda f: ME.SparseTensor(f, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
    y = gp(x)
    assert y.F.dtype == torch.float64 and y.F.shape == (2, 3)
    # values: per batch index
    b = coords[:, 0].long()
    for i in range(2):
        rows = feats.detach()[b == i]
        want = {"sum": rows.sum(0), "avg": rows.mean(0), "max": rows.max(0).values}[mode]
        assert torch.allclose(y.F[i], want, rtol=0, atol=1e-13)
    assert gradcheck(lambda f: gp(mk(f)).F, (x.F,), **GC)
    glob = y.F.detach().clone().requires_grad_()
    for bc in (ME.MinkowskiBroadcastAddition(), ME.MinkowskiBroadcastMultiplication()):
        assert gradcheck(lambda f, gl: bc(mk(f), ME.SparseTensor(gl, coordinate_map_key=y.coordinate_map_key,
                                                                 coordinate_manager=y.coordinate_manager)).F,
                         (x.F, glob), **GC)


def test_float64_network_arbitrates_float32(device):
    """What the float64 path is for besides gradcheck: a small conv -> batch norm -> pooling -> conv stack in float64
    and in float32 on the same weights — the fp32 kernels stay within 1e-4 of the float64 result."""
    import minkowskiengine_amd as ME
    coords = make_cloud(4000, 16, 3, seed=3, batch=2).to(device)
    g = torch.Generator().manual_seed(2)
    feats = torch.rand(coords.shape[0], 8, generator=g, dtype=torch.float64).to(device)

    def net(dtype):
        torch.manual_seed(0)
        c1 = ME.MinkowskiConvolution(8, 16, kernel_size=3, dimension=3)
        c2 = ME.MinkowskiConvolution(16, 16, kernel_size=2, stride=2, dimension=3)
        c3 = ME.MinkowskiConvolutionTranspose(16, 8, kernel_size=2, stride=2, dimension=3)
        mods = torch.nn.ModuleList([c1, c2, c3]).to(device).to(dtype)
        x = ME.SparseTensor(feats.to(dtype), coords)
        y = c3(ME.MinkowskiReLU()(c2(ME.MinkowskiReLU()(c1(x)))))
        return y.F.double()
    a, b = net(torch.float64), net(torch.float32)
    assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(a.abs().max()))
