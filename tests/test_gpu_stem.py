"""GPU parity of the stacked-offset bf16 convolution for layers with at most 8 source channels (csrc/conv_stem.hip, round 5):
the MinkUNet stem (3 -> 32 channels, 5^3 offsets) and its relatives.  Same oracle and tolerance as tests/test_gpu_bf16.py
(the fp32 reference algorithm, oracle/me_oracle.py, on the bf16-rounded operands; |err| <= 2^-8 |ref| + 1e-3 max|ref|).
The kernel is the DEFAULT for these shapes (me_conv_stem_use_bf16); every case also runs with ME's tile-plan kernels
(me_debug_set_stem(0)) and the two schedules must agree to bf16 rounding."""
import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from helpers import assert_close, make_cloud
from test_gpu_bf16 import _run_layer, assert_bf16_close, bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture
def stem():
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    try:
        yield lambda mode, groups=0: lib.me_debug_set_stem(mode, groups)
    finally:
        lib.me_debug_set_stem(-1, 0)


STEM_CASES = [
    # n, extent, cin, cout, ks, stride, dil
    (6000, 20, 3, 32, 5, 1, 1),      # the MinkUNet stem: 125 offsets = 32 quads (three padded slots in the last)
    (3000, 14, 8, 64, 3, 1, 1),      # 8 real channels, four column blocks, 27 offsets = 7 quads
    (3000, 14, 4, 16, 3, 1, 1),      # one column block
    (3000, 30, 1, 32, 3, 1, 2),      # one channel, dilated, sparse map
    (3000, 14, 3, 32, 2, 2, 1),      # strided: the output map differs from the input map, 8 offsets = 2 quads
    (3000, 14, 32, 8, 3, 1, 1),      # 8 OUTPUT channels: the input gradient (8 -> 32, transposed weights) takes the kernel
    (100, 5, 3, 32, 3, 1, 1),        # a single partial tile
    (129, 5, 6, 64, 3, 1, 1),        # one row beyond a tile
]


@pytest.mark.parametrize("groups", [0, 1, 2], ids=["g-policy", "g1", "g2"])
@pytest.mark.parametrize("n,extent,cin,cout,ks,stride,dil", STEM_CASES)
def test_stem_conv_forward_backward_vs_oracle(device, host_layer, stem, n, extent, cin, cout, ks, stride, dil, groups):
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    stem(-1, groups)                  # rows per wave: 16 x groups (tiles of 64 / 128 / 256 rows)
    assert lib.me_conv_stem_use_bf16(n, ks ** 3, 8, cout if cin <= 8 else cin) == 1      # (channels as the host pads them)
    coords = make_cloud(n, extent, 3, seed=n + cin, batch=2 if n > 200 else 1, negative=True)
    conv, x, y, feats, gy = _run_layer(device, coords, cin, cout, ks, stride, dil)
    in_c, out_c = coords.numpy(), y.C.cpu().numpy()
    _, km = O.kernel_map(in_c, out_c, O.make_region(3, ks, dil, 1))       # (region on the INPUT tensor stride, 1)
    w = conv.kernel.detach().float().cpu().numpy()
    assert_bf16_close(y.F.detach().float().cpu().numpy(), O.conv_forward(feats.numpy(), w, km, len(out_c)), "forward")
    gi, gw = O.conv_backward(feats.numpy(), gy.numpy(), w, km)
    assert_bf16_close(x.F.grad.float().cpu().numpy(), gi, "grad_in")
    assert_close(conv.kernel.grad.cpu().numpy(), gw)
    # the tile-plan kernels on the same layer: another summation order, the same sums
    stem(0)
    _, x2, y2, _, _ = _run_layer(device, coords, cin, cout, ks, stride, dil)
    for what, a, b in (("forward", y.F, y2.F), ("grad_in", x.F.grad, x2.F.grad)):
        a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
        tol = 2.0 ** -7 * np.abs(b) + 2e-3 * max(1.0, np.abs(b).max())
        assert not (np.abs(a - b) > tol).any(), (what, float(np.abs(a - b).max()))


def test_stem_conv_transposed_layer_and_bf16_kernel(device, host_layer):
    """a generative / transposed layer with 3 source channels (the kernel map's other side) and a kernel STORED in bf16"""
    import minkowskiengine_amd as ME
    coords = make_cloud(2500, 12, 3, seed=3)
    g = torch.Generator().manual_seed(1)
    feats = bf16_round(torch.rand(coords.shape[0], 3, generator=g) - 0.3)
    conv = ME.MinkowskiConvolutionTranspose(3, 32, kernel_size=3, stride=1, dimension=3)
    with torch.no_grad():
        conv.kernel.copy_(bf16_round(torch.rand(conv.kernel.shape, generator=g) - 0.5))
    conv = conv.to(device).to(torch.bfloat16)
    x = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device))
    y = conv(x)
    w = conv.kernel.detach().float().cpu().numpy()
    # the transposed layer's map = the forward map of (output -> input) with the sides swapped
    # (src/coordinate_map_manager.cpp:560-626)
    _, fkm = O.kernel_map(y.C.cpu().numpy(), coords.numpy(), O.make_region(3, 3, 1, 1))
    km = {k: np.stack((v[1], v[0])) for k, v in fkm.items()}
    assert_bf16_close(y.F.detach().float().cpu().numpy(), O.conv_forward(feats.numpy(), w, km, y.F.shape[0]), "forward")


def test_stem_conv_is_bitwise_reproducible_and_host_independent(device):
    import minkowskiengine_amd as ME
    coords = make_cloud(5000, 16, 3, seed=9)
    prev = ME.get_host()
    res = {}
    try:
        for h in ("python", "native"):
            ME.set_host(h)
            r1 = _run_layer(device, coords, 3, 32, 5)
            r2 = _run_layer(device, coords, 3, 32, 5)
            assert torch.equal(r1[2].F, r2[2].F)
            res[h] = r1
    finally:
        ME.set_host(prev)
    assert torch.equal(res["python"][2].F, res["native"][2].F)


@pytest.mark.parametrize("groups", [1, 2, 4])
def test_stem_conv_batch_norm_partials(device, stem, groups):
    """the tile statistics the kernel leaves behind give the batch norm the mean / variance of the stored matrix"""
    import minkowskiengine_amd as ME
    stem(-1, groups)
    coords = make_cloud(5000, 20, 3, seed=4)
    g = torch.Generator().manual_seed(0)
    feats = bf16_round(torch.rand(coords.shape[0], 3, generator=g) - 0.4)
    for host in ("python", "native"):
        prev = ME.get_host()
        ME.set_host(host)
        try:
            conv = ME.MinkowskiConvolution(3, 32, kernel_size=5, dimension=3).to(device)
            bn = ME.MinkowskiBatchNorm(32).to(device)
            conv.train(), bn.train()
            x = ME.SparseTensor(feats.to(device).to(torch.bfloat16), coords.to(device))
            y = conv(x)
            z = bn(y)
            yf = y.F.float()
            mean, var = yf.mean(0), yf.var(0, unbiased=False)
            ref = (yf - mean) / torch.sqrt(var + bn.bn.eps) * bn.bn.weight.float() + bn.bn.bias.float()
            err = (z.F.float() - ref).abs().max().item()
            assert err <= 2.0 ** -7 * ref.abs().max().item() + 1e-2, (host, err)
            assert torch.allclose(bn.bn.running_mean.float(), 0.1 * mean, atol=1e-3, rtol=1e-2), host
        finally:
            ME.set_host(prev)


def test_stem_policy(device, stem):
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    assert lib.me_conv_stem_use_bf16(200000, 125, 8, 32) == 1
    assert lib.me_conv_stem_use_bf16(200000, 27, 8, 64) == 1
    assert lib.me_conv_stem_use_bf16(200000, 1, 8, 32) == 0          # a 1 x 1 layer: nothing to stack
    assert lib.me_conv_stem_use_bf16(200000, 125, 16, 32) == 0        # more than 8 source channels
    assert lib.me_conv_stem_use_bf16(200000, 125, 8, 24) == 0         # column count not 16 / 32 / 64
    assert lib.me_conv_stem_use_bf16(200000, 125, 8, 128) == 0
    stem(0)
    assert lib.me_conv_stem_use_bf16(200000, 125, 8, 32) == 0
    stem(1)
    assert lib.me_conv_stem_use_bf16(200000, 1, 8, 32) == 1


def test_stem_conv_without_source_rows(device):
    """a source side without rows (every neighbour absent): zeros and statistics of zeros, no launch on a null matrix"""
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    n_tgt, K, cout = 300, 27, 32
    tbl = torch.full((K, n_tgt), -1, dtype=torch.int32, device=device)
    w = torch.rand(K, 8, cout, device=device)
    out = torch.full((n_tgt, cout), 7.0, dtype=torch.bfloat16, device=device)
    tiles = -(-n_tgt // int(lib.me_conv_stem_tile_rows()))
    part = torch.full((2, tiles, cout), 3.0, dtype=torch.float32, device=device)
    _lib.check(lib.me_conv_stem_bf16(None, 0, 8, w.data_ptr(), 1, 0, K, cout, tbl.data_ptr(), None, None, out.data_ptr(), n_tgt,
                                     part[0].data_ptr(), part[1].data_ptr(), None))
    torch.cuda.synchronize()
    assert float(out.float().abs().max()) == 0.0 and float(part.abs().max()) == 0.0
