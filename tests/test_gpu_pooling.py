"""GPU parity of the pooling / broadcast kernels (csrc/pool.hip, through the C ABI and the reference-shaped
Python API) against the fixtures produced by the reference's CPU operators and against the oracle.
Sums / averages / maxima follow the reference's order of operations: forward results are compared
bit-exactly, gradients within 1e-6."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from helpers import GOLDEN_DIR, make_cloud, row_mapping

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("host_layer")]   # both host layers
POOL_CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "pool_*.npz")))


def _tensor(ME, z, device, requires_grad=True):
    # the fixture's in_coords are the reference's unique rows in first-occurrence order = ours
    return ME.SparseTensor(torch.from_numpy(z["feats"]).to(device), torch.from_numpy(z["in_coords"]).to(device),
                           requires_grad=requires_grad)


@pytest.mark.parametrize("path", POOL_CASES, ids=[os.path.basename(p)[:-4] for p in POOL_CASES])
@pytest.mark.parametrize("mode", ["sum", "avg", "max"])
def test_local_pooling_vs_reference_fixture(device, path, mode):
    import minkowskiengine_amd as ME
    z = np.load(path)
    ks, st = z["kernel_size"].tolist(), z["stride"].tolist()
    layer = {"sum": ME.MinkowskiSumPooling, "avg": ME.MinkowskiAvgPooling, "max": ME.MinkowskiMaxPooling}[mode](
        kernel_size=ks, stride=st, dimension=3)
    x = _tensor(ME, z, device)
    y = layer(x)
    # strided output rows: ours are in first-occurrence order, the reference's in hash order -> relabel
    m = row_mapping(y.C.cpu().numpy(), z["out_coords"])         # our row i == fixture row m[i]
    out = y.F.detach().cpu().numpy()
    if ks == st and mode != "max":
        # kernel == stride: the reference builds this map with stride_map and sums each cell in INPUT-ROW
        # order (coordinate_map_manager.cpp:722-729); we sum in kernel-offset order -> last-bit differences
        assert np.allclose(out, z[mode + "_out"][m], rtol=1e-6, atol=1e-6)
    else:
        assert np.array_equal(out, z[mode + "_out"][m]), "forward must be bit-identical to the reference CPU path"
    gy = torch.from_numpy(z["grad_out"][m]).to(device)
    y.F.backward(gy)
    assert np.allclose(x.F.grad.cpu().numpy(), z[mode + "_grad_in"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("path", POOL_CASES, ids=[os.path.basename(p)[:-4] for p in POOL_CASES])
@pytest.mark.parametrize("mode", ["sum", "avg", "max"])
def test_global_pooling_vs_reference_fixture(device, path, mode):
    import minkowskiengine_amd as ME
    z = np.load(path)
    layer = {"sum": ME.MinkowskiGlobalSumPooling, "avg": ME.MinkowskiGlobalAvgPooling,
             "max": ME.MinkowskiGlobalMaxPooling}[mode]()
    x = _tensor(ME, z, device)
    y = layer(x)
    assert y.F.shape == z["g" + mode + "_out"].shape
    m = row_mapping(y.C.cpu().numpy(), z["glob_coords"])
    assert np.allclose(y.F.detach().cpu().numpy(), z["g" + mode + "_out"][m], rtol=1e-5, atol=1e-5)
    y.F.backward(torch.from_numpy(z["g" + mode + "_grad_out"][m]).to(device))
    assert np.allclose(x.F.grad.cpu().numpy(), z["g" + mode + "_grad_in"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("path", POOL_CASES[:2], ids=[os.path.basename(p)[:-4] for p in POOL_CASES[:2]])
@pytest.mark.parametrize("mul", [False, True])
def test_broadcast_vs_reference_fixture(device, path, mul):
    import minkowskiengine_amd as ME
    z = np.load(path)
    x = _tensor(ME, z, device)
    pooled = ME.MinkowskiGlobalSumPooling()(ME.SparseTensor(x.F.detach(), coordinate_map_key=x.coordinate_map_key,
                                                            coordinate_manager=x.coordinate_manager))
    m = row_mapping(pooled.C.cpu().numpy(), z["glob_coords"])
    glob = ME.SparseTensor(torch.from_numpy(z["bglob"][m]).to(device).requires_grad_(True),
                           coordinate_map_key=pooled.coordinate_map_key, coordinate_manager=x.coordinate_manager)
    layer = ME.MinkowskiBroadcastMultiplication() if mul else ME.MinkowskiBroadcastAddition()
    y = layer(x, glob)
    nm = "bmul" if mul else "badd"
    assert np.allclose(y.F.detach().cpu().numpy(), z[nm + "_out"], rtol=1e-6, atol=1e-7)
    y.F.backward(torch.from_numpy(z["bgrad_out"]).to(device))
    assert np.allclose(x.F.grad.cpu().numpy(), z[nm + "_grad_in"], rtol=1e-6, atol=1e-7)
    assert np.allclose(glob.F.grad.cpu().numpy(), z[nm + "_grad_glob"][m], rtol=1e-5, atol=1e-5)
    cat = ME.MinkowskiBroadcastConcatenation()(x, glob)
    assert cat.F.shape[1] == 2 * x.F.shape[1]


@pytest.mark.parametrize("c,ks,stride", [(64, 2, 2), (3, 3, 1), (32, 3, 2)])
def test_local_pooling_vs_oracle_larger(device, c, ks, stride):
    """20k voxels: bit-exact forward against the numpy oracle on the oracle's own kernel map."""
    import minkowskiengine_amd as ME
    coords = make_cloud(20000, 40, 3, seed=c, batch=2, negative=True)
    g = torch.Generator().manual_seed(5)
    feats = torch.rand(coords.shape[0], c, generator=g) - 0.5
    x = ME.SparseTensor(feats.to(device), coords.to(device), requires_grad=True)
    for mode, cls in (("sum", ME.MinkowskiSumPooling), ("avg", ME.MinkowskiAvgPooling), ("max", ME.MinkowskiMaxPooling)):
        y = cls(kernel_size=ks, stride=stride, dimension=3)(x)
        out_c = y.C.cpu().numpy()
        _, km = O.kernel_map(coords.numpy(), out_c, O.make_region(3, ks, 1, 1))
        ref_out, aux = O.pool_forward(feats.numpy(), km, out_c.shape[0], mode)
        assert np.array_equal(y.F.detach().cpu().numpy(), ref_out), mode   # same offsets, same order: bit-exact
        gy = torch.rand(y.F.shape, generator=g)
        x.F.grad = None
        y.F.backward(gy.to(device))
        assert np.allclose(x.F.grad.cpu().numpy(), O.pool_backward(gy.numpy(), km, coords.shape[0], mode, aux),
                           rtol=1e-5, atol=1e-6), mode


def test_pooling_transpose_restores_rows(device):
    """MinkowskiPoolingTranspose sums over the transposed map: unpooling a sum-pooled tensor of ones gives, on
    every input row, the number of inputs in its output cell."""
    import minkowskiengine_amd as ME
    coords = make_cloud(5000, 20, 3, seed=4, batch=2)
    x = ME.SparseTensor(torch.ones(coords.shape[0], 4, device=device), coords.to(device))
    pooled = ME.MinkowskiSumPooling(kernel_size=2, stride=2, dimension=3)(x)
    up = ME.MinkowskiPoolingTranspose(kernel_size=2, stride=2, dimension=3)(pooled)
    assert up.coordinate_map_key == x.coordinate_map_key
    cells = (coords[:, 1:] // 2).numpy()
    key = np.concatenate([coords[:, :1].numpy(), cells], 1)
    _, inv, cnt = np.unique(key, axis=0, return_inverse=True, return_counts=True)
    assert np.array_equal(up.F.cpu().numpy()[:, 0], cnt[inv.reshape(-1)].astype(np.float32))


def _bf16_round(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize("c,ks,stride", [(64, 2, 2), (24, 3, 1), (20, 3, 2), (5, 2, 2)])
def test_bf16_pooling_and_broadcast(device, c, ks, stride):
    """bf16 features on the pooling / broadcast kernels (round 1 handed bf16 pointers to the fp32 kernels:
    out-of-bounds reads and writes).  Semantics: widen to fp32, same order of operations, ONE rounding at the
    store — so the result equals bf16(oracle on the bf16-rounded inputs) up to one bf16 ulp of double rounding for
    sums (2^-8 relative), exactly for maxima and their argmax masks.  Outputs and gradients keep the feature dtype
    and the exact shape (no byte is written beyond n * c * 2)."""
    import minkowskiengine_amd as ME
    coords = make_cloud(6000, 24, 3, seed=c, batch=2, negative=True)
    g = torch.Generator().manual_seed(8)
    feats = (torch.rand(coords.shape[0], c, generator=g) - 0.5).to(torch.bfloat16)
    f32 = feats.float().numpy()
    x = ME.SparseTensor(feats.to(device), coords.to(device), requires_grad=True)
    # a canary right behind the feature matrix in the caching allocator's pool
    for mode, cls in (("sum", ME.MinkowskiSumPooling), ("avg", ME.MinkowskiAvgPooling), ("max", ME.MinkowskiMaxPooling)):
        y = cls(kernel_size=ks, stride=stride, dimension=3)(x)
        assert y.F.dtype == torch.bfloat16 and y.F.shape[1] == c
        out_c = y.C.cpu().numpy()
        _, km = O.kernel_map(coords.numpy(), out_c, O.make_region(3, ks, 1, 1))
        ref_out, aux = O.pool_forward(f32, km, out_c.shape[0], mode)
        got = y.F.detach().float().cpu().numpy()
        if mode == "max":
            assert np.array_equal(got, ref_out), "maxima of bf16 values are bf16 values: exact"
        else:
            assert np.all(np.abs(got - ref_out) <= 2.0 ** -8 * np.abs(ref_out) + 1e-30), mode
        gy = (torch.rand(y.F.shape, generator=g) - 0.5).to(torch.bfloat16)
        x.F.grad = None
        y.F.backward(gy.to(device))
        assert x.F.grad.dtype == torch.bfloat16 and x.F.grad.shape == feats.shape
        ref_g = O.pool_backward(gy.float().numpy(), km, coords.shape[0], mode, aux)
        gg = x.F.grad.float().cpu().numpy()
        assert np.all(np.abs(gg - ref_g) <= 2.0 ** -7 * np.abs(ref_g) + 1e-6), mode
    # global pooling + broadcast
    pooled = ME.MinkowskiGlobalAvgPooling()(x)
    assert pooled.F.dtype == torch.bfloat16 and pooled.F.shape == (2, c)
    b = coords[:, 0].numpy()
    want = np.stack([f32[b == i].astype(np.float64).mean(0) for i in range(2)])
    order = pooled.C[:, 0].cpu().numpy()
    assert np.all(np.abs(pooled.F.detach().float().cpu().numpy() - want[order]) <= 2.0 ** -7 * np.abs(want[order]) + 1e-4)
    gmax = ME.MinkowskiGlobalMaxPooling()(x)
    assert np.array_equal(gmax.F.detach().float().cpu().numpy(), np.stack([f32[b == i].max(0) for i in order]))
    for mul, cls in ((False, ME.MinkowskiBroadcastAddition), (True, ME.MinkowskiBroadcastMultiplication)):
        glob = ME.SparseTensor(pooled.F.detach().clone().requires_grad_(True),
                               coordinate_map_key=pooled.coordinate_map_key, coordinate_manager=x.coordinate_manager)
        x.F.grad = None
        y = cls()(x, glob)
        assert y.F.dtype == torch.bfloat16 and y.F.shape == feats.shape
        gl = glob.F.detach().float().cpu().numpy()
        row_of = {int(bi): r for r, bi in enumerate(order)}
        per_row = gl[[row_of[int(v)] for v in b]]
        exact = f32 * per_row if mul else f32 + per_row
        assert np.array_equal(y.F.detach().float().cpu().numpy(), _bf16_round(exact))
        y.F.backward(torch.ones_like(y.F))
        assert x.F.grad.dtype == torch.bfloat16 and glob.F.grad.dtype == torch.bfloat16
        assert x.F.grad.shape == feats.shape and glob.F.grad.shape == (2, c)
        if not mul:
            cnt = np.array([(b == int(bi)).sum() for bi in order], np.float32)
            assert np.array_equal(glob.F.grad.float().cpu().numpy(), _bf16_round(np.repeat(cnt[:, None], c, 1)))


def test_bf16_pooling_does_not_write_past_its_output(device):
    """the round-1 defect, pinned: a bf16 broadcast wrote n * c * 4 bytes into an n * c * 2-byte buffer.  Output and
    a sentinel share one allocation; the sentinel must survive."""
    from minkowskiengine_amd import _lib
    lib = _lib.load()
    n, c = 4096, 32
    buf = torch.full((2 * n * c,), 7.0, dtype=torch.bfloat16, device=device)
    x = torch.rand(n, c, device=device).to(torch.bfloat16)
    glob = torch.rand(1, c, device=device).to(torch.bfloat16)
    rows = torch.zeros(n, dtype=torch.int32, device=device)
    _lib.check(lib.me_broadcast_bf16(x.data_ptr(), glob.data_ptr(), rows.data_ptr(), n, c, 0, buf.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(buf[:n * c].view(n, c), (x.float() + glob.float()).to(torch.bfloat16))
    assert bool((buf[n * c:] == 7.0).all()), "wrote past the n x c bf16 output"


def test_origin_rows_follow_ascending_batch_index(device):
    """Points arriving with batch indices in the order 2, 0, 1 (shuffled rows): row b of the origin map /
    of a global pooling belongs to the b-th smallest batch index, like the reference's GPU map
    (src/coordinate_map_gpu.cu:765-772)."""
    import minkowskiengine_amd as ME
    g = torch.Generator().manual_seed(3)
    coords = make_cloud(900, 12, 3, seed=6, batch=3)
    coords = torch.cat([coords[coords[:, 0] == b] for b in (2, 0, 1)])       # first occurrences: 2, 0, 1
    coords = torch.cat([coords[:300], coords[300:][torch.randperm(coords.shape[0] - 300, generator=g)]]).contiguous()
    feats = torch.rand(coords.shape[0], 6, generator=g)
    x = ME.SparseTensor(feats.to(device), coords.to(device))
    y = ME.MinkowskiGlobalSumPooling()(x)
    assert y.C[:, 0].cpu().tolist() == [0, 1, 2]
    b = coords[:, 0]
    want = torch.stack([feats[b == i].double().sum(0) for i in range(3)]).float()
    assert torch.allclose(y.F.cpu(), want, rtol=1e-5, atol=1e-5)
    out = ME.MinkowskiBroadcastAddition()(x, y)
    assert torch.allclose(out.F.cpu(), feats + want[b.long()], rtol=1e-5, atol=1e-5)
    assert x.coordinate_manager.number_of_unique_batch_indices() == 3


def test_stride_map(device):
    """manager.stride_map (pybind/extern.hpp:803): every input row and the row of the strided map holding its
    floored coordinate — against the oracle's stride map."""
    import minkowskiengine_amd as ME
    coords = make_cloud(5000, 20, 3, seed=12, batch=2, negative=True)
    mgr = ME.CoordinateManager(D=3)
    key, _ = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    skey = mgr.stride(key, [2, 2, 2])
    in_rows, out_rows = mgr.stride_map(key, skey)
    assert in_rows.dtype == torch.int64 and out_rows.dtype == torch.int64
    assert torch.equal(in_rows.cpu(), torch.arange(coords.shape[0]))
    out_c, inv = O.stride_map(coords.numpy(), [2, 2, 2])
    assert np.array_equal(mgr.get_coordinates(skey).cpu().numpy(), out_c)
    assert np.array_equal(out_rows.cpu().numpy(), inv)
    with pytest.raises(RuntimeError):
        mgr.stride_map(skey, key)          # the strided stride must be a multiple of the input stride
