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

pytestmark = pytest.mark.gpu
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
