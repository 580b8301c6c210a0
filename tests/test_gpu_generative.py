"""SURVEY 8f rank 4 on the GPU: generative transposed convolution, expanding convolution, pruning and union
against the fixture produced by the reference's CPU operators (tests/golden/make_golden_generative.py).
New maps are compared as coordinate sets, features after matching rows by coordinate; tolerance 1e-4."""
import os

import numpy as np
import pytest
import torch

from oracle import me_oracle as O
from helpers import GOLDEN_DIR, assert_close, make_cloud, row_mapping

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("host_layer")]   # both host layers


def _z():
    return np.load(os.path.join(GOLDEN_DIR, "generative_3d.npz"))


@pytest.mark.parametrize("name,ks,st", [("gen_k2s2", 2, 2), ("gen_k3s1", 3, 1)])
def test_generative_transposed_convolution(device, name, ks, st):
    import minkowskiengine_amd as ME
    z = _z()
    ts_in = 2 if st == 2 else 1
    feats = torch.from_numpy(z[f"{name}/feats"]).to(device).requires_grad_(True)
    x = ME.SparseTensor(feats, torch.from_numpy(z[f"{name}/coords"]).to(device), tensor_stride=ts_in)
    conv = ME.MinkowskiGenerativeConvolutionTranspose(6, 5, kernel_size=ks, stride=st, dimension=3)
    with torch.no_grad():
        conv.kernel.copy_(torch.from_numpy(z[f"{name}/kernel"]))
    conv = conv.to(device)
    y = conv(x)
    assert y.tensor_stride == z[f"{name}/out_tensor_stride"].tolist()
    assert y.coordinate_map_key != x.coordinate_map_key
    m = row_mapping(y.C.cpu().numpy(), z[f"{name}/out_coords"])
    assert_close(y.F.detach().cpu().numpy(), z[f"{name}/out"][m])
    y.F.backward(torch.from_numpy(z[f"{name}/grad_out"][m]).to(device))
    assert_close(feats.grad.cpu().numpy(), z[f"{name}/grad_in"])
    assert_close(conv.kernel.grad.cpu().numpy(), z[f"{name}/grad_kernel"])
    # a second generative layer on the same input creates another map (random string id), never reuses
    y2 = conv(x)
    assert y2.coordinate_map_key != y.coordinate_map_key and y2.F.shape == y.F.shape


def test_transposed_convolution_creates_missing_map(device):
    """Without expand_coordinates a transposed convolution reuses the map of the output stride when it exists
    and generates it otherwise (src/coordinate_map_manager.cpp:450-463)."""
    import minkowskiengine_amd as ME
    z = _z()
    x = ME.SparseTensor(torch.from_numpy(z["gen_k2s2/feats"]).to(device),
                        torch.from_numpy(z["gen_k2s2/coords"]).to(device), tensor_stride=2)
    up = ME.MinkowskiConvolutionTranspose(6, 5, kernel_size=2, stride=2, dimension=3).to(device)
    with torch.no_grad():
        up.kernel.copy_(torch.from_numpy(z["gen_k2s2/kernel"]).to(device))
    y = up(x)                                   # no stride-1 map yet: generated, same result as the generative layer
    m = row_mapping(y.C.cpu().numpy(), z["gen_k2s2/out_coords"])
    assert_close(y.F.detach().cpu().numpy(), z["gen_k2s2/out"][m])
    y2 = up(x)                                  # now the stride-1 map exists: reused
    assert y2.coordinate_map_key == y.coordinate_map_key


def test_expanding_convolution(device):
    import minkowskiengine_amd as ME
    z = _z()
    x = ME.SparseTensor(torch.from_numpy(z["expand/feats"]).to(device), torch.from_numpy(z["expand/coords"]).to(device))
    conv = ME.MinkowskiConvolution(4, 3, kernel_size=3, stride=2, expand_coordinates=True, dimension=3)
    with torch.no_grad():
        conv.kernel.copy_(torch.from_numpy(z["expand/kernel"]))
    y = conv.to(device)(x)
    assert y.tensor_stride == [2, 2, 2]
    m = row_mapping(y.C.cpu().numpy(), z["expand/out_coords"])
    assert_close(y.F.detach().cpu().numpy(), z["expand/out"][m])


def test_pruning(device):
    import minkowskiengine_amd as ME
    z = _z()
    feats = torch.from_numpy(z["prune/feats"]).to(device).requires_grad_(True)
    x = ME.SparseTensor(feats, torch.from_numpy(z["prune/coords"]).to(device))
    keep = torch.from_numpy(z["prune/keep"])
    y = ME.MinkowskiPruning()(x, keep.to(device))
    assert y.coordinate_map_key.get_key()[1].startswith("pruned") and y.tensor_stride == [1, 1, 1]
    # kept rows in input-row order: bit-exact
    assert np.array_equal(y.C.cpu().numpy(), z["prune/coords"][z["prune/keep"]])
    m = row_mapping(y.C.cpu().numpy(), z["prune/out_coords"])
    assert np.array_equal(y.F.detach().cpu().numpy(), z["prune/out"][m])
    y.F.backward(torch.from_numpy(z["prune/grad_out"][m]).to(device))
    assert np.array_equal(feats.grad.cpu().numpy(), z["prune/grad_in"])
    # a convolution on the pruned tensor uses the pruned map
    conv = ME.MinkowskiConvolution(5, 4, kernel_size=3, dimension=3).to(device)
    out = conv(y)
    _, km = O.kernel_map(y.C.cpu().numpy(), y.C.cpu().numpy(), O.make_region(3, 3))
    ref = O.conv_forward(y.F.detach().cpu().numpy(), conv.kernel.detach().cpu().numpy(), km, y.F.shape[0])
    assert_close(out.F.detach().cpu().numpy(), ref)
    # everything pruned: an empty tensor, not an error
    none = ME.MinkowskiPruning()(x, torch.zeros(x.F.shape[0], dtype=torch.bool, device=device))
    assert none.F.shape == (0, 5)


def test_union(device):
    import minkowskiengine_amd as ME
    z = _z()
    fa = torch.from_numpy(z["union/fa"]).to(device).requires_grad_(True)
    fb = torch.from_numpy(z["union/fb"]).to(device).requires_grad_(True)
    a = ME.SparseTensor(fa, torch.from_numpy(z["union/a"]).to(device))
    b = ME.SparseTensor(fb, torch.from_numpy(z["union/b"]).to(device), coordinate_manager=a.coordinate_manager)
    u = ME.MinkowskiUnion()(a, b)
    m = row_mapping(u.C.cpu().numpy(), z["union/out_coords"])
    assert_close(u.F.detach().cpu().numpy(), z["union/out"][m], 1e-6, 1e-6)
    g = torch.rand(u.F.shape, generator=torch.Generator().manual_seed(0)).to(device)
    u.F.backward(g)
    # every input row receives the gradient of its union row
    uc = {tuple(r): i for i, r in enumerate(u.C.cpu().numpy().tolist())}
    ia = np.array([uc[tuple(r)] for r in z["union/a"].tolist()])
    ib = np.array([uc[tuple(r)] for r in z["union/b"].tolist()])
    assert np.array_equal(fa.grad.cpu().numpy(), g.cpu().numpy()[ia])
    assert np.array_equal(fb.grad.cpu().numpy(), g.cpu().numpy()[ib])


def test_generative_large(device):
    """100k voxels at tensor stride 2, k = 2 generative up-sampling: 800k candidates, coordinate set against the
    oracle, and the round trip down (k = 2, s = 2) recovers the input map."""
    import minkowskiengine_amd as ME
    coords = make_cloud(100000, 60, 3, seed=1)
    coords[:, 1:] *= 2
    x = ME.SparseTensor(torch.rand(100000, 16).to(device), coords.to(device), tensor_stride=2)
    up = ME.MinkowskiGenerativeConvolutionTranspose(16, 8, kernel_size=2, stride=2, dimension=3).to(device)
    y = up(x)
    ref = O.stride_region(coords.numpy(), O.make_region(3, 2, 1, 1))
    assert y.C.shape[0] == len(ref) == 800000
    row_mapping(y.C.cpu().numpy(), ref)
    down = ME.MinkowskiConvolution(8, 4, kernel_size=2, stride=2, dimension=3).to(device)
    zt = down(y)
    assert zt.coordinate_map_key == x.coordinate_map_key
