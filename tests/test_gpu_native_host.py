"""The native host layer (csrc_host/ -> minkowskiengine_amd/_me_host.so: C++ CoordinateMapManager, operators and autograd
functions — what the reference builds as MinkowskiEngineBackend._C) against its Python twin (backend.py): the same
kernels on the same plans, so every result must be BIT-identical; plus the reference-signature operators, the build
recipe (map prefetch) and a hipGraph capture of a whole training step."""
import numpy as np
import pytest
import torch

from helpers import assert_close, make_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture
def native():
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import host
    if host.native_module() is None:
        pytest.fail(f"native host layer not built / not loadable: {host.native_error()}")
    prev = ME.get_host()
    yield host.native_module()
    ME.set_host(prev)
    ME.clear_global_coordinate_manager()


def _on_host(name, fn):
    import minkowskiengine_amd as ME
    ME.set_host(name)
    ME.clear_global_coordinate_manager()
    try:
        return fn()
    finally:
        ME.clear_global_coordinate_manager()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,ks,stride,D", [(64, 128, 3, 1, 3), (32, 32, 2, 2, 3), (3, 32, 5, 1, 3), (32, 64, 3, 1, 4),
                                                   (96, 20, 1, 1, 3)])
def test_convolution_layer_is_bit_identical_on_both_hosts(device, native, dtype, cin, cout, ks, stride, D):
    import minkowskiengine_amd as ME
    coords = make_cloud(4000, 10 if D == 4 else 16, D, seed=cin + cout, batch=2, negative=True).to(device)
    g = torch.Generator().manual_seed(0)
    feats = (torch.rand(coords.shape[0], cin, generator=g) - 0.4).to(dtype)
    w = torch.rand(ks ** D, cin, cout, generator=g) - 0.5

    def run():
        conv = ME.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, dimension=D).to(device)
        with torch.no_grad():
            conv.kernel.copy_(w.view(conv.kernel.shape))
        x = ME.SparseTensor(feats.to(device), coords, requires_grad=True)
        y = conv(x)
        gy = torch.rand(y.F.shape, generator=torch.Generator().manual_seed(1)).to(dtype).to(device)
        y.F.backward(gy)
        km = x.coordinate_manager.kernel_map(x.coordinate_map_key, y.coordinate_map_key, stride=stride, kernel_size=ks)
        return y.C.clone(), y.F.detach().clone(), x.F.grad.clone(), conv.kernel.grad.clone(), km

    a = _on_host("python", run)
    b = _on_host("native", run)
    for u, v in zip(a[:4], b[:4]):
        assert torch.equal(u, v)
    assert sorted(a[4]) == sorted(b[4])
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k])


def test_minkunet_training_step_is_bit_identical_on_both_hosts(device, native):
    """MinkUNet14 forward + loss + backward on the native host (C++ autograd functions for convolutions and batch
    norms) equals the Python host bit for bit: output, loss and every parameter gradient."""
    import os
    import sys
    import minkowskiengine_amd as ME
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import minkunet as MU
    coords = make_cloud(6000, 40, 3, seed=1).to(device)
    g = torch.Generator().manual_seed(2)
    feats = torch.rand(coords.shape[0], 3, generator=g).to(device)
    labels = torch.randint(0, 20, (coords.shape[0],), generator=g).to(device)

    def run(dtype):
        torch.manual_seed(0)
        net = MU.MinkUNet14(3, 20, D=3).to(device)
        x = ME.SparseTensor(feats.to(dtype), coords)
        out = net(x)
        loss = MU.cross_entropy(out.F.float(), labels)
        loss.backward()
        return out.F.detach().clone(), loss.detach().clone(), [p.grad.clone() for p in net.parameters()]

    for dtype in (torch.float32, torch.bfloat16):
        a = _on_host("python", lambda: run(dtype))
        b = _on_host("native", lambda: run(dtype))
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for u, v in zip(a[2], b[2]):
            assert torch.equal(u, v)


def test_pooling_broadcast_pruning_union_on_the_native_host(device, native):
    import minkowskiengine_amd as ME
    coords = make_cloud(5000, 20, 3, seed=3, batch=3).to(device)
    g = torch.Generator().manual_seed(4)
    feats = torch.rand(coords.shape[0], 16, generator=g).to(device)

    def run():
        x = ME.SparseTensor(feats.clone(), coords, requires_grad=True)    # (a fresh leaf: .grad must not accumulate over runs)
        outs = []
        for layer in (ME.MinkowskiMaxPooling(kernel_size=3, stride=2, dimension=3),
                      ME.MinkowskiAvgPooling(kernel_size=2, stride=2, dimension=3),
                      ME.MinkowskiSumPooling(kernel_size=3, stride=1, dimension=3)):
            y = layer(x)
            outs += [y.C.clone(), y.F.detach().clone()]
        p = ME.MinkowskiAvgPooling(kernel_size=2, stride=2, dimension=3)(x)
        up = ME.MinkowskiPoolingTranspose(kernel_size=2, stride=2, dimension=3)(p)
        gp = ME.MinkowskiGlobalAvgPooling()(x)
        gm = ME.MinkowskiGlobalMaxPooling()(x)
        b = ME.MinkowskiBroadcastMultiplication()(x, gp)
        keep = feats[:, 0] > 0.5
        pr = ME.MinkowskiPruning()(x, keep)
        total = up.F.sum() + b.F.sum() + gm.F.sum() + pr.F.sum() * 2
        total.backward()
        outs += [up.F.detach().clone(), gp.F.detach().clone(), gm.F.detach().clone(), b.F.detach().clone(),
                 pr.C.clone(), pr.F.detach().clone(), x.F.grad.clone()]
        return outs

    a = _on_host("python", run)
    b = _on_host("native", run)
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), (i, float((u.double() - v.double()).abs().max()), int((u != v).sum()), u.numel())


def test_reference_signature_operators_and_manager_of_the_native_module(device, native):
    """The native module exposes the reference's names and positional signatures (pybind/extern.hpp:53-181, 767-806):
    manager methods, ConvolutionForwardGPU / BackwardGPU on keys; results against the oracle."""
    from oracle import me_oracle as O
    C = native
    coords = make_cloud(3000, 14, 3, seed=5, batch=2, negative=True)
    g = torch.Generator().manual_seed(6)
    feats = torch.rand(coords.shape[0], 16, generator=g)
    w = torch.rand(27, 16, 32, generator=g) - 0.5
    mgr = C.CoordinateMapManagerGPU_c10(C.MinkowskiAlgorithm.DEFAULT, 0)
    key, (umap, imap) = mgr.insert_and_map(coords.to(device), [1, 1, 1], "")
    assert key.get_key() == ([1, 1, 1], "") and mgr.size(key) == coords.shape[0]
    assert torch.equal(mgr.get_coordinates(key).cpu(), coords)
    out_key = C.CoordinateMapKey(4)
    y = C.ConvolutionForwardGPU(feats.to(device), w.to(device), [3] * 3, [1] * 3, [1] * 3, C.RegionType.HYPER_CUBE,
                                torch.IntTensor(), False, C.ConvolutionMode.DEFAULT, key, out_key, mgr)
    assert out_key.is_key_set() and out_key == key
    co = coords.numpy()
    _, km = O.kernel_map(co, co, O.make_region(3, 3))
    assert_close(y, O.conv_forward(feats.numpy(), w.numpy(), km, len(co)))
    gy = torch.rand(y.shape, generator=g)
    gi, gw = C.ConvolutionBackwardGPU(feats.to(device), gy.to(device), w.to(device), [3] * 3, [1] * 3, [1] * 3,
                                      C.RegionType.HYPER_CUBE, torch.IntTensor(), C.ConvolutionMode.DEFAULT, key, out_key,
                                      mgr)
    ri, rw = O.conv_backward(feats.numpy(), gy.numpy(), w.numpy(), km)
    assert_close(gi, ri)
    assert_close(gw, rw)
    O.assert_same_kernel_map(mgr.kernel_map(key, out_key, [3] * 3, [1] * 3, [1] * 3, C.RegionType.HYPER_CUBE,
                                            torch.IntTensor(), False, False), km)
    skey = mgr.stride(key, [2, 2, 2])
    assert skey.get_tensor_stride() == [2, 2, 2]
    assert np.array_equal(mgr.get_coordinates(skey).cpu().numpy(), O.stride_map(co, [2] * 3)[0])
    with pytest.raises(RuntimeError):
        mgr.insert_and_map(coords, [1, 1, 1], "")          # CPU coordinates
    with pytest.raises(RuntimeError):
        C.CoordinateMapKey(4).get_key()                    # unset key


def test_build_recipe_replay_on_the_native_host(device, native):
    """A fresh scene's manager replays the previous scene's request log in ONE native call (strided maps, kernel maps,
    tile plans, weight-gradient geometries): the following training step adds no request, and its results equal the
    lazily built ones bit for bit."""
    import os
    import sys
    import minkowskiengine_amd as ME
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import minkunet as MU
    ME.set_host("native")
    coords = make_cloud(8000, 40, 3, seed=7).to(device)
    feats = torch.rand(coords.shape[0], 3, generator=torch.Generator().manual_seed(8)).to(device).bfloat16()
    torch.manual_seed(0)
    net = MU.MinkUNet14(3, 20, D=3).to(device)

    def step(x):
        net.zero_grad(set_to_none=True)
        out = net(x)
        out.F.float().sum().backward()
        return out.F.detach().clone(), [p.grad.clone() for p in net.parameters()]

    x1 = ME.SparseTensor(feats, coords)
    lazy = step(x1)
    recipe = x1.coordinate_manager.recipe()
    assert any(r.startswith("kernel_map;") for r in recipe) and any(r.startswith("conv_cfg;") for r in recipe)
    x2 = ME.SparseTensor(feats, coords)
    done = x2.coordinate_manager.prefetch(recipe)
    assert done == len(recipe)
    before = list(x2.coordinate_manager.recipe())
    replayed = step(x2)
    assert x2.coordinate_manager.recipe() == before        # every map / plan was already there
    assert torch.equal(lazy[0], replayed[0])
    for u, v in zip(lazy[1], replayed[1]):
        assert torch.equal(u, v)
    # a chain of scenes: the prefetched manager's OWN recipe is complete, so scene 3 can be prefetched from
    # scene 2 (and 4 from 3) — not only every other scene
    assert sorted(before) == sorted(recipe)
    x3 = ME.SparseTensor(feats, coords)
    assert x3.coordinate_manager.prefetch(x2.coordinate_manager.recipe()) == len(recipe)
    x4 = ME.SparseTensor(feats, coords)
    assert x4.coordinate_manager.prefetch(x3.coordinate_manager.recipe()) == len(recipe)
