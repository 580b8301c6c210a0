"""Packed weight images: the multi-layer pack launch (csrc/pack.hip, backend._WeightPacker) produces bit-identical
images to the single-layer pack kernels, follows in-place weight updates through the tensor's version counter, and never
serves a stale image for a temporary weight tensor."""
import pytest
import torch

from helpers import make_cloud

pytestmark = pytest.mark.gpu

SHAPES = [(27, 64, 128), (27, 96, 96), (8, 32, 32), (125, 3, 32), (27, 20, 24), (1, 256, 128), (27, 384, 256)]


@pytest.mark.parametrize("mode", ["bf16", "f32x3"])
def test_multi_pack_images_are_bit_identical(device, mode):
    from minkowskiengine_amd import _lib, backend as MEB
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    pk = MEB._WeightPacker(device)
    weights, want = [], []
    for (k, ci, co) in SHAPES:
        if mode == "f32x3" and ci % 8:
            continue
        w = (torch.rand(k, ci, co, generator=g) - 0.5).to(device)
        weights.append(w)
        for transposed in (False, True):
            cs, cd = (co, ci) if transposed else (ci, co)
            if mode == "f32x3" and cs % 8:
                continue
            if mode == "bf16":
                elems = int(lib.me_conv_packed_weight_elems_bf16(k, cs, cd))
                ref = torch.zeros(elems, dtype=torch.bfloat16, device=device)
                _lib.check(lib.me_conv_pack_weights_bf16(w.data_ptr(), 1, k, cs, cd, int(transposed), ref.data_ptr(),
                                                         torch.cuda.current_stream().cuda_stream))
                m = _lib.ME_PACK_BF16
            else:
                elems = int(lib.me_conv_packed_weight_elems_f32x3(k, cs, cd))
                ref = torch.zeros(elems, dtype=torch.bfloat16, device=device)
                _lib.check(lib.me_conv_pack_weights_f32x3(w.data_ptr(), k, cs, cd, int(transposed), ref.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream))
                m = _lib.ME_PACK_F32X3
            want.append((w, m, transposed, cs, cd, elems, ref))
    for w, m, tr, cs, cd, elems, ref in want:       # entries are created one by one (single-job launches)
        got = pk.get(w, m, tr, cs, cd, elems)
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    # every weight moves on (as after an optimizer step): the next request repacks ALL images in one launch
    for w in weights:
        w.mul_(1.5)
    first = want[0]
    pk.get(first[0], first[1], first[2], first[3], first[4], first[5])
    assert len(pk.table[3]) == len(want)
    torch.cuda.synchronize()
    for w, m, tr, cs, cd, elems, _ in want:
        ref = torch.zeros(elems, dtype=torch.bfloat16, device=device)
        k = w.shape[0]
        st = torch.cuda.current_stream().cuda_stream
        if m == _lib.ME_PACK_BF16:
            _lib.check(lib.me_conv_pack_weights_bf16(w.data_ptr(), 1, k, cs, cd, int(tr), ref.data_ptr(), st))
        else:
            _lib.check(lib.me_conv_pack_weights_f32x3(w.data_ptr(), k, cs, cd, int(tr), ref.data_ptr(), st))
        ent = pk.entries[(w.data_ptr(), m, tr)]
        assert ent.version == w._version
        assert torch.equal(ent.packed.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cached_images_follow_weight_updates_and_temporaries(device, dtype, monkeypatch):
    """A layer's output after an in-place weight update equals a fresh layer's with the same weights; temporary weight
    tensors that reuse a freed address are packed again (cache on = cache off, bit for bit)."""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import backend as MEB
    coords = make_cloud(3000, 14, 3, seed=2).to(device)
    g = torch.Generator().manual_seed(1)
    feats = torch.rand(coords.shape[0], 64, generator=g).to(device).to(dtype)
    x = ME.SparseTensor(feats, coords)
    conv = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).to(device)
    mgr = x.coordinate_manager._manager
    km = mgr._kernel_map(x.coordinate_map_key, x.coordinate_map_key, [3] * 3, [1] * 3, [1] * 3, MEB.RegionType.HYPER_CUBE,
                         None, False, False)
    outs = {}
    for cache in (True, False):
        monkeypatch.setattr(MEB, "_PACK_CACHE", cache)
        torch.manual_seed(5)
        with torch.no_grad():
            conv.kernel.copy_(torch.rand(conv.kernel.shape, device=device) - 0.5)
        res = [conv(x).F.clone()]
        with torch.no_grad():
            conv.kernel.add_(0.25)                     # in place: version bump, same storage
        res.append(conv(x).F.clone())
        for i in range(4):                             # temporaries: freed and re-allocated at the same address
            w = (torch.full((27, 64, 128), 0.01 * (i + 1), device=device))
            res.append(MEB._conv_forward(feats, w, km).clone())
            del w
        outs[cache] = res
    assert not torch.equal(outs[True][0], outs[True][1])
    assert not torch.equal(outs[True][2], outs[True][3])
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_data_writes_need_the_invalidate_call_and_optimizer_steps_do_not(device, host_layer, dtype):
    """a write through `p.data` does not bump the version counter the cache is validated by.  The cache
    epoch covers it: `invalidate_packed_weights()` (explicit), every torch.optim step (global post-hook) and
    distributed.broadcast_parameters.  After either, the layer's output equals a fresh layer's with the same weights."""
    import minkowskiengine_amd as ME
    coords = make_cloud(3000, 14, 3, seed=2).to(device)
    feats = torch.rand(coords.shape[0], 64, generator=torch.Generator().manual_seed(1)).to(device).to(dtype)
    x = ME.SparseTensor(feats, coords)
    conv = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).to(device)

    def fresh(w):
        c = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).to(device)
        with torch.no_grad():
            c.kernel.copy_(w)
        return c(ME.SparseTensor(feats, coordinate_map_key=x.coordinate_map_key,
                                 coordinate_manager=x.coordinate_manager)).F

    y0 = conv(x).F.clone()
    v = conv.kernel._version
    conv.kernel.data.add_(0.125)                               # invisible to the version counter
    assert conv.kernel._version == v
    ME.invalidate_packed_weights()
    y1 = conv(x).F.clone()
    assert not torch.equal(y0, y1) and torch.equal(y1, fresh(conv.kernel.detach()))
    # an optimizer that writes through .data (Apex / DeepSpeed style): the step post-hook advances the epoch
    class DataSGD(torch.optim.Optimizer):
        def __init__(self, params):
            super().__init__(params, {})

        def step(self, closure=None):
            for gr in self.param_groups:
                for p in gr["params"]:
                    p.data.mul_(0.5)
    opt = DataSGD(conv.parameters())
    v = conv.kernel._version
    opt.step()
    assert conv.kernel._version == v
    y2 = conv(x).F.clone()
    assert not torch.equal(y1, y2) and torch.equal(y2, fresh(conv.kernel.detach()))
    # the hook invalidates only the stepping optimizer's OWN parameters: a frozen layer that is not in the
    # optimizer keeps its image — shown with a .data write on it that nothing announces: still the old output after the
    # other layer's optimizer step, the new one after the targeted call
    frozen = ME.MinkowskiConvolution(64, 128, kernel_size=3, dimension=3).to(device)
    z0 = frozen(x).F.clone()
    frozen.kernel.data.add_(0.25)
    opt.step()
    assert torch.equal(frozen(x).F, z0)
    ME.invalidate_packed_weights([frozen.kernel])
    z1 = frozen(x).F.clone()
    assert not torch.equal(z1, z0) and torch.equal(z1, fresh(frozen.kernel.detach()))
    # reset_parameters writes in place on the parameter itself: seen by the version counter
    torch.manual_seed(3)
    conv.reset_parameters()
    assert conv.kernel._version > v
    assert torch.equal(conv(x).F, fresh(conv.kernel.detach()))


def test_objects_stay_with_the_host_that_made_them(device):
    """operators and new keys are resolved from the input tensor's manager / key, not from the global host
    switch — a SparseTensor made under one host keeps working after set_host() (pooling, broadcast, pruning, union,
    convolution, batch norm)."""
    import minkowskiengine_amd as ME
    coords = make_cloud(2000, 12, 3, seed=4, batch=2).to(device)
    feats = torch.rand(coords.shape[0], 16, generator=torch.Generator().manual_seed(2)).to(device)
    prev = ME.get_host()
    results = {}
    try:
        for made_under, run_under in (("python", "native"), ("native", "python"), ("python", "python")):
            ME.set_host(made_under)
            x = ME.SparseTensor(feats, coords)
            conv = ME.MinkowskiConvolution(16, 16, kernel_size=2, stride=2, dimension=3).to(device)
            torch.manual_seed(0)
            conv.reset_parameters()
            ME.set_host(run_under)
            y = conv(x)
            p = ME.MinkowskiMaxPooling(kernel_size=3, stride=2, dimension=3)(x)
            b = ME.MinkowskiBroadcastMultiplication()(x, ME.MinkowskiGlobalAvgPooling()(x))
            pr = ME.MinkowskiPruning()(x, x.F[:, 0] > 0.5)
            u = ME.MinkowskiUnion()(x, pr)
            n = ME.MinkowskiBatchNorm(16).to(device)(y)
            results[(made_under, run_under)] = [t.F.clone() for t in (y, p, b, pr, u, n)]
            assert bool(x.coordinate_manager._native) == (made_under == "native")
    finally:
        ME.set_host(prev)
    base = results[("python", "python")]
    for k, r in results.items():
        for a, c in zip(base, r):
            assert torch.equal(a, c), k
