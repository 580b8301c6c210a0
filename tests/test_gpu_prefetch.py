"""Map prefetch (set_map_prefetch / CoordinateManager.prefetch): replaying the previous scene's build requests on a
new scene must build exactly what the network will ask for — no further build request during the step — and must not
change any result."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def _step(net, x, labels):
    import minkunet as MU
    net.zero_grad(set_to_none=True)
    out = net(x)
    loss = MU.cross_entropy(out.F, labels)
    loss.backward()
    return out.F.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_prefetch_builds_everything_and_changes_nothing(device, dtype):
    import minkowskiengine_amd as ME
    import minkunet as MU
    torch.manual_seed(0)
    net = MU.MinkUNet14(3, 5, D=3).to(device)
    scenes = [MU.synthetic_scene(6000, grid=48, seed=s).to(device) for s in (1, 2)]
    feats = [torch.rand(c.shape[0], 3, device=device).to(dtype) for c in scenes]
    labels = [torch.randint(0, 5, (c.shape[0],), device=device) for c in scenes]
    try:
        ME.set_map_prefetch(False)
        xa = ME.SparseTensor(feats[0], scenes[0])
        _step(net, xa, labels[0])
        recipe = xa.coordinate_manager.recipe()
        kinds = {op[0] for op in recipe}
        assert {"stride", "kernel_map", "conv_cfg", "wgrad_cfg"} <= kinds, kinds
        # scene B without prefetch: the reference results
        xb = ME.SparseTensor(feats[1], scenes[1])
        want_out, want_grads = _step(net, xb, labels[1])
        assert len(xb.coordinate_manager.recipe()) == len(recipe)
        # scene B with prefetch from scene A's manager (the most recent one is xb's: same requests)
        ME.set_map_prefetch(True)
        xc = ME.SparseTensor(feats[1], scenes[1])
        mgr = xc.coordinate_manager
        built = len(mgr.recipe())
        assert built == len(recipe), (built, len(recipe))
        got_out, got_grads = _step(net, xc, labels[1])
        assert len(mgr.recipe()) == built, "the step asked for a map / plan that the prefetch had not built"
        assert torch.equal(got_out, want_out)
        for n in want_grads:
            assert torch.equal(got_grads[n], want_grads[n]), n
    finally:
        ME.set_map_prefetch(False)


def test_prefetch_skips_requests_that_do_not_apply(device):
    """A recipe from another network / dimension: unknown maps are skipped, nothing raises."""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import backend as MEB
    coords = torch.randint(0, 20, (500, 4), dtype=torch.int32, device=device)
    coords[:, 0] = 0
    x = ME.SparseTensor(torch.rand(500, 4, device=device), coords)
    bogus = [("stride", ((8, 8, 8), ""), (2, 2, 2), ""),
             ("kernel_map", (((4, 4, 4), ""), ((4, 4, 4), ""), (3, 3, 3), (1, 1, 1), (1, 1, 1),
                             int(MEB.RegionType.HYPER_CUBE), False, False)),
             ("conv_cfg", ("nope",), "out", 16, 16, False), ("wgrad_cfg", ("nope",), 16, 16, False)]
    assert x.coordinate_manager.prefetch(bogus) == 0
    good = [("stride", ((1, 1, 1), ""), (2, 2, 2), "")]
    assert x.coordinate_manager.prefetch(good) == 1
    assert x.coordinate_manager.exists_coordinate_map_key(ME.CoordinateMapKey([2, 2, 2], ""))


def test_failed_plan_batch_leaves_no_unbuilt_plan_behind(device):
    """ADVICE r4 (manager.cpp PlanBatch): plans of a recipe replay are cached before they are built together.  When that
    build throws, the cached plans hold uninitialised arrays: they must leave the caches, so that a caller who catches the
    error gets correct results from the same manager afterwards (plans rebuilt on use), not garbage gathers."""
    import minkowskiengine_amd as ME
    import minkunet as MU
    from minkowskiengine_amd import host
    native = host.native_module()
    if native is None:
        pytest.fail(f"native host layer not available: {host.native_error()}")
    prev = ME.get_host()
    ME.set_host("native")
    try:
        torch.manual_seed(0)
        net = MU.MinkUNet14(3, 5, D=3).to(device)
        scenes = [MU.synthetic_scene(6000, grid=48, seed=s).to(device) for s in (1, 2)]
        feats = [torch.rand(c.shape[0], 3, device=device).to(torch.bfloat16) for c in scenes]
        labels = [torch.randint(0, 5, (c.shape[0],), device=device) for c in scenes]
        ME.set_map_prefetch(False)
        xa = ME.SparseTensor(feats[0], scenes[0])
        _step(net, xa, labels[0])
        recipe = xa.coordinate_manager.recipe()
        xb = ME.SparseTensor(feats[1], scenes[1])
        want_out, want_grads = _step(net, xb, labels[1])
        # the same scene on a fresh manager: the replay's batched plan build fails once
        xc = ME.SparseTensor(feats[1], scenes[1])
        native.debug_fail_next_plan_batch()
        with pytest.raises(RuntimeError, match="injected failure"):
            xc.coordinate_manager.prefetch(recipe)
        got_out, got_grads = _step(net, xc, labels[1])          # (plans built lazily now)
        assert torch.equal(got_out, want_out)
        for n in want_grads:
            assert torch.equal(got_grads[n], want_grads[n]), n
    finally:
        ME.set_map_prefetch(False)
        ME.set_host(prev)


def test_prefetch_tags_keep_two_networks_apart(device):
    """two networks of the same dimension alternate in one process.  Untagged, the small network's scenes replay
    the big network's (longest) recipe; inside `map_prefetch_tag` each replays its own."""
    import gc
    import minkowskiengine_amd as ME
    import minkunet as MU
    from minkowskiengine_amd import coordinate_manager as CM
    torch.manual_seed(0)
    big = MU.MinkUNet14(3, 5, D=3).to(device)
    small = ME.MinkowskiConvolution(3, 16, kernel_size=3, dimension=3).to(device)
    scenes = [MU.synthetic_scene(5000, grid=48, seed=s).to(device) for s in (1, 2, 3, 4)]
    feats = [torch.rand(c.shape[0], 3, device=device) for c in scenes]
    CM._published_recipes.clear()
    CM._recent_managers.clear()
    try:
        ME.set_map_prefetch(True)
        lens = {}
        for rnd in range(2):
            for name, net, i in (("big", big, 2 * rnd), ("small", small, 2 * rnd + 1)):
                with ME.map_prefetch_tag(name):
                    x = ME.SparseTensor(feats[i], scenes[i])
                    before = len(x.coordinate_manager.recipe())      # what the replay built
                    net(x)
                    after = len(x.coordinate_manager.recipe())
                    lens.setdefault(name, []).append((before, after))
                    del x
                    gc.collect()
        # second round: each network's scene was prefetched from ITS OWN previous log, completely and with nothing else
        assert lens["small"][1][0] == lens["small"][0][1] == lens["small"][1][1], lens
        assert lens["big"][1][0] == lens["big"][0][1] == lens["big"][1][1], lens
        assert lens["small"][1][0] < lens["big"][1][0], lens
    finally:
        ME.set_map_prefetch(False)
        CM._published_recipes.clear()
        CM._recent_managers.clear()
