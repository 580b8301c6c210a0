"""ME.utils.ScenePrefetcher: the next scene's coordinate manager (maps, plans) is built by a loader thread on a side
stream while the training thread runs the current step.  Results must equal the lazily built ones bit for bit, for
every scene of a sequence of DIFFERENT scenes, on both hosts."""
import os
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("host_layer")]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_prefetched_scenes_train_like_lazily_built_ones(device):
    import minkowskiengine_amd as ME
    import minkunet as MU
    torch.manual_seed(0)
    net = MU.MinkUNet14(3, 5, D=3).to(device)
    scenes = [MU.synthetic_scene(5000 + 500 * s, grid=48, seed=s).to(device) for s in range(5)]
    feats = [torch.rand(c.shape[0], 3, device=device).bfloat16() for c in scenes]

    def step(x):
        net.zero_grad(set_to_none=True)
        out = net(x)
        out.F.float().square().mean().backward()
        return out.F.detach().clone(), [p.grad.detach().clone() for p in net.parameters()]

    assert not ME.map_prefetch_enabled()
    want = [step(ME.SparseTensor(f, c)) for f, c in zip(feats, scenes)]
    torch.cuda.synchronize()
    got = [step(x) for x in ME.utils.ScenePrefetcher(zip(feats, scenes), depth=2)]
    assert not ME.map_prefetch_enabled()           # restored
    assert len(got) == len(want)
    for (yo, go), (yw, gw) in zip(got, want):
        assert torch.equal(yo, yw)
        for a, b in zip(go, gw):
            assert torch.equal(a, b)


def test_prefetcher_hands_the_iterators_error_to_the_consumer(device):
    import minkowskiengine_amd as ME

    def scenes():
        yield torch.rand(10, 3, device=device), torch.randint(0, 9, (10, 4), device=device, dtype=torch.int32)
        raise ValueError("no more scenes")
    it = iter(ME.utils.ScenePrefetcher(scenes()))
    assert next(it).F.shape[0] <= 10
    with pytest.raises(ValueError):
        next(it)


def test_scenes_dropped_while_the_consumer_stream_is_still_busy(device):
    """ADVICE r4 (cross-stream use-after-free): the scene's feature matrix, coordinates and maps are allocated under the
    LOADER's stream — here by host-to-device copies the iterator issues, as the docstring recommends.  The consumer queues
    a long kernel, then the step on the scene, and drops the scene at once while all of that is still queued; the loader
    immediately builds the next scenes.  Without record_stream on the scene's own tensors the caching allocator hands
    their blocks back to the loader's pool and the next build overwrites them under the queued kernels."""
    import minkowskiengine_amd as ME
    import minkunet as MU
    torch.manual_seed(0)
    conv = ME.MinkowskiConvolution(16, 32, kernel_size=3, dimension=3).to(device)
    n_scenes = 12
    host = [(torch.rand(20000, 16).pin_memory(), MU.synthetic_scene(20000, grid=64, seed=s).pin_memory())
            for s in range(n_scenes)]

    def lazily(f, c):
        with torch.no_grad():
            return conv(ME.SparseTensor(f.to(device), c.to(device))).F.clone()
    want = [lazily(f, c) for f, c in host]
    torch.cuda.synchronize()

    def scenes():                 # runs in the loader thread under the loader's stream: the copies allocate there
        for f, c in host:
            yield f.to(device, non_blocking=True), c.to(device, non_blocking=True)
    busy = torch.rand(4096, 4096, device=device)
    got = []
    side = torch.cuda.Stream()    # the consumer is NOT on the stream that was current when iteration began
    it = iter(ME.utils.ScenePrefetcher(scenes(), depth=3))
    with torch.cuda.stream(side):
        for x in it:
            for _ in range(6):
                busy = (busy @ busy).clamp_(-1, 1)           # ~ms of queued work ahead of the step
            with torch.no_grad():
                got.append(conv(x).F.clone())
            del x                                            # dropped while everything above is still queued
    torch.cuda.synchronize()
    assert len(got) == n_scenes
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"scene {i} was corrupted"
