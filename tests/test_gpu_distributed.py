"""The multi-GPU path with the REAL HIP kernels, on the one GPU a test box has: two ranks (two processes) share
cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the collectives are the same torch.distributed
calls).  Each rank owns its own scene, coordinate manager and kernel maps; DistributedDataParallel averages the
gradients.  Checked against one process that runs both scenes itself."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import assert_close, make_cloud

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(rank):
    coords = make_cloud(3000 + 500 * rank, 16, 3, seed=40 + rank)
    g = torch.Generator().manual_seed(50 + rank)
    return coords, torch.rand(coords.shape[0], 8, generator=g)


def _conv_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import distributed as D
    r, w, lr = D.init_from_env()                       # 1 GPU, 2 ranks -> gloo
    assert D.backend_name() == "gloo"
    dev = D.local_device(lr)
    assert dev.index == 0
    torch.manual_seed(1000 + rank)                     # ranks start from different weights; DDP broadcasts rank 0's
    conv = ME.MinkowskiConvolution(8, 16, kernel_size=3, dimension=3, bias=True).to(dev)
    net = D.data_parallel(conv, dev)
    coords, feats = _scene(rank)
    x = ME.SparseTensor(feats.to(dev), coords.to(dev))
    y = net(x)
    (y.F * y.F).sum().backward()
    torch.cuda.synchronize()
    out[rank] = (conv.kernel.detach().cpu(), conv.bias.detach().cpu(), conv.kernel.grad.cpu(), conv.bias.grad.cpu())
    D.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_on_one_gpu_average_the_gradients_of_the_hip_convolution(device):
    import minkowskiengine_amd as ME
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_conv_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    k0, b0, gk0, gb0 = out[0]
    k1, b1, gk1, gb1 = out[1]
    assert torch.equal(k0, k1) and torch.equal(b0, b1), "DDP did not broadcast rank 0's parameters"
    assert torch.equal(gk0, gk1) and torch.equal(gb0, gb1), "ranks hold different averaged gradients"
    # one process, both scenes, same weights
    conv = ME.MinkowskiConvolution(8, 16, kernel_size=3, dimension=3, bias=True).to(device)
    with torch.no_grad():
        conv.kernel.copy_(k0)
        conv.bias.copy_(b0)
    gk, gb = [], []
    for rank in range(world):
        conv.zero_grad(set_to_none=True)
        coords, feats = _scene(rank)
        y = conv(ME.SparseTensor(feats.to(device), coords.to(device)))
        (y.F * y.F).sum().backward()
        gk.append(conv.kernel.grad.clone())
        gb.append(conv.bias.grad.clone())
    assert_close(gk0, ((gk[0] + gk[1]) / 2).cpu(), 1e-5, 1e-6, "averaged kernel gradient")
    assert_close(gb0, ((gb[0] + gb[1]) / 2).cpu(), 1e-5, 1e-6, "averaged bias gradient")


def _unet_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import distributed as D
    import minkunet
    from make_golden_minkunet_weights import seeded_parameters
    r, w, lr = D.init_from_env()
    dev = D.local_device(lr)
    net = minkunet.MinkUNet14(3, 5, D=3)
    seeded_parameters(net.named_parameters())
    net = net.to(dev).train()
    ddp = D.data_parallel(net, dev, sync_batchnorm=True)        # the reference example's recipe
    assert any(isinstance(m, ME.MinkowskiSyncBatchNorm) for m in ddp.modules())
    coords = minkunet.synthetic_scene(3000, grid=48, seed=3 + rank)
    g = torch.Generator().manual_seed(60 + rank)
    feats = torch.rand(coords.shape[0], 3, generator=g)
    lw = torch.rand(coords.shape[0], 5, generator=g) - 0.5
    y = ddp(ME.SparseTensor(feats.to(dev), coords.to(dev)))
    assert torch.equal(y.C.cpu(), coords)
    # DDP averages the per-rank gradients: scale so that the global objective is sum over both scenes
    (y.F * lw.to(dev)).sum().backward()
    torch.cuda.synchronize()
    params = dict(ddp.module.named_parameters())
    out[rank] = {k: params[k].grad.cpu() for k in ("conv0p1s1.kernel", "final.kernel", "block4.0.conv1.kernel",
                                                   "bn0.bn.weight", "convtr7p2s2.kernel")}
    out[f"y{rank}"] = y.F.detach().cpu()
    D.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_with_sync_batchnorm_matches_one_process_on_the_batched_scenes(device):
    """MinkUNet14 under DistributedDataParallel + MinkowskiSyncBatchNorm on two ranks (one scene each) == the same
    network in one process on ONE sparse tensor holding both scenes (batch indices 0 and 1): synchronised batch
    statistics are the statistics of the batched tensor, and DDP's gradient average is half the gradient of the
    summed objective."""
    import minkowskiengine_amd as ME
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import minkunet
    from make_golden_minkunet_weights import seeded_parameters
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_unet_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    net = minkunet.MinkUNet14(3, 5, D=3)
    seeded_parameters(net.named_parameters())
    net = net.to(device).train()
    cs, fs, ws = [], [], []
    for rank in range(world):
        c = minkunet.synthetic_scene(3000, grid=48, seed=3 + rank, batch_index=rank)
        g = torch.Generator().manual_seed(60 + rank)
        cs.append(c)
        fs.append(torch.rand(c.shape[0], 3, generator=g))
        ws.append(torch.rand(c.shape[0], 5, generator=g) - 0.5)
    y = net(ME.SparseTensor(torch.cat(fs).to(device), torch.cat(cs).to(device)))
    assert torch.equal(y.C.cpu(), torch.cat(cs))
    (y.F * torch.cat(ws).to(device)).sum().backward()
    n0 = cs[0].shape[0]
    scale = float(y.F.abs().max())
    assert_close(out["y0"], y.F[:n0].detach().cpu(), 2e-4 * scale, 1e-4, "rank 0 output")
    assert_close(out["y1"], y.F[n0:].detach().cpu(), 2e-4 * scale, 1e-4, "rank 1 output")
    # gradients: both ranks hold the same averaged tensors; against the one-process run they are compared in norm.
    # The two runs differ in arithmetic (torch's SyncBatchNorm kernels vs csrc/norm.hip), and the gradients of this
    # random-sign loss through 40 training-mode batch norms are ill-conditioned: the reference's OWN fp32 run
    # deviates from its float64 run by ~5 % of an element's size on this network (tests/golden/minkunet14_3k.npz,
    # noise/*), so a per-element bound would test the noise.  Relative L2 error <= 3 %, cosine >= 0.999.
    params = dict(net.named_parameters())
    for key, g0 in out[0].items():
        want = (params[key].grad / world).cpu().double()
        assert torch.equal(g0, out[1][key]), key
        got = g0.double()
        rel = float((got - want).norm() / want.norm())
        cos = float((got * want).sum() / (got.norm() * want.norm()))
        assert rel <= 0.03 and cos >= 0.999, (key, rel, cos)


def _arena_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import distributed as D
    import minkunet
    from make_golden_minkunet_weights import seeded_parameters
    r, w, lr = D.init_from_env()
    dev = D.local_device(lr)
    net = minkunet.MinkUNet14(3, 5, D=3)
    if rank == 0:
        seeded_parameters(net.named_parameters())      # rank 1 starts from torch's init: the arena broadcasts rank 0's
    net = net.to(dev).train()
    arena = D.GradientArena(net, chunks=4)
    coords = minkunet.synthetic_scene(3000, grid=48, seed=3 + rank)
    g = torch.Generator().manual_seed(60 + rank)
    feats = torch.rand(coords.shape[0], 3, generator=g)
    lw = torch.rand(coords.shape[0], 5, generator=g) - 0.5
    for _ in range(3):                                  # step 0 learns the arrival order, steps 1 - 2 overlap pieces
        arena.zero_grad()
        y = net(ME.SparseTensor(feats.to(dev), coords.to(dev)))
        (y.F * lw.to(dev)).sum().backward()
        arena.all_reduce()
    torch.cuda.synchronize()
    out[rank] = ({n: p.grad.cpu() for n, p in net.named_parameters()}, arena.describe())
    D.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_gradient_arena_on_two_ranks_averages_the_hip_gradients(device):
    """distributed.GradientArena (gradients born in one flat buffer by the HIP kernels, pieces all-reduced from inside the
    backward pass) on two ranks sharing the GPU over gloo == the average of the two scenes' gradients computed by one
    process (per-rank batch statistics: each scene on its own)"""
    import minkowskiengine_amd as ME
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import minkunet
    from make_golden_minkunet_weights import seeded_parameters
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_arena_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (g0, d0), (g1, d1) = out[0], out[1]
    assert d0["overlapped_pieces"] == 3 and d1["overlapped_pieces"] == 3, (d0, d1)
    net = minkunet.MinkUNet14(3, 5, D=3)
    seeded_parameters(net.named_parameters())
    net = net.to(device).train()
    acc = None
    for rank in range(world):
        net.zero_grad(set_to_none=True)
        coords = minkunet.synthetic_scene(3000, grid=48, seed=3 + rank)
        g = torch.Generator().manual_seed(60 + rank)
        feats = torch.rand(coords.shape[0], 3, generator=g)
        lw = torch.rand(coords.shape[0], 5, generator=g) - 0.5
        y = net(ME.SparseTensor(feats.to(device), coords.to(device)))
        (y.F * lw.to(device)).sum().backward()
        cur = {n: p.grad.double().cpu() for n, p in net.named_parameters()}
        acc = cur if acc is None else {n: acc[n] + cur[n] for n in cur}
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n                     # both ranks hold the same averaged gradient
        want = (acc[n] / 2).float()
        assert torch.allclose(g0[n], want, rtol=1e-4, atol=1e-5 * float(want.abs().max()) + 1e-7), n
