"""First contact with RCCL on the one GPU a test box has: a ONE-rank process group with backend "nccl"
(= RCCL on ROCm) — librccl is loaded and initialised, DistributedDataParallel's bucket hooks fire on the native
torch::autograd::Functions of the convolution / batch norm (side streams included), its all-reduces go through RCCL, and
the gradients equal those of the unwrapped module (an average over one rank).  What N > 1 adds — other ranks — is
covered by tests/test_gpu_distributed.py (2 ranks over gloo on the same GPU) and tests/test_distributed_cpu.py.
Reference recipe: examples/multigpu_ddp.py:81-95 (init_process_group -> DDP -> convert_sync_batchnorm)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from helpers import make_cloud

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _mapped_libraries():
    with open("/proc/self/maps") as f:
        return sorted({line.split()[-1] for line in f if ".so" in line})


def _worker(rank, port, host, out):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      ME_AMD_HOST=host)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import distributed as D
    import minkunet
    from make_golden_minkunet_weights import seeded_parameters
    assert ME.get_host() == host
    r, w, lr = D.init_from_env(backend="nccl")          # a one-rank group, because the backend is named
    assert (r, w) == (0, 1) and D.backend_name() == "nccl" and D.exchange_active()
    info = D.collective_info()
    assert info["backend"] == "nccl" and info["rccl_version"], info
    dev = D.local_device(lr)
    res = {"rccl_version": info["rccl_version"]}

    # 1. one convolution (fp32 and bf16): DDP-wrapped == unwrapped, bit for bit (the average over one rank)
    coords = make_cloud(4000, 16, 3, seed=7)
    g = torch.Generator().manual_seed(1)
    feats = torch.rand(coords.shape[0], 32, generator=g)
    for dtype in (torch.float32, torch.bfloat16):
        torch.manual_seed(11)
        conv = ME.MinkowskiConvolution(32, 64, kernel_size=3, dimension=3, bias=True).to(dev)
        grads = []
        for wrap in (False, True):
            net = D.data_parallel(conv, dev) if wrap else conv
            assert (type(net).__name__ == "DistributedDataParallel") == wrap
            conv.zero_grad(set_to_none=True)
            x = ME.SparseTensor(feats.to(dev).to(dtype), coords.to(dev))
            y = net(x)
            (y.F.float() * y.F.float()).sum().backward()
            torch.cuda.synchronize()
            grads.append((conv.kernel.grad.clone(), conv.bias.grad.clone()))
        assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1]), dtype
    D.allreduce_gradients(conv)                          # the explicit bucket path: sum over one rank, averaged
    torch.cuda.synchronize()
    assert torch.equal(conv.kernel.grad, grads[1][0])

    # 2. MinkUNet14 with MinkowskiSyncBatchNorm under DDP (the reference example's recipe) == plain MinkUNet14
    scene = minkunet.synthetic_scene(3000, grid=48, seed=3)
    f3 = torch.rand(scene.shape[0], 3, generator=g)
    lw = torch.rand(scene.shape[0], 5, generator=g) - 0.5
    outs = []
    for wrap in (False, True):
        net = minkunet.MinkUNet14(3, 5, D=3)
        seeded_parameters(net.named_parameters())
        net = net.to(dev).train()
        run = D.data_parallel(net, dev, sync_batchnorm=True) if wrap else net
        if wrap:
            assert any(isinstance(m, ME.MinkowskiSyncBatchNorm) for m in run.modules())
        y = run(ME.SparseTensor(f3.to(dev), scene.to(dev)))
        (y.F * lw.to(dev)).sum().backward()
        torch.cuda.synchronize()
        params = dict((run.module if wrap else run).named_parameters())
        outs.append({k: params[k].grad.detach().cpu() for k in ("conv0p1s1.kernel", "final.kernel",
                                                                "block4.0.conv1.kernel", "bn0.bn.weight")})
        outs[-1]["y"] = y.F.detach().cpu()
    for k in outs[0]:
        a, b = outs[0][k].double(), outs[1][k].double()
        # (torch's SyncBatchNorm on one rank computes the statistics with torch kernels, the plain module with
        # csrc/norm.hip: same formula, another summation order)
        err = float((a - b).abs().max() / max(1e-6, float(b.abs().max())))
        assert err < 2e-3, (k, err)
    # 3. gradients born in the all-reduce buffer (distributed.GradientArena): the same gradients as the plain module, bit
    # for bit (an average over one rank), every convolution / batch-norm gradient written in place by its kernel
    nets = []
    for use_arena in (False, True):
        net = minkunet.MinkUNet14(3, 5, D=3)
        seeded_parameters(net.named_parameters())
        net = net.to(dev).train()
        arena = D.GradientArena(net, chunks=4) if use_arena else None
        for _ in range(3):                                   # step 0 learns the arrival order, steps 1 - 2 overlap
            if arena is not None:
                arena.zero_grad()
            else:
                net.zero_grad(set_to_none=True)
            y = net(ME.SparseTensor(f3.to(dev).to(torch.bfloat16), scene.to(dev)))
            (y.F.float() * lw.to(dev)).sum().backward()
            if arena is not None:
                arena.all_reduce()
        torch.cuda.synchronize()
        nets.append({n: p.grad.detach().clone() for n, p in net.named_parameters()})
        if arena is not None:
            d = arena.describe()
            n_params = len(list(net.parameters()))
            # all but the head's bias (a torch broadcast-add gradient) and the two kernels whose channels are padded for
            # the tile kernels (3 -> 8 input channels of the stem, 5 -> 16 classes of the head: torch slices the gradient)
            assert d["born_in_place"] >= n_params - 4 and d["born_in_place"] + d["copied_in"] == n_params, d
            assert d["overlapped_pieces"] == 3, d            # three pieces all-reduced through RCCL inside the backward pass
            res["arena"] = d
            arena.close()
    for k in nets[0]:
        assert torch.equal(nets[0][k], nets[1][k]), k
    res["libs"] = [l for l in _mapped_libraries() if "rccl" in l or "me_amd" in l or "_me_host" in l]
    out["res"] = res
    D.barrier()
    D.shutdown()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("host", ["native", "python"])
def test_one_rank_rccl_group_runs_ddp_over_the_hip_kernels(device, host):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), host, out), nprocs=1, join=True)
    res = out["res"]
    assert res["rccl_version"]
    assert any("rccl" in l for l in res["libs"]), res["libs"]            # librccl.so is mapped into the process
    assert any("libme_amd" in l for l in res["libs"]), res["libs"]
    if host == "native":
        assert any("_me_host" in l for l in res["libs"]), res["libs"]


@pytest.mark.timeout(300)
def test_one_rank_rccl_group_in_this_process(device, monkeypatch):
    """the same first contact inside the pytest process (so that librccl shows up in ITS maps): group, one all-reduce
    through RCCL, a DDP-wrapped HIP convolution, group destroyed again"""
    import minkowskiengine_amd as ME
    from minkowskiengine_amd import distributed as D
    if torch.distributed.is_initialized():
        pytest.skip("a process group already exists in this process")
    for k, v in dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())).items():
        monkeypatch.setenv(k, v)
    try:
        D.init_from_env(backend="nccl")
        assert D.exchange_active() and D.collective_info()["rccl_version"]
        t = torch.arange(8, dtype=torch.float32, device=device)
        torch.distributed.all_reduce(t)
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
        assert D.max_over_ranks(3.5, device) == 3.5 and D.gather_over_ranks(2.0, device) == [2.0]
        conv = ME.MinkowskiConvolution(16, 32, kernel_size=3, dimension=3).to(device)
        net = D.data_parallel(conv, device)
        assert type(net).__name__ == "DistributedDataParallel"
        coords = make_cloud(2000, 12, 3, seed=3)
        x = ME.SparseTensor(torch.rand(coords.shape[0], 16).to(device), coords.to(device))
        net(x).F.sum().backward()
        g_ddp = conv.kernel.grad.clone()
        conv.zero_grad(set_to_none=True)
        conv(x).F.sum().backward()
        assert torch.equal(g_ddp, conv.kernel.grad)
        assert any("rccl" in l for l in _mapped_libraries())
    finally:
        D.shutdown()
    assert not D.exchange_active()
