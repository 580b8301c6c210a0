"""The N>1 path on CPU: world_size-2 gloo processes exercise scene sharding, parameter broadcast, the explicit
bucketed gradient all-reduce (missing gradients, mixed dtypes, non-contiguous gradients, single-tensor buckets),
the DistributedDataParallel wrapper of `distributed.data_parallel` and max-over-ranks timing.  The compute
stand-in is a torch module; the HIP kernels under two ranks are covered by tests/test_gpu_distributed.py."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Mixed(torch.nn.Module):
    """parameters of two dtypes, one that never receives a gradient on rank 1, one with a transposed gradient"""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.zeros(5, 3))
        self.b = torch.nn.Parameter(torch.zeros(7, dtype=torch.float64))
        self.c = torch.nn.Parameter(torch.zeros(4, 6))
        self.d = torch.nn.Parameter(torch.zeros(2))


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from minkowskiengine_amd import distributed as D
    import minkowskiengine_amd as ME
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world and D.backend_name() == "gloo"
    torch.manual_seed(100 + rank)                      # different init per rank ...
    conv = ME.MinkowskiConvolution(4, 6, kernel_size=3, dimension=3, bias=True)
    D.broadcast_parameters(conv)                       # ... identical after the broadcast
    scenes = D.shard_scenes(5, rank, world)
    conv.kernel.grad = torch.full_like(conv.kernel, float(rank + 1))
    conv.bias.grad = torch.full_like(conv.bias, float(10 * (rank + 1)))
    D.allreduce_gradients(conv, average=True, bucket_bytes=1024)   # small buckets: multi-bucket path
    # the defects of the round-1 version: a gradient missing on one rank, dtypes mixed in one bucket, a
    # non-contiguous gradient alone in its bucket
    m = _Mixed()
    m.a.grad = torch.full((5, 3), float(rank + 1))
    m.b.grad = torch.full((7,), float(rank + 1), dtype=torch.float64)
    m.c.grad = torch.full((6, 4), float(rank + 1)).t()            # non-contiguous view
    assert not m.c.grad.is_contiguous()
    if rank == 0:
        m.d.grad = torch.full((2,), 4.0)                          # rank 1 has no gradient for d
    D.allreduce_gradients(m, average=True, bucket_bytes=64)       # one tensor per bucket
    mixed = (m.a.grad.mean().item(), m.b.grad.mean().item(), m.b.grad.dtype == torch.float64,
             m.c.grad.mean().item(), m.d.grad.mean().item())
    # DistributedDataParallel wrapper: averaged gradients, parameters broadcast from rank 0
    torch.manual_seed(7 + rank)
    lin = torch.nn.Linear(3, 2)
    ddp = D.data_parallel(lin)
    x = torch.full((4, 3), float(rank + 1))
    ddp(x).sum().backward()
    ddp_out = (lin.weight.detach().sum().item(), lin.weight.grad.mean().item())
    t = D.max_over_ranks(0.5 + rank)
    s = D.sum_over_ranks(len(scenes))
    # what bench.py --gpus N records about the exchange (round 4): per-rank values, the collective layer, and a
    # gradient-accumulation window under DDP.no_sync — the micro-step inside it leaves the gradients un-reduced
    gathered = D.gather_over_ranks(10.0 + rank)
    info = D.collective_info()
    lin.weight.grad = None
    with D.no_sync(ddp):
        ddp(x).sum().backward()
    local_grad = lin.weight.grad.mean().item()             # this rank's own gradient: 4 * (rank + 1)
    ddp(x).sum().backward()                                # the closing micro-step reduces the ACCUMULATED gradient
    window_grad = lin.weight.grad.mean().item()
    with D.no_sync(lin):                                   # an unwrapped module: a no-op context
        pass
    D.barrier()
    out[rank] = (conv.kernel.detach().sum().item(), conv.kernel.grad.mean().item(), conv.bias.grad.mean().item(),
                 t, s, scenes, mixed, ddp_out, gathered, info, local_grad, window_grad)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a[0] == b[0], "parameters differ after broadcast"
    assert a[1] == b[1] == 1.5 and a[2] == b[2] == 15.0, "gradient average wrong"
    assert a[3] == b[3] == 1.5, "max over ranks wrong"
    assert a[4] == b[4] == 5.0 and a[5] == [0, 2, 4] and b[5] == [1, 3]
    assert a[6] == b[6] == (1.5, 1.5, True, 1.5, 2.0), (a[6], b[6])
    assert a[7][0] == b[7][0], "DDP did not broadcast the parameters"
    assert a[7][1] == b[7][1] == pytest.approx(4 * 1.5), (a[7], b[7])   # d(sum)/dW = sum of inputs, averaged
    assert a[8] == b[8] == [10.0, 11.0]
    assert a[9] == b[9] == {"backend": "gloo", "world_size_reported": 2}
    assert a[10] == pytest.approx(4.0) and b[10] == pytest.approx(8.0), "no_sync reduced the gradient"
    assert a[11] == b[11] == pytest.approx(2 * 4 * 1.5), "the window's closing step must reduce the accumulated gradient"


def test_backend_choice():
    from minkowskiengine_amd import distributed as D
    assert D.pick_backend(8, n_gpus=8) == "nccl" and D.pick_backend(2, n_gpus=8) == "nccl"
    assert D.pick_backend(2, n_gpus=1) == "gloo"      # ranks sharing a GPU: RCCL refuses, gloo stages through host
    assert D.pick_backend(2, n_gpus=0) == "gloo"


def _arena_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from minkowskiengine_amd import distributed as D
    D.init_from_env(backend="gloo")
    torch.manual_seed(50 + rank)                       # different weights per rank: the arena broadcasts rank 0's
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    net[2].bias.requires_grad_(False)                  # a frozen parameter is not in the arena
    extra = torch.nn.Parameter(torch.zeros(4, dtype=torch.float64))      # a second dtype: its own flat buffer
    net.register_parameter("extra", extra)
    arena = D.GradientArena(net)
    w0 = net[0].weight.detach().clone()
    x = torch.full((4, 6), float(rank + 1))
    res = []
    for step in range(2):                              # the second step reuses the slices
        arena.zero_grad()
        loss = net(x).sum()
        if rank == 0 or step == 0:                     # step 1: rank 1 has NO gradient for `extra`
            loss = loss + (extra * float(rank + 1)).sum()
        loss.backward()
        arena.all_reduce()
        res.append((net[0].weight.grad.clone(), net[2].weight.grad.clone(),
                    None if extra.grad is None else extra.grad.clone(), net[2].bias.grad))
    desc = arena.describe()
    out[rank] = (w0, res, desc)
    D.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradient_arena_averages_like_ddp_on_two_gloo_ranks():
    """distributed.GradientArena on CPU tensors (no producing kernels here: every gradient takes the copy-in path): flat
    buffers per dtype, parameters broadcast from rank 0, gradients averaged by ONE all-reduce per buffer, `p.grad`
    re-pointed at its slice, a parameter without a gradient on one rank, a frozen parameter left alone"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_arena_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (w0a, ra, da), (w0b, rb, db) = out[0], out[1]
    assert torch.equal(w0a, w0b), "parameters were not broadcast"
    # one process, both inputs, same weights: the averaged gradient
    torch.manual_seed(50)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    assert torch.equal(net[0].weight.detach(), w0a)
    grads = []
    for r in range(world):
        net.zero_grad(set_to_none=True)
        net(torch.full((4, 6), float(r + 1))).sum().backward()
        grads.append((net[0].weight.grad.clone(), net[2].weight.grad.clone()))
    for step in range(2):
        for res in (ra, rb):
            assert torch.allclose(res[step][0], (grads[0][0] + grads[1][0]) / 2)
            assert torch.allclose(res[step][1], (grads[0][1] + grads[1][1]) / 2)
            assert res[step][3] is None                                  # frozen
    # `extra`: gradients 1 (rank 0) and 2 (rank 1) in step 0 -> 1.5 on both; step 1: only rank 0 produces one (1.0), rank
    # 1's missing gradient counts as zeros and is materialised: 0.5 on BOTH ranks (every rank applies the same update)
    assert torch.allclose(ra[0][2], torch.full((4,), 1.5, dtype=torch.float64)) and torch.allclose(rb[0][2], ra[0][2])
    assert torch.allclose(ra[1][2], torch.full((4,), 0.5, dtype=torch.float64)) and torch.allclose(rb[1][2], ra[1][2])
    assert set(da["buffers"]) == {"torch.float32@cpu", "torch.float64@cpu"} and da["copied_in"] >= 3 and da["born_in_place"] == 0


def _arena_overlap_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from minkowskiengine_amd import distributed as D
    D.init_from_env(backend="gloo")
    torch.manual_seed(9)
    layers = [torch.nn.Linear(8, 8) for _ in range(12)]
    net = torch.nn.Sequential(*layers)
    arena = D.GradientArena(net, chunks=4)
    x = torch.full((3, 8), 0.1 * (rank + 1))
    res = []
    for step in range(5):
        arena.zero_grad()
        if step < 3:
            net(x).sum().backward()
        elif step == 3:
            net[:6](x).sum().backward()                # layers 6..11 get no gradient: their pieces cannot be launched early
        else:
            net(x).sum().backward()                    # ... and the full network again
        arena.all_reduce()
        res.append(([None if p.grad is None else p.grad.clone() for p in net.parameters()], arena.describe()))
    out[rank] = res
    D.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradient_arena_overlapped_pieces_on_two_gloo_ranks():
    """chunks = 4 on a 24-parameter network: step 0 learns the arrival order (and still averages correctly), steps 1 - 2
    all-reduce three pieces from inside the backward pass and the rest at the end, step 3 runs half the network (the
    pieces of the unused half fall back to the closing all-reduce, missing gradients count as zeros), step 4 is the full
    network again — every step's gradients equal the two-rank average"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_arena_overlap_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    torch.manual_seed(9)
    net = torch.nn.Sequential(*[torch.nn.Linear(8, 8) for _ in range(12)])

    def expected(half):
        gs = []
        for r in range(world):
            net.zero_grad(set_to_none=True)
            (net[:6] if half else net)(torch.full((3, 8), 0.1 * (r + 1))).sum().backward()
            gs.append([torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in net.parameters()])
        return [(a + b) / 2 for a, b in zip(*gs)]
    full, half = expected(False), expected(True)
    for rank in range(world):
        for step, (grads, desc) in enumerate(out[rank]):
            want = half if step == 3 else full
            for g, w_ in zip(grads, want):
                assert g is not None and torch.allclose(g, w_, rtol=1e-5, atol=1e-7), (rank, step)
            if step in (1, 2, 4):
                assert desc["overlapped_pieces"] == 3, (step, desc)
            if step == 0:
                assert desc["overlapped_pieces"] == 0
            if step == 3:
                assert desc["overlapped_pieces"] < 3, desc


def _arena_accum_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from minkowskiengine_amd import distributed as D
    D.init_from_env(backend="gloo")
    torch.manual_seed(11)
    net = torch.nn.Sequential(*[torch.nn.Linear(8, 8) for _ in range(12)])
    arena = D.GradientArena(net, chunks=4)
    xs = [torch.full((3, 8), 0.1 * (rank + 1)), torch.full((3, 8), -0.05 * (rank + 2))]
    res = []
    for step in range(4):
        arena.zero_grad()
        with arena.no_sync():                          # every backward pass of the window but the last
            net(xs[0]).sum().backward()
        net(xs[1]).sum().backward()
        arena.all_reduce()
        res.append(([p.grad.clone() for p in net.parameters()], arena.describe()))
    # the same window WITHOUT no_sync: the second pass reaches pieces that are already reduced -> loud, not wrong
    arena.zero_grad()
    net(xs[0]).sum().backward()
    raised = False
    try:
        net(xs[1]).sum().backward()
    except RuntimeError as e:
        raised = "no_sync" in str(e)
    out[rank] = (res, raised)
    D.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradient_arena_accumulation_window_with_overlapped_pieces():
    """chunks > 1 with TWO backward passes per step.  Inside `arena.no_sync()` nothing is sent, the closing
    pass sends the pieces: every step's gradients equal the two-rank average of the summed passes on both ranks; the
    same window without no_sync raises instead of mixing a reduced sum with a local gradient."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_arena_accum_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    torch.manual_seed(11)
    net = torch.nn.Sequential(*[torch.nn.Linear(8, 8) for _ in range(12)])
    gs = []
    for r in range(world):
        net.zero_grad(set_to_none=True)
        net(torch.full((3, 8), 0.1 * (r + 1))).sum().backward()
        net(torch.full((3, 8), -0.05 * (r + 2))).sum().backward()
        gs.append([p.grad.clone() for p in net.parameters()])
    want = [(a + b) / 2 for a, b in zip(*gs)]
    for rank in range(world):
        res, raised = out[rank]
        assert raised, "a second backward pass on reduced pieces must raise"
        for step, (grads, desc) in enumerate(res):
            for g, w_ in zip(grads, want):
                assert torch.allclose(g, w_, rtol=1e-5, atol=1e-7), (rank, step)
            if step >= 1:
                assert desc["overlapped_pieces"] == 3, (step, desc)
    for a, b in zip(out[0][0][-1][0], out[1][0][-1][0]):
        assert torch.equal(a, b), "ranks disagree"


def test_gradient_arena_low_precision_parameters_and_two_arenas():
    """bf16 parameters get a bf16 flat buffer (p.grad must have the parameter's dtype); two arenas in
    one process arm only their own destinations (zero_grad of one leaves the other's live gradients alone) and release
    their host-layer entries when closed."""
    from minkowskiengine_amd import distributed as D
    from minkowskiengine_amd import backend as MEB
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 2)).bfloat16()
    arena = D.GradientArena(net, broadcast=False)
    arena.zero_grad()
    net(torch.ones(3, 4, dtype=torch.bfloat16)).sum().backward()
    arena.all_reduce()                                                    # (no process group: copy-in + re-point only)
    assert all(p.grad is not None and p.grad.dtype == torch.bfloat16 for p in net.parameters())
    assert set(arena.describe()["buffers"]) == {"torch.bfloat16@cpu"}
    # per-arena arming, on the Python host's table (CPU tensors are never registered: use fake entries)
    saved = dict(MEB._GRAD_DEST)
    try:
        MEB._GRAD_DEST.clear()
        MEB._GRAD_DEST[1000] = ["a", False]
        MEB._GRAD_DEST[2000] = ["b", False]
        MEB.arm_grad_destinations([1000])
        assert MEB._GRAD_DEST[1000][1] is True and MEB._GRAD_DEST[2000][1] is False
        MEB.drop_grad_destinations([1000])
        assert 1000 not in MEB._GRAD_DEST and 2000 in MEB._GRAD_DEST
        MEB.arm_grad_destinations()
        assert MEB._GRAD_DEST[2000][1] is True
    finally:
        MEB._GRAD_DEST.clear()
        MEB._GRAD_DEST.update(saved)
    arena.close()
    arena.close()                                                         # idempotent
