"""Sample-sharded data parallelism: one process per GPU, every rank owns whole scenes and its own
coordinate manager / hash tables / kernel maps (no cross-GPU coordinate maps); the only exchange is
the gradient all-reduce once per step — RCCL over xGMI through torch.distributed backend "nccl"
(reference: examples/multigpu_ddp.py:72-131, PyTorch DDP over NCCL; no native collectives exist in
the reference).  Device-agnostic so the logic is testable with backend "gloo" on CPU."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (as torch.distributed.run sets them).  Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_scenes(n_scenes, rank, world):
    """Indices of the scenes rank `rank` owns (round-robin by sample, whole scenes only)."""
    return list(range(rank, n_scenes, world))


def broadcast_parameters(module, src=0):
    if world_size() == 1:
        return
    for p in module.parameters():
        dist.broadcast(p.data, src=src)
    for b in module.buffers():
        dist.broadcast(b.data, src=src)


def allreduce_gradients(module, average=True, bucket_bytes=25 * 1024 * 1024):
    """Sum (or average) all parameter gradients over the ranks in flat buckets (few, large
    collectives: xGMI rings are per-link bound)."""
    w = world_size()
    if w == 1:
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket]) if len(bucket) > 1 else bucket[0].reshape(-1)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(w)
        if len(bucket) > 1:
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()

    for g in grads:
        if size + g.numel() * g.element_size() > bucket_bytes and bucket:
            flush()
            bucket, size = [], 0
        bucket.append(g)
        size += g.numel() * g.element_size()
    flush()


def max_over_ranks(value, device=None):
    """Max of a python float over all ranks (used for the step time)."""
    if world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if world_size() > 1:
        dist.barrier()
