"""Sample-sharded data parallelism: one process per GPU, every rank owns whole scenes and its own
coordinate manager / hash tables / kernel maps (no cross-GPU coordinate maps); the only exchange is
the gradient all-reduce — RCCL over xGMI through torch.distributed backend "nccl" — done by
torch's DistributedDataParallel: gradient buckets are reduced while the rest of the backward pass
still runs (reference: examples/multigpu_ddp.py:81-95: init_process_group -> DDP ->
MinkowskiSyncBatchNorm.convert_sync_batchnorm; the reference has no native collectives).

Device-agnostic so the logic is testable with backend "gloo": on CPU tensors (tests/test_distributed_cpu.py)
and with several ranks sharing ONE GPU (tests/test_gpu_distributed.py, bench.py --gpus N on a 1-GPU box) —
RCCL refuses two ranks on one device, gloo stages CUDA tensors through the host."""
import os

import torch
import torch.distributed as dist


def visible_gpus():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def pick_backend(world, n_gpus=None):
    """"nccl" (= RCCL on ROCm) when every rank gets its own GPU, else "gloo" (CPU, or ranks sharing a GPU)."""
    n_gpus = visible_gpus() if n_gpus is None else n_gpus
    return "nccl" if n_gpus >= world and n_gpus > 0 else "gloo"


_SINGLE_RANK_GROUP = [False]   # a one-rank process group was asked for by name: the collectives really run (see init_from_env)


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (as torch.distributed.run sets them).  Returns (rank, world_size, local_rank).
    With WORLD_SIZE = 1 a group is only created when `backend` is NAMED ("nccl" | "gloo"): a one-rank RCCL group on a
    one-GPU box — DistributedDataParallel's bucket hooks, MinkowskiSyncBatchNorm and the all-reduce calls then run
    through librccl exactly as on N GPUs, with nobody to talk to (tests/test_gpu_rccl.py, `bench.py --gpus 1 --backend
    nccl`); without a named backend a single rank stays free of torch.distributed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world == 1 and backend is not None and not dist.is_initialized():
        _SINGLE_RANK_GROUP[0] = True
    if (world > 1 or _SINGLE_RANK_GROUP[0]) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = pick_backend(world)
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % visible_gpus())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def local_device(local_rank):
    """cuda device of a rank; ranks wrap around when there are fewer GPUs than ranks (gloo test mode)"""
    return torch.device("cuda", local_rank % visible_gpus())


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def exchange_active():
    """True when collectives are to be issued: more than one rank, or a one-rank group created by name (init_from_env)"""
    return dist.is_initialized() and (dist.get_world_size() > 1 or _SINGLE_RANK_GROUP[0])


def backend_name():
    return dist.get_backend() if dist.is_initialized() else None


def shard_scenes(n_scenes, rank, world):
    """Indices of the scenes rank `rank` owns (round-robin by sample, whole scenes only)."""
    return list(range(rank, n_scenes, world))


def broadcast_parameters(module, src=0):
    if not exchange_active():
        return
    # (p.detach() shares p's version counter, `p.data` does not: the weight-image cache of the convolutions is keyed
    # by it, so a broadcast into `.data` would leave stale packed weights on the receiving ranks)
    with torch.no_grad():
        for p in module.parameters():
            dist.broadcast(p.detach(), src=src)
        for b in module.buffers():
            dist.broadcast(b.detach(), src=src)
    from .host import invalidate_packed_weights
    invalidate_packed_weights()


def data_parallel(module, device=None, bucket_cap_mb=25, sync_batchnorm=False):
    """Wrap `module` for sample-sharded training: torch DistributedDataParallel (parameters broadcast from rank 0,
    gradients averaged in `bucket_cap_mb` buckets overlapped with the backward pass; few, large collectives suit
    the per-link-bound xGMI rings).  `sync_batchnorm` converts MinkowskiBatchNorm layers first, as the reference's
    example does (examples/multigpu_ddp.py:95).  With one rank (and no group created by name) the module is returned
    unchanged."""
    if not exchange_active():
        return module
    if sync_batchnorm:
        from .layers import MinkowskiSyncBatchNorm
        module = MinkowskiSyncBatchNorm.convert_sync_batchnorm(module)
    from torch.nn.parallel import DistributedDataParallel
    if device is not None and device.type == "cuda":
        return DistributedDataParallel(module, device_ids=[device.index], output_device=device.index,
                                       bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
    return DistributedDataParallel(module, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)


def allreduce_gradients(module, average=True, bucket_bytes=25 * 1024 * 1024):
    """Explicit post-backward all-reduce of all parameter gradients in flat buckets, for loops that do not use
    `data_parallel` (no overlap with the backward pass — DDP is the training path).  Every rank issues the same
    collectives whatever its local state: ALL parameters that require a gradient take part in registration order
    (a missing gradient counts as zeros and is materialised), buckets never mix dtypes, and the reduced values are
    always copied back into `.grad` (also for a single-tensor bucket and for non-contiguous gradients)."""
    w = world_size()
    if not exchange_active():
        return
    params = [p for p in module.parameters() if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])      # always an explicit buffer (one dtype)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(w)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view(g.shape))
            off += g.numel()

    for p in params:
        g = p.grad
        nbytes = g.numel() * g.element_size()
        if bucket and (size + nbytes > bucket_bytes or g.dtype != bucket[0].dtype or g.device != bucket[0].device):
            flush()
            bucket, size = [], 0
        bucket.append(g)
        size += nbytes
    flush()


def _reduce_scalar(value, op, device):
    if not exchange_active():
        return float(value)
    if device is None and backend_name() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=op)
    return float(t.item())


def max_over_ranks(value, device=None):
    """Max of a python float over all ranks (used for the step time)."""
    return _reduce_scalar(value, dist.ReduceOp.MAX, device)


def sum_over_ranks(value, device=None):
    return _reduce_scalar(value, dist.ReduceOp.SUM, device)


def gather_over_ranks(value, device=None):
    """-> [value of rank 0, value of rank 1, ...] of a python float, on every rank"""
    w = world_size()
    if not exchange_active():
        return [float(value)]
    if device is None and backend_name() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(w)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def collective_info():
    """what the JSON line of bench.py records about the exchange layer: backend as torch.distributed reports it, the
    world size the process group reports, and the RCCL version when the backend is "nccl" (= RCCL on ROCm)"""
    info = {"backend": backend_name(), "world_size_reported": world_size()}
    if info["backend"] == "nccl":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            info["rccl_version"] = None
    return info


def no_sync(module):
    """context manager: backward passes inside it do NOT all-reduce (DistributedDataParallel.no_sync — gradient
    accumulation windows); a no-op for an unwrapped module (one rank)"""
    import contextlib
    fn = getattr(module, "no_sync", None)
    return fn() if callable(fn) and exchange_active() else contextlib.nullcontext()


def barrier():
    if exchange_active():
        dist.barrier()


def shutdown():
    """destroy the default process group (if any) and forget a one-rank group created by name"""
    if dist.is_initialized():
        dist.destroy_process_group()
    _SINGLE_RANK_GROUP[0] = False
