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
            from . import _lib
            _lib.preload_device(local_rank % visible_gpus())   # this rank's device, now that it is selected (not at import)
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


class GradientArena:
    """Gradients that are BORN in their all-reduce buffer (MI355X-first alternative to torch DDP for this engine).

    DistributedDataParallel hooks every parameter, copies every gradient into a bucket view and synchronises its
    communication stream with the backward pass bucket by bucket: on a step that is already bound by its ~900 kernel
    launches that machinery costs 1.7 ms of a 12.1 ms MinkUNet34C step on ONE rank, where the all-reduce itself is free
    (bench.py --gpus 1 --backend nccl: multi_gpu.allreduce_ms.exposed_in_step; scripts/ddp_overhead.py).  Here every
    parameter owns a slice of ONE flat fp32 buffer per device; the host layers hand that slice to the kernels that produce
    the gradient (weight gradient of a convolution, weight / bias gradient of a batch norm: host.set_grad_destination), so
    autograd receives an alias of the slice and keeps it as `p.grad` without a copy; gradients produced by other operators
    are copied in by one multi-tensor launch.  The exchange is ONE RCCL all-reduce per buffer after the backward pass — 151 MB
    for MinkUNet34C: ~1 ms on an 8-GPU xGMI ring, the size few-large-collectives rings like — and the average is folded into
    the same buffer.  Same arithmetic as DDP's: sum over ranks in RCCL's ring order, one division.

        arena = D.GradientArena(model)          # after model.to(device); parameters are broadcast from rank 0
        for x, y in data:
            arena.zero_grad()                    # p.grad = None; the slices are armed for ONE backward pass
            loss(model(x), y).backward()
            arena.all_reduce()                   # no-op without a process group
            optimizer.step()

    Gradient accumulation (several backward passes between zero_grad and all_reduce): with chunks = 1 nothing is sent
    before all_reduce(), so a window needs no marking.  With chunks > 1 every backward pass but the LAST of a window must
    run inside `with arena.no_sync():` (DistributedDataParallel.no_sync's contract) — a piece that has been sent cannot
    take further local gradients; a backward pass that reaches such a piece raises instead of corrupting it.
    Parameters are keyed by their own dtype: bf16 / fp16 parameters get their own flat buffer and are reduced in it.
    """

    def __init__(self, module, average=True, broadcast=True, chunks=1):
        """chunks > 1: the flat buffer is cut into `chunks` pieces of about equal bytes IN THE ORDER THE GRADIENTS ARRIVE
        (learned during the first backward pass), and a piece is all-reduced asynchronously — on RCCL's stream, beside the
        rest of the backward pass — as soon as its last gradient has arrived (one hook per piece, not per parameter; a
        piece whose gradients are not all there when its hook fires is reduced at the end instead, so a changed
        execution order costs overlap, never correctness).  chunks = 1 (default): one all-reduce after the backward pass.
        Measured on ONE rank (scripts/ddp_overhead.py, MinkUNet34C bf16, 11.37 ms plain): chunks = 1 +0.00 ms, chunks = 4
        +0.49 ms (RCCL's kernels beside the backward pass) — the overlap has to hide more than that of a real exchange
        (~0.9 ms for 151 MB on an 8-GPU ring) to pay; it is an option, not the default."""
        self.average = average
        self.chunks = max(1, int(chunks))
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._order = list(range(len(self.params)))     # layout order of the slices: registration order until learned
        self._learned = self.chunks == 1
        self._learn_hooks, self._arrival = [], []
        self._sent_hooks, self._pieces, self._work = [], [], []
        self._born = self._copied = self._overlapped = 0
        self._defer = 0                                  # depth of no_sync(): overlapped pieces are not sent inside it
        self._ptrs = []                                  # parameter addresses whose destinations this arena registered
        self._layout()
        # the host layers keep the slices alive through their destination tables: release them with the arena
        import weakref
        self._finalizer = weakref.finalize(self, GradientArena._release, self._ptrs)
        if broadcast:
            broadcast_parameters(module)

    @staticmethod
    def _release(ptrs):
        from . import host
        host.drop_grad_destinations(list(ptrs))
        del ptrs[:]

    @staticmethod
    def _key(p):
        # (a gradient has its parameter's dtype: a bf16 parameter cannot take an fp32 slice as p.grad — ADVICE r5)
        return (p.device, p.dtype)

    def _layout(self):
        """(re)allocate the flat buffers with the slices in self._order and hand the slices to the host layers"""
        from . import host
        sizes = {}
        for i in self._order:
            p = self.params[i]
            sizes[self._key(p)] = sizes.get(self._key(p), 0) + (p.numel() + 63) // 64 * 64   # 256-byte aligned slices
        self._flat = {k: torch.zeros(n, dtype=k[1], device=k[0]) for k, n in sizes.items()}
        self._views, self._span = {}, {}
        offs = {k: 0 for k in sizes}
        for i in self._order:
            p = self.params[i]
            k = self._key(p)
            self._views[id(p)] = self._flat[k][offs[k]:offs[k] + p.numel()].view(p.shape)
            self._span[id(p)] = (k, offs[k], offs[k] + (p.numel() + 63) // 64 * 64)
            offs[k] += (p.numel() + 63) // 64 * 64
            if p.dtype == torch.float32 and p.is_cuda:
                host.set_grad_destination(p, self._views[id(p)])           # the producing kernels write here
                if p.data_ptr() not in self._ptrs:
                    self._ptrs.append(p.data_ptr())

    # ---- learning the arrival order (first backward pass) -----------------------------------------------------------------
    def _start_learning(self):
        self._arrival = []
        for i, p in enumerate(self.params):
            self._learn_hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._arrival.append(i)))

    def _finish_learning(self):
        for h in self._learn_hooks:
            h.remove()
        self._learn_hooks = []
        seen = set(self._arrival)
        order = [i for i in dict.fromkeys(self._arrival)] + [i for i in range(len(self.params)) if i not in seen]
        self._order = order
        self._learned = True
        # the slices move: gradients of THIS step are carried over (p.grad keeps its tensor until the next zero_grad)
        old = {id(p): p.grad for p in self.params}
        self._layout()
        src, dst = [], []
        for p in self.params:
            g = old[id(p)]
            if g is not None:
                src.append(g.detach())
                dst.append(self._views[id(p)])
                p.grad = self._views[id(p)]
        if src:
            torch._foreach_copy_(dst, src)
        # pieces of about equal bytes along the arrival order, per buffer; sentinel = the LAST parameter of a piece
        self._pieces = []
        for k, flat in self._flat.items():
            idx = [i for i in order if self._key(self.params[i]) == k and i in seen]
            if len(idx) < 2 * self.chunks:
                continue
            total = sum(self._span[id(self.params[i])][2] - self._span[id(self.params[i])][1] for i in idx)
            target, acc, cur = total / self.chunks, 0, []
            for i in idx:
                cur.append(i)
                acc += self._span[id(self.params[i])][2] - self._span[id(self.params[i])][1]
                if acc >= target * (len([q for q in self._pieces if q["key"] == k]) + 1) and len([q for q in self._pieces if q["key"] == k]) < self.chunks - 1:
                    self._pieces.append({"key": k, "params": cur, "lo": self._span[id(self.params[cur[0]])][1],
                                         "hi": self._span[id(self.params[cur[-1]])][2]})
                    cur = []
            # (the last piece of a buffer — the gradients that arrive at the very end — is reduced by all_reduce())
        for piece in self._pieces:
            sentinel = self.params[piece["params"][-1]]
            self._sent_hooks.append(sentinel.register_post_accumulate_grad_hook(lambda _p, piece=piece: self._launch(piece)))

    def _settle(self, indices):
        """make p.grad of these parameters their slices (copy-in for gradients born elsewhere) -> False if one is missing"""
        src, dst = [], []
        for i in indices:
            p = self.params[i]
            g, v = p.grad, self._views[id(p)]
            if g is None:
                return False
            if g.data_ptr() != v.data_ptr() or g.dtype != v.dtype:
                src.append(g.detach())
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
            self._copied += len(src)
        for i in indices:
            p = self.params[i]
            if p.grad.data_ptr() != self._views[id(p)].data_ptr():
                p.grad = self._views[id(p)]
        return True

    def no_sync(self):
        """Context manager for every backward pass of an accumulation window but the last (chunks > 1): the overlapped
        pieces are not sent inside it, gradients only accumulate in their slices."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            self._defer += 1
            try:
                yield self
            finally:
                self._defer -= 1
        return ctx()

    def _launch(self, piece):
        if not exchange_active():
            return
        if piece.get("done"):
            # the piece has been all-reduced and autograd has just ADDED this rank's next gradient to the sum: the slice
            # can no longer be repaired (sum over ranks of pass 1 + local pass 2).  Loud, not wrong.
            raise RuntimeError(
                "GradientArena(chunks > 1): a backward pass reached gradients that were already all-reduced in this "
                "window; run every backward pass of an accumulation window except the last inside "
                "`with arena.no_sync():` (or use chunks=1)")
        if self._defer > 0:
            return                                   # (inside no_sync(): the closing backward pass or all_reduce() sends it)
        if not self._settle(piece["params"]):
            return                                   # (the order changed: this piece is reduced at the end)
        flat = self._flat[piece["key"]]
        self._work.append(dist.all_reduce(flat[piece["lo"]:piece["hi"]], op=dist.ReduceOp.SUM, async_op=True))
        piece["done"] = True
        self._overlapped += 1

    # ---- the step -----------------------------------------------------------------------------------------------------------
    def zero_grad(self):
        """p.grad = None for every parameter and the slices are armed: the NEXT backward pass writes each gradient into
        its slice; further backward passes before the next zero_grad (an accumulation window) produce ordinary tensors
        that autograd adds to p.grad — the slice — in place."""
        from . import host
        for p in self.params:
            p.grad = None
        for piece in self._pieces:
            piece["done"] = False
        self._born = self._copied = self._overlapped = 0
        if not self._learned and not self._learn_hooks and exchange_active():
            self._start_learning()
        host.arm_grad_destinations(self._ptrs)      # this arena's slices only: another arena's may hold live gradients

    def all_reduce(self):
        """Finish the exchange: gradients that were not born in the arena are copied in (one multi-tensor launch) and
        re-pointed at their slice; what the overlapped pieces have not covered is all-reduced now; everything is averaged.
        With a process group a parameter WITHOUT a gradient on this rank counts as zeros and receives the average of the
        others (as allreduce_gradients: every rank must apply the same update); without one it keeps `grad = None`."""
        active = exchange_active()
        if self._learn_hooks:
            self._finish_learning()                 # (this step: everything is reduced below, in one piece per buffer)
        done_spans = {}
        for piece in self._pieces:
            if piece.get("done"):
                done_spans.setdefault(piece["key"], []).append((piece["lo"], piece["hi"]))
        covered = set()
        for piece in self._pieces:
            if piece.get("done"):
                covered.update(piece["params"])
        src, dst, missing = [], [], []
        born = 0
        for i, p in enumerate(self.params):
            g, v = p.grad, self._views[id(p)]
            if i in covered:
                born += 1
                continue
            if g is None:
                if active:
                    missing.append(v)
                    p.grad = v
                continue
            if g.data_ptr() == v.data_ptr() and g.dtype == v.dtype:
                born += 1
                continue
            src.append(g.detach())
            dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)
        if missing:
            torch._foreach_zero_(missing)
        self._born, self._copied = born - self._copied, self._copied + len(src)
        if not active:
            return
        w = world_size()
        for k, flat in self._flat.items():
            spans = sorted(done_spans.get(k, []))
            lo = 0
            for a, b in spans + [(flat.numel(), flat.numel())]:     # the gaps between the pieces already reduced
                if a > lo:
                    dist.all_reduce(flat[lo:a], op=dist.ReduceOp.SUM)
                lo = max(lo, b)
        for wk in self._work:
            wk.wait()
        self._work = []
        if self.average and w > 1:
            for flat in self._flat.values():
                flat.div_(w)

    def describe(self):
        return {"buffers": {f"{k[1]}@{k[0]}": int(v.numel() * v.element_size()) for k, v in self._flat.items()},
                "born_in_place": self._born, "copied_in": self._copied, "pieces": len(self._pieces) + len(self._flat),
                "overlapped_pieces": self._overlapped}

    def close(self):
        from . import host
        for h in self._learn_hooks + self._sent_hooks:
            h.remove()
        self._learn_hooks, self._sent_hooks = [], []
        self._finalizer()                            # (idempotent: drops this arena's destinations on both hosts)


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
