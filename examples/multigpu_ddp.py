"""Multi-GPU training of MinkUNet34C with DistributedDataParallel on the MI355X engine — the counterpart of the
reference's examples/multigpu_ddp.py:72-131 (init_process_group -> model -> DistributedDataParallel ->
MinkowskiSyncBatchNorm.convert_sync_batchnorm -> SGD loop), BASELINE configs[3]: batch = 8 scenes sharded across
8 x MI355X, gradient all-reduce over RCCL / xGMI.

One process per GPU; every rank owns whole scenes, its own coordinate manager, hash tables and kernel maps — no
coordinate data crosses GPUs.  The only exchange is DDP's bucketed gradient all-reduce (overlapped with the
backward pass) plus, with --sync-bn, the per-layer statistics of MinkowskiSyncBatchNorm.

    python examples/multigpu_ddp.py --max_ngpu 8                 # spawns one rank per visible GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/multigpu_ddp.py

Data: synthetic plane-union scenes (the reference downloads 1.ply, there is no network here), a new
scene per rank and iteration, voxelised on the GPU by ME.utils.sparse_quantize / batched by sparse_collate.
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import minkowskiengine_amd as ME  # noqa: E402
from minkowskiengine_amd import distributed as D  # noqa: E402
from minkunet import MinkUNet34C, synthetic_scene  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument("--batch_size", type=int, default=1, help="scenes per GPU and iteration")
parser.add_argument("--max_ngpu", type=int, default=8)
parser.add_argument("--points", type=int, default=200000, help="voxels per scene")
parser.add_argument("--iterations", type=int, default=10)
parser.add_argument("--dtype", choices=("f32", "bf16"), default="bf16")
parser.add_argument("--sync-bn", action="store_true", help="MinkowskiSyncBatchNorm, as the reference's example")
parser.add_argument("--oversubscribe", type=int, default=0,
                    help="run this many ranks although fewer GPUs are visible (gloo; functional test only)")
parser.add_argument("--exchange", choices=("arena", "ddp"), default="arena",
                    help="gradient exchange: distributed.GradientArena (one all-reduce of a flat buffer the gradients are "
                         "born in) or torch DistributedDataParallel (the reference's recipe)")


def load_scene(points, seed, device):
    """a synthetic scan: float points jittered around the plane-union voxels, colours as features"""
    vox = synthetic_scene(points, seed=seed)[:, 1:].float()
    g = torch.Generator().manual_seed(seed)
    pts = (vox + torch.rand(vox.shape, generator=g)) * 0.05               # metres, 5 cm voxels
    colors = torch.rand(vox.shape[0], 3, generator=g)
    coords, feats = ME.utils.sparse_quantize(pts.to(device), colors.to(device), quantization_size=0.05)
    labels = torch.zeros(feats.shape[0], dtype=torch.long, device=device)
    return coords, feats, labels


def main_worker(rank, world, args, port=None):
    if port is not None:                                  # spawned by main(); torch.distributed.run sets these itself
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port))
    rank, world, local_rank = D.init_from_env()           # backend "nccl" = RCCL (gloo when ranks share a GPU)
    device = D.local_device(local_rank)
    torch.cuda.set_device(device)
    torch.manual_seed(0)
    model = MinkUNet34C(3, 20, D=3).to(device)
    arena = None
    if args.exchange == "ddp":
        net = D.data_parallel(model, device, sync_batchnorm=args.sync_bn)     # DDP (+ SyncBN): the reference's recipe
    else:
        # gradients born in ONE flat buffer, one RCCL all-reduce per step (distributed.GradientArena): no per-parameter
        # hooks or bucket copies on this launch-bound step
        net = ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(model) if (args.sync_bn and world > 1) else model
        arena = D.GradientArena(net)
    criterion = nn.CrossEntropyLoss()
    optimizer = torch.optim.SGD(net.parameters(), lr=1e-1)
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    min_time = float("inf")
    for iteration in range(args.iterations):
        if arena is not None:
            arena.zero_grad()
        else:
            optimizer.zero_grad()
        batch = [load_scene(args.points, seed=1000 * iteration + rank * args.batch_size + b, device=device)
                 for b in range(args.batch_size)]
        coords_, feats_, labels_ = zip(*batch)
        coordinates, features = ME.utils.sparse_collate(coords_, feats_, device=device)
        inputs = ME.SparseTensor(features.to(tdt), coordinates, device=device)
        labels = torch.cat(labels_)
        torch.cuda.synchronize()
        st = time.perf_counter()
        outputs = net(inputs)
        loss = criterion(outputs.F.float(), labels)
        loss.backward()
        if arena is not None:
            arena.all_reduce()
        optimizer.step()
        torch.cuda.synchronize()
        t = D.max_over_ranks(time.perf_counter() - st, device if D.backend_name() == "nccl" else None)
        min_time = min(min_time, t)
        if rank == 0:
            n = inputs.F.shape[0]
            print(f"Iteration: {iteration}, Loss: {loss.item():.4f}, Time: {t * 1e3:.1f} ms, Min time: "
                  f"{min_time * 1e3:.1f} ms, voxels/rank: {n}, ranks: {world}", flush=True)
    D.shutdown()


def main():
    args = parser.parse_args()
    if "WORLD_SIZE" in os.environ:                        # launched by torch.distributed.run
        return main_worker(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), args)
    num_devices = args.oversubscribe or min(args.max_ngpu, torch.cuda.device_count())
    print("Testing", num_devices, "GPUs. Total batch size:", num_devices * args.batch_size)
    if num_devices <= 1:
        return main_worker(0, 1, args)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(main_worker, nprocs=num_devices, args=(num_devices, args, port))


if __name__ == "__main__":
    main()
