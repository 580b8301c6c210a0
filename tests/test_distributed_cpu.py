"""The N>1 path on CPU: world_size-2 gloo processes exercise scene sharding, parameter broadcast,
bucketed gradient all-reduce and max-over-ranks timing (the compute stand-in is a torch module; the
HIP path needs a GPU)."""
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


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from minkowskiengine_amd import distributed as D
    import minkowskiengine_amd as ME
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world
    torch.manual_seed(100 + rank)                      # different init per rank ...
    conv = ME.MinkowskiConvolution(4, 6, kernel_size=3, dimension=3, bias=True)
    D.broadcast_parameters(conv)                       # ... identical after the broadcast
    scenes = D.shard_scenes(5, rank, world)
    conv.kernel.grad = torch.full_like(conv.kernel, float(rank + 1))
    conv.bias.grad = torch.full_like(conv.bias, float(10 * (rank + 1)))
    D.allreduce_gradients(conv, average=True, bucket_bytes=1024)   # small buckets: multi-bucket path
    t = D.max_over_ranks(0.5 + rank)
    s = D.sum_over_ranks(len(scenes))
    D.barrier()
    out[rank] = (conv.kernel.detach().sum().item(), conv.kernel.grad.mean().item(), conv.bias.grad.mean().item(),
                 t, s, scenes)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
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
