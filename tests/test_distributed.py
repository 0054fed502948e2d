"""N>1 path on CPU: world_size-2 gloo processes exercising the flat-bucket gradient exchange that
bench.py uses over RCCL (same code, backend "gloo" instead of "nccl")."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(7)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 4, 3, padding=1))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stereo_toolbox_amd.distributed import FlatGradSync, broadcast_parameters
    model = _net()
    if rank != 0:       # perturb, then the broadcast must restore rank 0's parameters
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    broadcast_parameters(model)
    sync = FlatGradSync(model)
    torch.manual_seed(100)
    data = torch.randn(4, 3, 6, 6)                 # global batch; each rank takes its shard
    shard = data[rank * 2:(rank + 1) * 2]
    for it in range(2):                            # two steps: the bucket views must survive
        if it == 0:
            sync.zero_grad()                       # accumulate-in-place flavour
            model(shard).square().mean().backward()
        else:
            sync.detach_grads()                    # assign-then-pack flavour (what bench.py uses)
            model(shard).square().mean().backward()
            sync.pack()
        assert sync.views_intact()
        w = sync.all_reduce(async_op=True)
        w.wait()
    q.put((rank, sync.flat.clone(), [p.detach().clone() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_sync_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # both ranks hold identical averaged gradients and identical (broadcast) parameters
    assert torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)
    # and the average equals the single-process gradient of the mean over the two shards
    model = _net()
    torch.manual_seed(100)
    data = torch.randn(4, 3, 6, 6)
    loss = 0.5 * (model(data[:2]).square().mean() + model(data[2:]).square().mean())
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(res[0][1], ref, rtol=1e-5, atol=1e-7)


def test_flat_grad_sync_single_process():
    sys.path.insert(0, ROOT)
    from stereo_toolbox_amd.distributed import FlatGradSync
    model = _net()
    sync = FlatGradSync(model)
    model(torch.randn(2, 3, 5, 5)).sum().backward()
    assert sync.views_intact() and sync.flat.abs().sum() > 0
    assert sync.all_reduce() is None                # world size 1: no collective
    sync.zero_grad()
    assert sync.flat.abs().sum() == 0 and sync.views_intact()
