"""N>1 path on CPU: world_size-2 gloo processes exercising the flat-bucket gradient exchange that
bench.py uses over RCCL (same code, backend "gloo" instead of "nccl")."""
import os
import socket
import sys

import pytest
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


def _worker(rank, world, port, q, overlap=False):
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
    sync = FlatGradSync(model, buckets=2, overlap=True) if overlap else FlatGradSync(model)
    torch.manual_seed(100)
    data = torch.randn(4, 3, 6, 6)                 # global batch; each rank takes its shard
    shard = data[rank * 2:(rank + 1) * 2]
    for it in range(2):                            # two steps: the bucket views must survive
        if overlap:
            # ranges of the flat buffer are all-reduced from autograd hooks while backward() is still running;
            # both step flavours (accumulate into the views / assign fresh gradients) end with finish()
            assert sync.nb == 2 and sync.overlap
            (sync.zero_grad if it == 0 else sync.detach_grads)()
            model(shard).square().mean().backward()
            sync.finish()
            assert sync.views_intact() and all(sync._launched)
            continue
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
    # numpy arrays are pickled by value (torch tensors travel as shared-memory handles that die with the worker)
    q.put((rank, sync.flat.detach().numpy().copy(), [p.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_flat_grad_sync_two_ranks_gloo(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [(r[0], torch.from_numpy(r[1]), [torch.from_numpy(a) for a in r[2]]) for r in res]
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


def _cl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stereo_toolbox_amd.distributed import FlatGradSync, broadcast_parameters
    model = _net()
    for m in model:
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    if rank != 0:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    broadcast_parameters(model)                    # one flat collective; must restore rank 0's values in logical order
    sync = FlatGradSync(model, buckets=2, overlap=True)
    torch.manual_seed(100)
    data = torch.randn(4, 3, 6, 6)
    sync.detach_grads()
    model(data[rank * 2:(rank + 1) * 2]).square().mean().backward()
    sync.finish()
    q.put((rank, torch.cat([v.reshape(-1) for v in sync.views]).detach().numpy().copy(),
           [p.detach().contiguous().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_sync_channels_last_two_ranks_gloo():
    """CPU twin of the RCCL value test for the layout path: with channels_last conv weights the flat buffer holds each
    gradient in MEMORY (NHWC) order; values are compared per parameter in logical order against the single-process mean
    gradient, and the one-collective parameter broadcast must reproduce rank 0's weights."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _net()
    for got, p in zip(res[1][2], model.parameters()):
        assert torch.equal(torch.from_numpy(got), p.detach())
    torch.manual_seed(100)
    data = torch.randn(4, 3, 6, 6)
    (0.5 * (model(data[:2]).square().mean() + model(data[2:]).square().mean())).backward()
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    for r in range(2):
        assert torch.allclose(torch.from_numpy(res[r][1]), want, rtol=1e-5, atol=1e-7)
    assert (res[0][1] == res[1][1]).all()


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


# ------------------------------------------------------------------------------ SyncBatchNorm
def _syncbn_block():
    torch.manual_seed(3)
    conv = nn.Conv3d(32, 32, 3, padding=1, bias=False)
    bn = nn.BatchNorm3d(32)
    with torch.no_grad():
        conv.weight.mul_(0.5)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.2, 0.2)
    return nn.Sequential(conv, bn)


def _syncbn_data():
    torch.manual_seed(5)
    return torch.randn(2, 32, 3, 4, 20) + 0.3, torch.randn(2, 32, 3, 4, 20)      # input (global batch 2), grad of output


def _syncbn_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_util import emu_product_path
    from stereo_toolbox_amd import aggregation
    seq = nn.SyncBatchNorm.convert_sync_batchnorm(_syncbn_block())
    assert isinstance(seq[1], nn.SyncBatchNorm)
    seq.train()
    x, gy = _syncbn_data()
    xs = x[rank:rank + 1].permute(0, 2, 3, 4, 1).contiguous().requires_grad_()     # NDHWC shard
    with emu_product_path():
        y = aggregation.convbn_block(xs, seq, relu=True)
        y.backward(gy[rank:rank + 1].permute(0, 2, 3, 4, 1).contiguous())
    # numpy arrays are pickled by value (torch tensors travel as shared-memory handles that die with the worker)
    q.put((rank,) + tuple(t.detach().numpy().copy() for t in (y, xs.grad, seq[0].weight.grad, seq[1].weight.grad,
                                                               seq[1].bias.grad, seq[1].running_mean, seq[1].running_var)))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_gloo():
    """convbn_3d + ReLU in train mode under SyncBatchNorm (reference trainer_torchrun.py:112-113): two gloo ranks with one
    sample each, run through the emulated product kernels, must reproduce stock BatchNorm over the global batch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emu_util import emu_lib
    emu_lib()                                      # build the emulator library once, before the ranks race for it
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [(r[0],) + tuple(torch.from_numpy(a) for a in r[1:]) for r in res]
    seq = _syncbn_block().train()
    x, gy = _syncbn_data()
    xr = x.clone().requires_grad_()
    yr = torch.relu(seq(xr))
    yr.backward(gy)

    def close(a, b, tol=2e-4):
        assert (a - b).abs().max().item() <= tol * (1 + b.abs().max().item()), (a - b).abs().max().item()
    for r in range(2):
        close(res[r][1].permute(0, 4, 1, 2, 3), yr[r:r + 1].detach())
        close(res[r][2].permute(0, 4, 1, 2, 3), xr.grad[r:r + 1])
        close(res[r][6], seq[1].running_mean)
        close(res[r][7], seq[1].running_var)
    # parameter gradients are local sums; their total over the ranks is the global-batch gradient (DDP then averages)
    close(res[0][3] + res[1][3], seq[0].weight.grad)
    close(res[0][4] + res[1][4], seq[1].weight.grad)
    close(res[0][5] + res[1][5], seq[1].bias.grad)


# ------------------------------------------------------------------------------ sharded evaluation metrics
def _metrics_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from stereo_toolbox_amd.metrics import DisparityMetrics
    torch.manual_seed(11)
    gt = torch.rand(6, 8, 10) * 70
    pred = gt + torch.randn(6, 8, 10) * 2
    acc = DisparityMetrics(64)
    acc.update(pred[rank::world], gt[rank::world])           # batch-sharded evaluation
    acc.all_reduce()
    q.put((rank, acc.compute()))
    dist.barrier()
    dist.destroy_process_group()


def test_metrics_all_reduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metrics_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from stereo_toolbox_amd.metrics import DisparityMetrics
    torch.manual_seed(11)
    gt = torch.rand(6, 8, 10) * 70
    pred = gt + torch.randn(6, 8, 10) * 2
    one = DisparityMetrics(64)
    one.update(pred, gt)
    want = one.compute()
    for r in range(2):
        assert abs(res[r][1]["epe"] - want["epe"]) < 1e-9
        assert all(abs(a - b) < 1e-9 for a, b in zip(res[r][1]["outliers"], want["outliers"]))


# ------------------------------------------------------------------------------ RCCL: gradient VALUES of the real model
def _rccl_model_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from oracle import torch_oracle as O
    from stereo_toolbox_amd.distributed import FlatGradSync, broadcast_parameters
    from stereo_toolbox_amd.models import GwcNet_GC
    from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor
    D, H, W = 64, 64, 128
    m = GwcNet_GC(D)
    sd = m.state_dict()
    fill_state_dict(sd)
    m.load_state_dict(sd)
    m = m.to(dev).train()
    broadcast_parameters(m)
    sync = FlatGradSync(m, buckets=2, overlap=True)
    left, right = synthetic_tensor((world, 3, H, W), 1), synthetic_tensor((world, 3, H, W), 2)
    gt = synthetic_tensor((world, H, W), 3, lo=0.0, hi=float(D - 2))
    sync.detach_grads()
    preds = m(left[rank:rank + 1].to(dev), right[rank:rank + 1].to(dev))
    O.smooth_l1_multi(preds, gt[rank:rank + 1].to(dev), D, (0.5, 0.5, 0.7, 1.0)).backward()
    sync.finish()
    torch.cuda.synchronize()
    # logical (NCHW) element order per parameter: the flat buffer itself holds channels_last parameters in memory order
    q.put((rank, torch.cat([v.reshape(-1) for v in sync.views]).detach().cpu().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_model_gradients_two_ranks_rccl():
    """What the 8-GPU scaling run relies on, checked by VALUE on a node with >= 2 GPUs: two RCCL ranks with one 64x128 pair
    each (per-replica BatchNorm statistics, the reference's default: trainer_torchrun.py:112-121) end up with the average
    of the two single-sample gradients in every rank's flat buffer -- overlapped 2-range all-reduce from autograd hooks."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 ROCm devices (the 1-GPU test box runs the gloo twins of this test)")
    sys.path.insert(0, ROOT)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    flat = [torch.from_numpy(r[1]) for r in res]
    assert torch.equal(flat[0], flat[1])
    from oracle import torch_oracle as O
    from stereo_toolbox_amd.models import GwcNet_GC
    from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor
    D, H, W = 64, 64, 128
    left, right = synthetic_tensor((2, 3, H, W), 1), synthetic_tensor((2, 3, H, W), 2)
    gt = synthetic_tensor((2, H, W), 3, lo=0.0, hi=float(D - 2))
    want = 0
    for s in range(2):                               # the same two samples, one after the other, on one device
        m = GwcNet_GC(D)
        sd = m.state_dict()
        fill_state_dict(sd)
        m.load_state_dict(sd)
        m = m.cuda().train()
        preds = m(left[s:s + 1].cuda(), right[s:s + 1].cuda())
        O.smooth_l1_multi(preds, gt[s:s + 1].cuda(), D, (0.5, 0.5, 0.7, 1.0)).backward()
        want = want + torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.requires_grad]).cpu() / 2
    err = (flat[0] - want).abs().max().item()
    assert err <= 1e-5 * want.abs().max().item() + 1e-7, err     # same kernels, same order: only the AVG's rounding differs


def test_flat_grad_sync_keeps_parameter_layout():
    """Gradient views of channels_last parameters (the 2-D conv weights of the models) carry the parameter's strides, so
    that parameter / gradient / optimizer state agree in layout (multi-tensor optimizer fast path) and still alias the
    flat buffer."""
    sys.path.insert(0, ROOT)
    from stereo_toolbox_amd.distributed import FlatGradSync
    model = _net()
    model[0].weight.data = model[0].weight.data.contiguous(memory_format=torch.channels_last)
    sync = FlatGradSync(model)
    assert model[0].weight.grad.stride() == model[0].weight.stride() != model[0].weight.contiguous().stride()
    sync.detach_grads()
    model(torch.randn(2, 3, 5, 5)).sum().backward()
    want = [p.grad.clone() for p in model.parameters()]
    sync.finish()
    assert sync.views_intact()
    for p, w in zip(model.parameters(), want):
        assert torch.equal(p.grad, w) and p.grad.stride() == p.stride()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt.step()                                       # runs on the views
    assert sync.views_intact()
