"""Data-parallel gradient exchange for the train step: one flat fp32 bucket, one collective.

Replaces the bucketed DDP all-reduce of reference trainer/trainer_torchrun.py:116-121 (triggered by
loss.backward(), :287/:294).  One process per GPU (torchrun), backend "nccl" = RCCL over xGMI on
ROCm, "gloo" on CPU.  The whole gradient of these models is 21-29 MB, i.e. a single message: every
parameter's .grad is a view into one flat buffer, so the step issues exactly one all-reduce and no
gradient copies.  The module stays a plain nn.Module and also works under torch DDP
(Trainer.prepare_model) -- FlatGradSync is the lean path used by bench.py.
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, model, process_group=None):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.group = process_group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        for p, v in zip(self.params, self.views):
            p.grad = v
        self.world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def zero_grad(self):
        """Zero the bucket in place (keeps the .grad views alive; do not call model.zero_grad(set_to_none=True)).
        backward() then accumulates straight into the bucket (one small add kernel per parameter)."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def detach_grads(self):
        """Alternative to zero_grad(): drop the .grad views so that backward() *assigns* fresh gradients
        (no per-parameter accumulate kernels); pack() then gathers them into the bucket with one
        fused multi-tensor copy."""
        for p in self.params:
            p.grad = None

    def pack(self):
        grads, views = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                grads.append(p.grad)
                views.append(v)
        if grads:
            torch._foreach_copy_(views, grads)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self, async_op=False):
        """Average gradients over ranks with a single collective; returns the work handle if async."""
        if self.world == 1:
            return None
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            return dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        w = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            class _Scaled:
                def __init__(s, work, flat, world):
                    s.work, s.flat, s.world = work, flat, world

                def wait(s):
                    s.work.wait()
                    s.flat.div_(s.world)
            return _Scaled(w, self.flat, self.world)
        self.flat.div_(self.world)
        return None

    def views_intact(self):
        """True while every p.grad still aliases the flat bucket (checked by tests)."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                return False
            off += p.numel()
        return True


def broadcast_parameters(model, src=0, group=None):
    """Rank-`src` parameters and buffers to all ranks (what the DDP constructor does once)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src, group=group)
