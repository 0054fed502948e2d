"""Data-parallel gradient exchange for the train step: one flat fp32 buffer, one or a few collectives.

Replaces the bucketed DDP all-reduce of reference trainer/trainer_torchrun.py:116-121 (triggered by
loss.backward(), :287/:294).  One process per GPU (torchrun), backend "nccl" = RCCL over xGMI on
ROCm, "gloo" on CPU.  The whole gradient of these models is 21-29 MB: every parameter's .grad is a
view into one flat buffer, so a step issues no gradient copies and

  * `buckets=1` (default): exactly one all-reduce after backward();
  * `buckets=K, overlap=True` (ONE backward() per step; a second one before finish() raises): the flat buffer is cut
    into K contiguous ranges (registration order);
    a range is all-reduced asynchronously from an autograd hook as soon as its last gradient has been
    accumulated, i.e. the aggregation / classifier ranges (registered last, differentiated first)
    travel over xGMI while the 2-D feature CNN is still in its backward pass.  `finish()` waits.

The module stays a plain nn.Module and also works under torch DDP (Trainer.prepare_model) --
FlatGradSync is the lean path used by bench.py.
"""
import torch
import torch.distributed as dist


class _Scaled:
    """Work handle of a SUM all-reduce that still has to be divided by the world size (gloo has no AVG)."""

    def __init__(self, work, flat, world):
        self.work, self.flat, self.world = work, flat, world

    def wait(self):
        self.work.wait()
        self.flat.div_(self.world)


class FlatGradSync:
    def __init__(self, model, process_group=None, buckets=1, overlap=False, collective_at_world_1=False):
        """`collective_at_world_1`: issue the collectives even in a 1-rank group (a world-size-1 RCCL communicator is a
        real communicator on a real stream) -- lets the hook -> async all-reduce -> finish() ordering be exercised on a
        single-GPU box; off by default (a lone rank has nothing to exchange)."""
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.group = process_group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views, self.offsets = [], []
        off = 0
        for p in self.params:
            seg = self.flat[off:off + p.numel()]
            # the view takes the parameter's own strides (the 2-D conv weights are stored channels_last): parameter, gradient
            # and optimizer state then agree in layout, which the multi-tensor optimizer kernels require -- one mismatching
            # tensor sends the whole foreach call down the one-kernel-per-tensor path
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            self.views.append(seg.as_strided(p.shape, p.stride()) if dense and not p.is_contiguous() else seg.view_as(p))
            self.offsets.append(off)
            off += p.numel()
        for p, v in zip(self.params, self.views):
            p.grad = v
        self.world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        # contiguous ranges of roughly equal size, cut at parameter boundaries
        buckets = max(1, min(int(buckets), len(self.params)))
        self.bucket_of, self.ranges = [], []
        target, start, b = n / buckets, 0, 0
        for i, p in enumerate(self.params):
            self.bucket_of.append(b)
            end = self.offsets[i] + p.numel()
            if (end >= target * (b + 1) and b < buckets - 1) or i == len(self.params) - 1:
                self.ranges.append((start, end))
                start, b = end, b + 1
        self.nb = len(self.ranges)
        self.exchange = self.world > 1 or (bool(collective_at_world_1) and dist.is_available() and dist.is_initialized())
        self.overlap = bool(overlap) and self.exchange
        self._pending = [0] * self.nb
        self._launched = [False] * self.nb
        self._works = []
        self._hooks = []
        if self.overlap:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    # ------------------------------------------------------------------ step protocol
    def zero_grad(self):
        """Zero the buffer in place (keeps the .grad views alive; do not call model.zero_grad(set_to_none=True)).
        backward() then accumulates straight into the buffer (one small add kernel per parameter).  Arms the
        per-range counters of the overlapped mode."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v
        self._arm()

    def detach_grads(self):
        """Alternative to zero_grad(): drop the .grad views so that backward() *assigns* fresh gradients
        (no per-parameter accumulate kernels); pack() then gathers them into the buffer with one
        fused multi-tensor copy.  (The overlapped mode copies each gradient into its view from the hook.)"""
        for p in self.params:
            p.grad = None
        self._arm()

    def _arm(self):
        self._works = []
        self._launched = [False] * self.nb
        self._pending = [0] * self.nb
        for b in self.bucket_of:
            self._pending[b] += 1

    def _make_hook(self, i):
        def hook(p):
            v = self.views[i]
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
            b = self.bucket_of[i]
            if self._launched[b] or self._pending[b] <= 0:
                # a second backward() before finish() (gradient accumulation, one backward per loss): the range is
                # already on the wire (or done) -- more local gradient added now would differ across ranks, silently
                raise RuntimeError("FlatGradSync(overlap=True) supports exactly one backward() per zero_grad() / "
                                   "detach_grads(): a gradient arrived for a range that was already all-reduced. Use "
                                   "overlap=False for gradient accumulation (finish() then reduces once, after the "
                                   "last backward).")
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi = self.ranges[b]
        self._launched[b] = True
        self._works.append(self._reduce(self.flat[lo:hi], True))

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

    def _reduce(self, t, async_op):
        if dist.get_backend(self.group) == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return _Scaled(w, t, self.world)
        t.div_(self.world)
        return None

    def all_reduce(self, async_op=False):
        """Average the whole buffer over ranks with a single collective; returns the work handle if async."""
        if not self.exchange:
            return None
        return self._reduce(self.flat, async_op)

    def finish(self):
        """End of backward(): non-overlapped mode = pack + one all-reduce; overlapped mode = launch the ranges
        whose hooks never fired (parameters without a gradient this step) and wait for all of them."""
        if not self.overlap:
            self.pack()
            self.all_reduce()
            return
        for p, v in zip(self.params, self.views):      # parameters the graph did not reach
            if p.grad is None:
                v.zero_()
                p.grad = v
        for b in range(self.nb):
            if not self._launched[b]:
                self._launch(b)
        for w in self._works:
            if w is not None:
                w.wait()
        self._works = []

    def views_intact(self):
        """True while every p.grad still aliases the flat buffer (checked by tests)."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                return False
        return True


def broadcast_parameters(model, src=0, group=None):
    """Rank-`src` parameters and buffers to all ranks (what the DDP constructor does once): the floating-point tensors
    travel packed in ONE flat buffer per dtype (like the gradient buffer above) instead of one collective per tensor
    (533 for GwcNet_GC); integer buffers (`num_batches_tracked`) in a second one."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    by_kind = {}
    for t in list(model.parameters()) + list(model.buffers()):
        by_kind.setdefault((t.dtype, t.device), []).append(t.data)
    for (_, _), ts in by_kind.items():
        if len(ts) == 1:
            dist.broadcast(ts[0], src, group=group)
            continue
        flat = torch.cat([t.reshape(-1) for t in ts])          # (reshape of a channels_last weight copies in logical order;
        dist.broadcast(flat, src, group=group)                 #  the copy_ below writes back through the same logical view)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view(t.shape))
            off += n
