#!/usr/bin/env python
"""Which op behind ACVNet's feature maps is not run-to-run reproducible on the GPU?  (round 5, GPU call B)

`test_acvnet_train_grads_hand_written_path_isolated[hip]` found all 128 gradient tensors of `ACVNet.aggregate()` different between
two runs on identical inputs, where GwcNet_GC's 3-D path is bit-for-bit reproducible.  ACVNet adds: the `concatconv` head (stock
MIOpen 2-D convolutions), the windowed attention block (stock torch: rocBLAS / hipBLASLt GEMMs, softmax), the patch convolutions,
the attention-weighted volume.  Each is run twice here, forward + backward, and compared bitwise; then the whole aggregate() with
pieces replaced (attention on the CPU, oracle concat features).  Writes one JSON line per experiment.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_oracle as O  # noqa: E402
from stereo_toolbox_amd import ops  # noqa: E402
from stereo_toolbox_amd.losses import masked_smooth_l1_multi  # noqa: E402
from stereo_toolbox_amd.models import ACVNet  # noqa: E402
from stereo_toolbox_amd.models.ACVNet.submodule import attention_block  # noqa: E402
from stereo_toolbox_amd.models.features2d import run_head2d  # noqa: E402
from stereo_toolbox_amd.utils import fill_state_dict, synthetic_tensor  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "r5_acv_determinism.jsonl")
dev = torch.device("cuda:0")


def emit(**rec):
    line = json.dumps(rec)
    print(line, flush=True)
    if os.path.isdir(os.path.dirname(OUT)):
        with open(OUT, "a") as f:
            f.write(line + "\n")


def compare(name, fn, **extra):
    a, b = fn(), fn()
    diff = {k: float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-30)) for k in a if not torch.equal(a[k], b[k])}
    emit(experiment=name, tensors=len(a), not_bitwise_equal=len(diff),
         worst_rel=max(diff.values(), default=0.0), first=sorted(diff, key=diff.get, reverse=True)[:4], **extra)


def main():
    H, W, D, B = 64, 128, 64, 2
    m = ACVNet(D)
    sd = m.state_dict()
    fill_state_dict(sd)
    m.load_state_dict(sd)
    sd = {k: v.clone() for k, v in sd.items()}
    m = m.to(dev).train()
    left, right = synthetic_tensor((B, 3, H, W), 1), synthetic_tensor((B, 3, H, W), 2)
    gt = synthetic_tensor((B, H, W), 3, lo=0.0, hi=float(D - 2))
    with torch.no_grad():
        cxf = O.Ctx({k: v.clone() for k, v in sd.items()}, True)
        gl, gr = O.features_gwc(cxf, left, False)[0], O.features_gwc(cxf, right, False)[0]
        cl, cr = O.acv_concat_features(cxf, gl), O.acv_concat_features(cxf, gr)

    # 1. attention block alone, [2,4,4,8,128]
    ab = m.dres2.attention_block
    x0 = synthetic_tensor((B, 4, 4, 8, 128), 7).to(dev)
    g0 = synthetic_tensor((B, 4, 4, 8, 128), 8).to(dev)

    def run_attention():
        ab.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        y = ab(x)
        y.backward(g0)
        torch.cuda.synchronize()
        return {"y": y.detach(), "gx": x.grad, **{k: p.grad.clone() for k, p in ab.named_parameters()}}
    compare("attention_block fwd+bwd (stock torch on the GPU)", run_attention)
    for mode in ("0",):
        os.environ["ROCBLAS_DEFAULT_ATOMICS_MODE"] = mode
    torch.use_deterministic_algorithms(True, warn_only=True)
    compare("attention_block, torch.use_deterministic_algorithms(True)", run_attention)
    torch.use_deterministic_algorithms(False)

    # 2. concatconv alone (MIOpen)
    gld = gl.to(dev)

    def run_concatconv():
        m.concatconv.zero_grad(set_to_none=True)
        x = gld.clone().requires_grad_()
        y = run_head2d(m.concatconv, x)
        y.square().sum().backward()
        torch.cuda.synchronize()
        return {"y": y.detach(), "gx": x.grad, **{k: p.grad.clone() for k, p in m.concatconv.named_parameters()}}
    compare("concatconv fwd+bwd (MIOpen)", run_concatconv)

    # 3. the HIP-only ACV pieces: gwc volume -> patch convolutions, attention-weighted volume
    def run_patch():
        for p in (m.patch, m.patch_l1, m.patch_l2, m.patch_l3):
            p.zero_grad(set_to_none=True)
        a, b_ = gl.to(dev).requires_grad_(), gr.to(dev).requires_grad_()
        gwc = ops.cost_volume(a, b_, None, None, D // 4, 40)
        d1, d2 = m._dilations(gwc.device)
        v = ops.dwconv_hw(gwc, m.patch.weight.reshape(40, 9), d1)
        w2 = torch.cat((m.patch_l1.weight.reshape(8, 9), m.patch_l2.weight.reshape(16, 9), m.patch_l3.weight.reshape(16, 9)), 0)
        pv = ops.dwconv_hw(v, w2, d2)
        pv.square().sum().backward()
        torch.cuda.synchronize()
        return {"pv": pv.detach(), "ga": a.grad, "gb": b_.grad, "gpatch": m.patch.weight.grad.clone(),
                "gl2": m.patch_l2.weight.grad.clone()}
    compare("gwc volume + patch convolutions (HIP)", run_patch)

    def run_acvol():
        a, b_ = cl.to(dev).requires_grad_(), cr.to(dev).requires_grad_()
        att = synthetic_tensor((B, D // 4, H // 4, W // 4), 9).to(dev).requires_grad_()
        vol = ops.ac_volume(a, b_, torch.softmax(att, dim=1), D // 4)
        vol.square().sum().backward()
        torch.cuda.synchronize()
        return {"vol": vol.detach(), "ga": a.grad, "gb": b_.grad, "gatt": att.grad}
    compare("softmax(att) * concat volume (HIP + torch.softmax)", run_acvol)

    # 4. whole aggregate(), pieces replaced
    def run_aggregate(concat_from_oracle, attention_on_cpu):
        saved = attention_block.forward
        if attention_on_cpu:
            attention_block.forward = lambda self, x: _cpu_attention(self, x, saved)
        try:
            m.zero_grad(set_to_none=True)
            a, b_ = gl.to(dev).requires_grad_(), gr.to(dev).requires_grad_()
            kw = dict(concat_left=cl.to(dev), concat_right=cr.to(dev)) if concat_from_oracle else {}
            preds = m.aggregate(a, b_, H, W, **kw)
            masked_smooth_l1_multi(preds, gt.to(dev), D, (0.5, 0.5, 0.7, 1.0)).backward()
            torch.cuda.synchronize()
            out = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
            out["d_feature[0]"], out["d_feature[1]"] = a.grad, b_.grad
            return out
        finally:
            attention_block.forward = saved
            m.to(dev)
    for cc, ac in ((False, False), (True, False), (False, True), (True, True)):
        compare("aggregate()", lambda cc=cc, ac=ac: run_aggregate(cc, ac), concat_features_from_oracle=cc, attention_on_cpu=ac)
    torch.use_deterministic_algorithms(True, warn_only=True)
    compare("aggregate(), torch.use_deterministic_algorithms(True)", lambda: run_aggregate(True, False),
            concat_features_from_oracle=True, attention_on_cpu=False)


class _CpuAttn(torch.autograd.Function):
    """attention block evaluated on the CPU (deterministic oneDNN / MKL path) inside a GPU graph: test harness only."""

    @staticmethod
    def forward(ctx, x, wq, bq, wf, bf, blk, fwd):
        ctx.blk, ctx.fwd = blk, fwd
        ctx.save_for_backward(x, wq, bq, wf, bf)
        with torch.no_grad():
            return _cpu_eval(blk, fwd, x, wq, bq, wf, bf)[0].to(x.device)

    @staticmethod
    def backward(ctx, g):
        x, wq, bq, wf, bf = ctx.saved_tensors
        with torch.enable_grad():
            y, leaves = _cpu_eval(ctx.blk, ctx.fwd, x, wq, bq, wf, bf, grad=True)
            grads = torch.autograd.grad(y, leaves, g.cpu())
        return tuple(t.to(x.device) for t in grads) + (None, None)


def _cpu_eval(blk, fwd, x, wq, bq, wf, bf, grad=False):
    import copy
    c = copy.deepcopy(blk).cpu()
    leaves = [t.detach().cpu().requires_grad_(grad) for t in (x, wq, bq, wf, bf)]
    c.qkv_3d.weight, c.qkv_3d.bias = torch.nn.Parameter(leaves[1]), torch.nn.Parameter(leaves[2])
    c.final1x1.weight, c.final1x1.bias = torch.nn.Parameter(leaves[3]), torch.nn.Parameter(leaves[4])
    # (Parameters re-wrap the leaves; differentiate w.r.t. the Parameters)
    y = fwd(c, leaves[0])
    return y, [leaves[0], c.qkv_3d.weight, c.qkv_3d.bias, c.final1x1.weight, c.final1x1.bias]


def _cpu_attention(self, x, fwd):
    return _CpuAttn.apply(x, self.qkv_3d.weight, self.qkv_3d.bias, self.final1x1.weight, self.final1x1.bias, self, fwd)


if __name__ == "__main__":
    main()
