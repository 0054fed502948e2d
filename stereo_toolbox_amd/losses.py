"""Supervised multi-output disparity loss without host synchronisation.

The reference trainer indexes predictions with a boolean mask (`pred[mask]`, trainer/trainer_torchrun.py:272-284), which
launches a dynamic-shape `nonzero` and stalls the host once per prediction.  The same value is obtained with a
select / sum, so the whole train step (bench.py) stays asynchronous.  Excluded pixels are *selected away* (torch.where),
not multiplied by zero: the reference's datasets mark missing ground truth with NaN / inf, and NaN * 0 = NaN would
poison the sum where `pred[mask]` simply never sees those pixels.
"""
import torch
import torch.nn.functional as F

GWCNET_WEIGHTS = (0.5, 0.5, 0.7, 1.0)      # GwcNet paper (the reference ships no supervised loss for these models)
PSMNET_WEIGHTS = (0.5, 0.7, 1.0)           # PSMNet paper: [pred1, pred2, pred3] (stackhourglass.py:159)
ACVNET_WEIGHTS = (0.5, 0.5, 0.7, 1.0)      # [pred_attention, pred0, pred1, pred2] (acv.py:235)
ACVNET_FROZEN_WEIGHTS = (0.5, 0.7, 1.0)    # freeze_attn_weights=True: [pred0, pred1, pred2] (acv.py:233)
PCWNET_WEIGHTS = (0.5, 0.5, 0.5, 0.7, 1.0, 1.3)   # 6 train-mode outputs (pcwnet.py:466-end)


def masked_smooth_l1_multi(preds, gt, maxdisp, weights=GWCNET_WEIGHTS):
    """sum_i w_i * mean_{valid} smooth_l1(pred_i, gt), valid = (gt > 0) & (gt < maxdisp - 1)
    (mask of trainer_torchrun.py:272 / evaluation/sceneflow_test.py:29).  preds: list of [B,H,W] or [B,1,H,W];
    one weight per prediction (a length mismatch raises instead of silently dropping outputs)."""
    if len(preds) != len(weights):
        raise ValueError(f"masked_smooth_l1_multi: {len(preds)} predictions but {len(weights)} weights "
                         "(see the per-model *_WEIGHTS tuples of this module)")
    mask = (gt > 0) & (gt < maxdisp - 1)             # NaN compares False: non-finite gt is excluded here
    inv = 1.0 / mask.sum().clamp_min(1).to(gt.dtype)
    zero = torch.zeros((), dtype=gt.dtype, device=gt.device)
    loss = 0.0
    for p, w in zip(preds, weights):
        p = p.squeeze(1) if p.dim() == 4 else p
        gt_safe = torch.where(mask, gt, p.detach())  # excluded pixels: zero residual, zero gradient, never NaN
        loss = loss + w * torch.where(mask, F.smooth_l1_loss(p, gt_safe, reduction="none"), zero).sum() * inv
    return loss
