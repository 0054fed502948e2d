"""Supervised multi-output disparity loss without host synchronisation.

The reference trainer indexes predictions with a boolean mask (`pred[mask]`, trainer/trainer_torchrun.py:272-284), which
launches a dynamic-shape `nonzero` and stalls the host once per prediction.  The same value is obtained with a
multiply / sum, so the whole train step (bench.py) stays asynchronous.
"""
import torch
import torch.nn.functional as F

GWCNET_WEIGHTS = (0.5, 0.5, 0.7, 1.0)      # GwcNet paper (the reference ships no supervised loss for these models)


def masked_smooth_l1_multi(preds, gt, maxdisp, weights=GWCNET_WEIGHTS):
    """sum_i w_i * mean_{valid} smooth_l1(pred_i, gt), valid = (gt > 0) & (gt < maxdisp - 1)
    (mask of trainer_torchrun.py:272 / evaluation/sceneflow_test.py:29).  preds: list of [B,H,W] or [B,1,H,W]."""
    mask = ((gt > 0) & (gt < maxdisp - 1)).to(gt.dtype)
    inv = 1.0 / mask.sum().clamp_min(1.0)
    loss = 0.0
    for p, w in zip(preds, weights):
        p = p.squeeze(1) if p.dim() == 4 else p
        loss = loss + w * (F.smooth_l1_loss(p, gt, reduction="none") * mask).sum() * inv
    return loss
