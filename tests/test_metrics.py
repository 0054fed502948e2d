"""DisparityMetrics (sync-free accumulators) vs a literal per-image restatement of the reference loops
(evaluation/sceneflow_test.py:26-47, evaluation/generalization_eval.py:29-58)."""
import numpy as np
import torch

from stereo_toolbox_amd.metrics import DisparityMetrics


def _reference_loops(preds, gts, nocs, maxdisp, thr):
    m = np.zeros(4)
    g = np.zeros(4)
    image_num = np.zeros(4)
    n = 0
    for pred, gt, noc in zip(preds, gts, nocs):
        n += 1
        mask = (gt > 0) * (gt < maxdisp - 1)
        valid = mask.sum().item()
        err = torch.abs(pred - gt)
        if valid > 0:
            e = err[mask]
            m[0] += e.mean().item()
            for k in (1, 2, 3):
                m[k] += (e > k).sum().item() / valid * 100
        nocm = noc.bool() * mask
        occm = ~nocm * mask
        if valid > 0:
            image_num[0] += 1
            g[0] += err[mask].mean().item()
            image_num[3] += 1
            g[3] += (err[mask] > thr).sum().item() / valid * 100
        if occm.sum().item() > 0:
            image_num[1] += 1
            g[1] += (err[occm] > thr).sum().item() / occm.sum().item() * 100
        if nocm.sum().item() > 0:
            image_num[2] += 1
            g[2] += (err[nocm] > thr).sum().item() / nocm.sum().item() * 100
    return m / n, g / image_num


def test_metrics_match_reference_loops():
    torch.manual_seed(0)
    H, W, maxdisp = 12, 20, 64
    preds, gts, nocs = [], [], []
    for i in range(7):
        gt = torch.rand(H, W) * 80 - 5                 # some invalid (<=0, >= maxdisp-1)
        if i == 3:
            gt = torch.zeros(H, W)                     # an image without valid pixels
        pred = gt + torch.randn(H, W) * 2
        noc = (torch.rand(H, W) > 0.3).float()
        if i == 5:
            noc = torch.ones(H, W)                     # no occluded pixels in this image
        preds.append(pred); gts.append(gt); nocs.append(noc)
    ref_sf, ref_gen = _reference_loops(preds, gts, nocs, maxdisp, 3.0)

    acc = DisparityMetrics(maxdisp)
    acc.update(torch.stack(preds[:3]), torch.stack(gts[:3]), torch.stack(nocs[:3]))          # batched
    for p, g, n in zip(preds[3:], gts[3:], nocs[3:]):
        acc.update(p[None, None], g[None, None], n[None, None])                               # [B,1,H,W] form
    out = acc.compute()
    assert abs(out["epe"] - ref_sf[0]) < 1e-5
    for k in range(3):
        assert abs(out["outliers"][k] - ref_sf[1 + k]) < 1e-4
    assert abs(out["epe_valid_images"] - ref_gen[0]) < 1e-5
    assert abs(out["occ"][2] - ref_gen[1]) < 1e-4
    assert abs(out["noc"][2] - ref_gen[2]) < 1e-4
    assert abs(out["all"][2] - ref_gen[3]) < 1e-4


def test_masked_loss_equals_boolean_indexing():
    """losses.masked_smooth_l1_multi == the reference-style `pred[mask]` formulation (oracle.smooth_l1_multi)."""
    from oracle import torch_oracle as O
    from stereo_toolbox_amd.losses import masked_smooth_l1_multi
    torch.manual_seed(2)
    gt = torch.rand(2, 9, 14) * 80 - 5
    preds = [gt + torch.randn(2, 9, 14) * s for s in (3.0, 2.0, 1.0, 0.5)]
    preds[1] = preds[1].unsqueeze(1)
    w = (0.5, 0.5, 0.7, 1.0)
    a = masked_smooth_l1_multi(preds, gt, 64, w)
    b = O.smooth_l1_multi(preds, gt, 64, w)
    assert abs(a.item() - b.item()) < 1e-5 * abs(b.item())
