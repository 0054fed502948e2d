"""On-device, sync-free accumulation of the toolbox's evaluation metrics.

The reference loops (`evaluation/sceneflow_test.py:26-47`, `evaluation/generalization_eval.py:29-58`,
`evaluation/drivingstereo_weather_test.py:25-49`) pull two to
seven scalars per image to the host with `.item()`; at GwcNet's inference rate on MI355X that serialises the GPU
behind the host.  Here every per-image statistic is accumulated in device tensors; `compute()` is the only host
synchronisation and `all_reduce()` merges the accumulators of a multi-GPU, batch-sharded evaluation (cfg5) with ONE
small collective.  The arithmetic follows the reference exactly:

  mask       = (gt > 0) & (gt < maxdisp - 1)
  EPE        = mean over images of mean_{mask} |pred - gt|
  k-px       = mean over images of 100 * #{mask & |pred - gt| > k} / #mask
  occ / noc  = the same outlier rate restricted to ~noc_mask / noc_mask (generalization_eval)

and keeps the reference's image counting: SceneFlow-style metrics divide by the number of images SEEN (an image
without valid pixels adds nothing but still counts, `sceneflow_test.py:31-33,47`), the occ/noc/all variants divide by
the number of images that had pixels of that kind (`generalization_eval.py:44-58`).
"""
import torch


class DisparityMetrics:
    def __init__(self, maxdisp=192, thresholds=(1.0, 2.0, 3.0), device=None):
        self.maxdisp = maxdisp
        self.thresholds = torch.tensor(thresholds, dtype=torch.float32, device=device)
        n = len(thresholds)
        dev = device
        self.seen = torch.zeros((), dtype=torch.float64, device=dev)             # images seen
        self.epe_sum = torch.zeros((), dtype=torch.float64, device=dev)          # sum of per-image EPE
        self.out_sum = torch.zeros(n, dtype=torch.float64, device=dev)           # sum of per-image outlier %
        self.n_all = torch.zeros((), dtype=torch.float64, device=dev)            # images with valid pixels
        self.occ_sum = torch.zeros(n, dtype=torch.float64, device=dev)
        self.noc_sum = torch.zeros(n, dtype=torch.float64, device=dev)
        self.n_occ = torch.zeros((), dtype=torch.float64, device=dev)
        self.n_noc = torch.zeros((), dtype=torch.float64, device=dev)

    @torch.no_grad()
    def update(self, pred, gt, noc_mask=None):
        """pred, gt: [B,H,W] or [B,1,H,W]; noc_mask (optional): same shape, non-zero where the pixel is non-occluded."""
        pred = pred.reshape(pred.shape[0], -1).float()
        gt = gt.reshape(gt.shape[0], -1).float()
        valid = (gt > 0) & (gt < self.maxdisp - 1)            # NaN / inf ground truth (missing pixels) is excluded here
        # select, never multiply: the reference indexes (`error_map[mask]`), so non-finite gt outside the mask must not
        # reach a sum (NaN * 0 = NaN)
        err = torch.where(valid, (pred - gt).abs(), torch.zeros_like(pred))
        thr = self.thresholds.to(err.device).view(1, -1, 1)

        def rates(mask):
            cnt = mask.sum(1).double()                                           # [B]
            has = cnt > 0
            bad = ((err.unsqueeze(1) > thr) & mask.unsqueeze(1)).sum(2).double()  # [B, n]
            pct = torch.where(has.unsqueeze(1), bad / cnt.clamp(min=1).unsqueeze(1) * 100.0, torch.zeros_like(bad))
            return cnt, has, pct

        cnt, has, pct = rates(valid)
        epe = torch.where(has, err.sum(1).double() / cnt.clamp(min=1), torch.zeros_like(cnt))
        self.seen += pred.shape[0]
        self.epe_sum += epe.sum()
        self.out_sum += pct.sum(0)
        self.n_all += has.sum()
        if noc_mask is not None:
            noc = (noc_mask.reshape(noc_mask.shape[0], -1) != 0) & valid
            occ = ~noc & valid
            _, has_o, pct_o = rates(occ)
            _, has_n, pct_n = rates(noc)
            self.occ_sum += pct_o.sum(0)
            self.noc_sum += pct_n.sum(0)
            self.n_occ += has_o.sum()
            self.n_noc += has_n.sum()

    def _state(self):
        return [self.seen, self.epe_sum, self.out_sum, self.n_all, self.occ_sum, self.noc_sum, self.n_occ, self.n_noc]

    def all_reduce(self, group=None):
        """Merge the accumulators of all ranks (one flat all-reduce)."""
        import torch.distributed as dist
        parts = self._state()
        flat = torch.cat([p.reshape(-1) for p in parts])
        dist.all_reduce(flat, group=group)
        o = 0
        for p in parts:
            n = p.numel()
            p.copy_(flat[o:o + n].reshape(p.shape))
            o += n

    def compute(self):
        """-> dict of python floats / lists (the only host synchronisation)."""
        seen = max(float(self.seen), 1.0)
        out = {"epe": float(self.epe_sum) / seen,                                       # sceneflow_test.py:47
               "outliers": [float(v) / seen for v in self.out_sum],
               "epe_valid_images": float(self.epe_sum) / max(float(self.n_all), 1.0),   # generalization_eval.py:58
               "all": [float(v) / max(float(self.n_all), 1.0) for v in self.out_sum],
               "occ": [float(v) / max(float(self.n_occ), 1.0) for v in self.occ_sum],
               "noc": [float(v) / max(float(self.n_noc), 1.0) for v in self.noc_sum]}
        return out


def drivingstereo_weather_table(per_split, thresholds=(3, 3, 3, 3)):
    """The 4 x 2 table of the reference's weather evaluation (`evaluation/drivingstereo_weather_test.py:13-66`): one row
    [EPE, outliers %] per split (sunny, cloudy, rainy, foggy), the outlier threshold of each row taken from `thresholds`
    (the reference's `threshlods` argument, default 3 px everywhere) and both columns averaged over the images that have
    valid pixels (:41-48).  per_split: one DisparityMetrics per split whose `thresholds` contain that split's value."""
    rows = []
    for acc, thr in zip(per_split, thresholds):
        out = acc.compute()
        k = [float(t) for t in acc.thresholds.tolist()].index(float(thr))
        rows.append([out["epe_valid_images"], out["all"][k]])
    return rows
