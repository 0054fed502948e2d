"""Golden fixture for the evaluation metrics: runs the REFERENCE's own loops
(`evaluation/sceneflow_test.py:13-59`, `evaluation/generalization_eval.py:13-83`,
`evaluation/drivingstereo_weather_test.py:13-66` with per-split thresholds) on synthetic predictions.

  python tests/golden/make_golden_metrics.py        (build container only: needs /root/reference)

The two reference files import `stereo_toolbox.datasets` (opencv, file-system datasets) and `tqdm`.  Their metric
arithmetic is what is pinned here, so the loops are fed through their own DataLoader from tiny in-memory datasets: a
package object named `stereo_toolbox.datasets` holding synthetic Dataset classes is placed in `sys.modules` before the
reference files are executed from where they lie (`importlib`), and the "model" handed to them returns a stored
prediction for the image index encoded in the left view.  Inputs are regenerated from seeds by the tests
(`synthetic_eval_set` below is the single definition, imported by tests/test_metrics.py); only the reference's results
are stored: tests/golden/metrics.npz.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from stereo_toolbox_amd.utils import synthetic_tensor  # noqa: E402

MAXDISP = 64
H, W = 12, 20


def synthetic_eval_set(n, seed):
    """n images: gt in [-5, 75) (so that both mask bounds bite), prediction = gt + heavy-tailed error, a random
    non-occlusion mask; special images: #1 has no valid pixel at all, #2 has no occluded pixel, #3 carries NaN / inf in
    the ground truth OUTSIDE the valid range test (missing pixels, as the reference's datasets mark them)."""
    gts, preds, nocs = [], [], []
    for i in range(n):
        gt = synthetic_tensor((H, W), seed + i, stream=0, lo=-5.0, hi=75.0)
        e = synthetic_tensor((H, W), seed + i, stream=1, lo=-1.0, hi=1.0)
        pred = gt + 4.0 * e * e * e + 0.3 * e
        noc = (synthetic_tensor((H, W), seed + i, stream=2, lo=0.0, hi=1.0) > 0.3).float()
        if i == 1:
            gt = torch.full((H, W), -1.0)
        if i == 2:
            noc = torch.ones(H, W)
        if i == 3:
            gt = gt.clone()
            gt[0, :5] = float("nan")
            gt[1, :5] = float("inf")
            gt[2, :5] = float("-inf")
        gts.append(gt)
        preds.append(pred)
        nocs.append(noc)
    return torch.stack(preds), torch.stack(gts), torch.stack(nocs)


class _SetBase(torch.utils.data.Dataset):
    N, SEED = 6, 100

    def __init__(self, split=None, training=False):
        self.pred, self.gt, self.noc = synthetic_eval_set(self.N, self.SEED)

    def __len__(self):
        return self.N

    def __getitem__(self, i):
        left = torch.full((3, H, W), float(self.SEED * 1000 + i))        # carries (set, index) to the stub model
        return {"left": left, "right": left.clone(), "gt_disp": self.gt[i], "noc_mask": self.noc[i]}


def _make(name, n, seed):
    return type(name, (_SetBase,), {"N": n, "SEED": seed})


SceneFlow_Dataset = _make("SceneFlow_Dataset", 7, 100)
KITTI2015_Dataset = _make("KITTI2015_Dataset", 5, 200)
KITTI2012_Dataset = _make("KITTI2012_Dataset", 4, 300)
MiddleburyEval3_Dataset = _make("MiddleburyEval3_Dataset", 4, 400)
ETH3D_Dataset = _make("ETH3D_Dataset", 5, 500)
WEATHER_SPLITS = {"test_half_sunny": (5, 600), "test_half_cloudy": (4, 700), "test_half_rainy": (6, 800), "test_half_foggy": (4, 900)}
WEATHER_THRESHOLDS = [3, 2, 1, 3]        # per-split outlier thresholds handed to the reference loop (its default is 3 everywhere)


class DrivingStereo_Dataset(_SetBase):
    """`DrivingStereo_Dataset(split=..., training=False)` of evaluation/drivingstereo_weather_test.py:27."""

    def __init__(self, split=None, training=False):
        self.N, self.SEED = WEATHER_SPLITS[split]
        super().__init__()


SETS = {"sceneflow": (7, 100), "kitti2015": (5, 200), "kitti2012": (4, 300), "middlebury": (4, 400), "eth3d": (5, 500),
        **WEATHER_SPLITS}


class StubModel(torch.nn.Module):
    """model(left, right) -> the stored prediction of the image whose (seed, index) the left view encodes."""

    def __init__(self):
        super().__init__()
        self.cache = {}

    def forward(self, left, right):
        code = int(round(float(left.flatten()[0])))
        seed, i = code // 1000, code % 1000
        if seed not in self.cache:
            n = [v[0] for v in SETS.values() if v[1] == seed][0]
            self.cache[seed] = synthetic_eval_set(n, seed)[0]
        return self.cache[seed][i].unsqueeze(0).to(left.device)


def _load_reference(fname):
    pkg = types.ModuleType("stereo_toolbox")
    pkg.__path__ = []
    ds = types.ModuleType("stereo_toolbox.datasets")
    for c in (SceneFlow_Dataset, KITTI2015_Dataset, KITTI2012_Dataset, MiddleburyEval3_Dataset, ETH3D_Dataset, DrivingStereo_Dataset):
        c.__module__ = "stereo_toolbox.datasets"          # picklable for the DataLoader workers
        setattr(ds, c.__name__, c)
    sys.modules.setdefault("stereo_toolbox", pkg)
    sys.modules["stereo_toolbox.datasets"] = ds
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except ImportError:
            t = types.ModuleType("tqdm")
            t.tqdm = lambda it, **k: it
            sys.modules["tqdm"] = t
    path = os.path.join("/root/reference/stereo_toolbox/evaluation", fname)
    spec = importlib.util.spec_from_file_location("ref_" + fname[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sf = _load_reference("sceneflow_test.py")
    ge = _load_reference("generalization_eval.py")
    model = StubModel()
    m_sf = sf.sceneflow_test(model, device="cpu", show_progress=False, maxdisp=MAXDISP)
    m_ge = ge.generalization_eval(model, device="cpu", maxdisp=MAXDISP)
    ws = _load_reference("drivingstereo_weather_test.py")
    m_ws = ws.drivingstereo_weather_test(model, device="cpu", threshlods=WEATHER_THRESHOLDS, maxdisp=MAXDISP)
    out = os.path.join(HERE, "metrics.npz")
    np.savez_compressed(out, sceneflow=np.asarray(m_sf, dtype=np.float64), generalization=np.asarray(m_ge, dtype=np.float64),
                        weather=np.asarray(m_ws, dtype=np.float64), maxdisp=np.int64(MAXDISP))
    print("wrote", out, "\n", m_sf, "\n", m_ge, "\n", m_ws)


if __name__ == "__main__":
    main()
