"""Golden fixtures for SURVEY.md 8f rank 4: the evaluation-path input step and the IGEV-family initial volume.

  python tests/golden/make_golden_igev.py        (build container only: needs /root/reference)

* `pad_to_2x` is the reference's own function (datasets/data_augmentation/__init__.py:57-80).  That file imports
  torchvision (for its colour-jitter augmentations) at the top; torchvision is absent from this image, so an empty
  package object named `torchvision.transforms` with inert `ColorJitter / Compose / functional` attributes is put into
  sys.modules before the reference file is executed from where it lies -- `pad_to_2x` itself is pure numpy.
* `get_transform()` (datasets/utils.py:62-69) IS torchvision (ToTensor + Normalize) and cannot be executed here; the
  oracle restates its published algorithm (oracle/torch_oracle.py: to_tensor_normalize) -- that part of the fixture
  chain is "restated, not executed".
* IGEV: `build_gwc_volume`, `disparity_regression` of models/IGEVStereo/submodule.py (imports torch / numpy only) on the
  96-channel / 8-group configuration of igev_stereo.py:206, forward and backward.

Inputs are regenerated from seeds by the tests; only the reference's outputs are stored: tests/golden/igev_preprocess.npz.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from stereo_toolbox_amd.utils import hash_uniform, synthetic_tensor  # noqa: E402

IMG_SHAPES = {"kitti_like": (37, 125), "exact": (96, 96), "tall": (100, 7)}


def synthetic_image(H, W, seed):
    """uint8 [H,W,3] from the hash generator."""
    return torch.from_numpy((hash_uniform(seed, 9, H * W * 3) * 256.0).astype(np.uint8).reshape(H, W, 3))


def _load_pad_to_2x():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tr.ColorJitter = lambda **k: (lambda img: img)
    tr.Compose = lambda fs: (lambda img: img)
    tr.functional = types.SimpleNamespace(adjust_gamma=lambda img, g: img)
    tv.transforms = tr
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tr)
    path = "/root/reference/stereo_toolbox/datasets/data_augmentation/__init__.py"
    spec = importlib.util.spec_from_file_location("ref_data_augmentation", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # the reference calls `np.lib.pad`, an alias of `np.pad` that numpy >= 2.0 removed (this image has numpy 2.2):
    # the alias to numpy's OWN function is restored, nothing is re-implemented
    if not hasattr(np.lib, "pad"):
        np.lib.pad = np.pad
    return mod.pad_to_2x


def main():
    out = {}
    pad_to_2x = _load_pad_to_2x()
    for tag, (H, W) in IMG_SHAPES.items():
        left, right = synthetic_image(H, W, 51).numpy(), synthetic_image(H, W, 52).numpy()
        disp = synthetic_tensor((H, W), 53, lo=0.0, hi=100.0).numpy()
        mask = (synthetic_tensor((H, W), 54, lo=0.0, hi=1.0) > 0.4).numpy()
        l, r, d, m = pad_to_2x(left, right, disp, mask)
        out[f"{tag}_left"], out[f"{tag}_right"], out[f"{tag}_disp"], out[f"{tag}_mask"] = l, r, d, m
    dist = synthetic_tensor((5, 37, 125), 55, lo=0.0, hi=1.0).numpy()           # a 3-D "distribution" ground truth
    _, _, d3, _ = pad_to_2x(synthetic_image(37, 125, 51).numpy(), synthetic_image(37, 125, 52).numpy(), dist, None)
    out["kitti_like_dist"] = d3

    sys.path.insert(0, "/root/reference/stereo_toolbox/models")
    from IGEVStereo.submodule import build_gwc_volume, disparity_regression
    B, C, H4, W4, maxdisp = 1, 96, 4, 40, 64
    ml = synthetic_tensor((B, C, H4, W4), 61).requires_grad_()
    mr = synthetic_tensor((B, C, H4, W4), 62).requires_grad_()
    vol = build_gwc_volume(ml, mr, maxdisp // 4, 8)                             # igev_stereo.py:206
    gv = synthetic_tensor(tuple(vol.shape), 63)
    vol.backward(gv)
    out["igev_volume"], out["igev_grad_left"], out["igev_grad_right"] = vol.detach().numpy(), ml.grad.numpy(), mr.grad.numpy()
    cost = (synthetic_tensor((B, 1, maxdisp // 4, H4, W4), 64) * 3).requires_grad_()
    prob = F.softmax(cost.squeeze(1), dim=1)                                    # igev_stereo.py:211
    disp = disparity_regression(prob, maxdisp // 4)                             # igev_stereo.py:212
    disp.backward(synthetic_tensor(tuple(disp.shape), 65))
    out["igev_init_disp"], out["igev_init_grad"] = disp.detach().numpy(), cost.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "igev_preprocess.npz"), **out)
    print("wrote igev_preprocess.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
