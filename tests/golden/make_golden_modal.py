"""Golden vectors for the modal disparity estimators, generated from the reference.

Run in the build container only (needs /root/reference): `python tests/golden/make_golden_modal.py`.
Writes tests/golden/estimators_modal.npz: outputs of the reference's
disparity_estimators/{unimodal,dominant_modal}_disparity_estimator.py on deterministic volumes that the tests
regenerate from seeds (stereo_toolbox_amd.utils.synthetic_modal_volume); only outputs are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/stereo_toolbox")

import disparity_estimators as ref_est  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("ref_split_mode", "/root/reference/stereo_toolbox/loss_functions/split_mode.py")
ref_split = importlib.util.module_from_spec(_spec)       # (the package __init__ imports losses that need absent dependencies)
_spec.loader.exec_module(ref_split)

from stereo_toolbox_amd.utils import synthetic_modal_volume, synthetic_tensor  # noqa: E402

CASES = {"a": (2, 32, 5, 9, 21), "b": (1, 48, 4, 7, 22), "c": (1, 192, 3, 5, 23)}


def main():
    out = {}
    for tag, (B, D, H, W, seed) in CASES.items():
        x = synthetic_modal_volume(B, D, H, W, seed)
        out[f"uni_{tag}"] = ref_est.unimodal_disparity_estimator(x, D).numpy()
        out[f"dom_{tag}"] = ref_est.dominant_modal_disparity_estimator(x, D).numpy()
    # a plain peaky softmax as well (single mode almost everywhere)
    peaky = torch.softmax(synthetic_tensor((2, 16, 6, 10), 13) * 4, 1)
    out["uni_peaky"] = ref_est.unimodal_disparity_estimator(peaky, 16).numpy()
    out["dom_peaky"] = ref_est.dominant_modal_disparity_estimator(peaky, 16).numpy()
    # gradients of the reference estimators w.r.t. the volume (they are differentiable inside the constant mode mask:
    # unimodal_disparity_estimator.py:20-25), upstream gradient = deterministic pattern re-made by the tests
    for tag in ("a", "b"):
        B, D, H, W, seed = CASES[tag]
        for name, fn in (("uni", ref_est.unimodal_disparity_estimator), ("dom", ref_est.dominant_modal_disparity_estimator)):
            x = synthetic_modal_volume(B, D, H, W, seed).requires_grad_()
            gy = synthetic_tensor((B, 1, H, W), 40 + seed)
            fn(x, D).backward(gy)
            out[f"{name}_grad_{tag}"] = x.grad.numpy()
    # split_mode (loss_functions/split_mode.py:9-35): mode and mask of the raw volume, bit-packed mask
    for tag, (B, D, H, W, seed) in CASES.items():
        x = synthetic_modal_volume(B, D, H, W, seed)
        mode, mask = ref_split.split_mode(x, D)
        assert mask.dtype == torch.bool
        out[f"split_mode_{tag}"] = mode.numpy()
        out[f"split_mask_{tag}"] = np.packbits(mask.numpy().reshape(-1))
    mode, mask = ref_split.split_mode(peaky, 16)
    out["split_mode_peaky"], out["split_mask_peaky"] = mode.numpy(), np.packbits(mask.numpy().reshape(-1))
    np.savez_compressed(os.path.join(HERE, "estimators_modal.npz"), **out)
    print("wrote estimators_modal.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
