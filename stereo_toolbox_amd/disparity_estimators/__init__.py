"""Drop-in for reference stereo_toolbox/disparity_estimators (`__init__.py:7-15`, `unimodal_disparity_estimator.py`,
`dominant_modal_disparity_estimator.py`) on HIP kernels."""
from .. import ops


def softargmax_disparity_estimator(x, maxdisp=192):
    """sum_d d * x[:, d] -> [B, 1, H, W]."""
    return ops.softargmax(x, maxdisp, keepdim=True)


def argmax_disparity_estimator(x, maxdisp=192):
    """argmax over the disparity axis -> int64 [B, 1, H, W]."""
    return ops.argmax_disparity(x)


def unimodal_disparity_estimator(x, maxdisp=192):
    """Expectation of d over the mode that contains the arg-max, re-normalised -> [B, 1, H, W]."""
    return ops.unimodal_disparity(x, maxdisp)


def dominant_modal_disparity_estimator(x, maxdisp=192):
    """Expectation of d over the heavier of the two main modes of the blurred volume -> [B, 1, H, W]."""
    return ops.dominant_modal_disparity(x, maxdisp)
