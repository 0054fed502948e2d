"""Drop-in for reference stereo_toolbox/disparity_estimators/__init__.py:7-15 (HIP kernels)."""
from .. import ops


def softargmax_disparity_estimator(x, maxdisp=192):
    """sum_d d * x[:, d] -> [B, 1, H, W]."""
    return ops.softargmax(x, maxdisp, keepdim=True)


def argmax_disparity_estimator(x, maxdisp=192):
    """argmax over the disparity axis -> int64 [B, 1, H, W]."""
    return ops.argmax_disparity(x)
