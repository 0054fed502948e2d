"""Drop-in for the part of reference stereo_toolbox/loss_functions that sits on the hot path: `split_mode`
(loss_functions/split_mode.py:9-35), the twin of the modal disparity estimators.  The photometric /
smoothness / auto-mask losses of that package belong to the self-supervised trainers and are out of scope."""
from ..ops import split_mode

__all__ = ["split_mode"]
