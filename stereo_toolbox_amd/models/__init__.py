"""Mirror of reference stereo_toolbox/models/__init__.py for the cost-volume-filtering family."""
import torch

from .GwcNet import GwcNet, GwcNet_G, GwcNet_GC  # noqa: F401
from .PSMNet import PSMNet  # noqa: F401
from .ACVNet import ACVNet  # noqa: F401
from .PCWNet import PCWNet_G, PCWNet_GC  # noqa: F401
from .CFNet import CFNet  # noqa: F401
from . import IGEVStereo  # noqa: F401  (initial-volume entry points only)
from . import FoundationStereo  # noqa: F401  (normalised initial volume only)


def load_checkpoint_flexible(model, checkpoint_path, state_dict_key=None):
    """reference models/__init__.py:20-51: load a checkpoint, adding/stripping the DDP `module.` prefix as needed,
    keeping only keys the model has, and reporting missing / unexpected keys like the reference does."""
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    sd = ckpt[state_dict_key] if state_dict_key is not None else ckpt
    target = model.state_dict()
    renamed = {}
    for k, v in sd.items():
        if k not in target:
            k = k[len("module."):] if k.startswith("module.") else "module." + k
        renamed[k] = v
    target.update({k: v for k, v in renamed.items() if k in target})
    missing, unexpected = model.load_state_dict(target)
    if missing:
        print("Missing keys: ", ",".join(missing))
    if unexpected:
        print("Unexpected keys: ", ",".join(unexpected))
    return model
