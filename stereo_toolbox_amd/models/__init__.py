"""Mirror of reference stereo_toolbox/models/__init__.py for the cost-volume-filtering family."""
import torch

from .GwcNet import GwcNet, GwcNet_G, GwcNet_GC  # noqa: F401
from .PSMNet import PSMNet  # noqa: F401
from .ACVNet import ACVNet  # noqa: F401
from .PCWNet import PCWNet_G, PCWNet_GC  # noqa: F401


def load_checkpoint_flexible(model, checkpoint_path, state_dict_key=None):
    """reference models/__init__.py:20-51: load a checkpoint, adding/stripping the DDP `module.`
    prefix as needed and keeping only keys the model has."""
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    sd = ckpt[state_dict_key] if state_dict_key is not None else ckpt
    target = model.state_dict()
    model_has_prefix = any(k.startswith("module.") for k in target)
    fixed = {}
    for k, v in sd.items():
        has = k.startswith("module.")
        if has and not model_has_prefix:
            k = k[len("module."):]
        elif not has and model_has_prefix:
            k = "module." + k
        if k in target:
            fixed[k] = v
    model.load_state_dict(fixed, strict=False)
    return model
