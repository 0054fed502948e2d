"""Shape, inputs and sampling rule shared by tests/golden/make_golden_toy_train.py (build container, imports the reference) and
the train-step tests that read its fixtures (tests/test_models.py; nothing here touches /root/reference)."""
from stereo_toolbox_amd.utils import synthetic_tensor

B, H, W, D = 2, 128, 256, 128          # 1/16 level: D/16 * H/16 * W/16 * B = 8 * 8 * 16 * 2 = 2048 voxels per channel
LOSS_W = (0.5, 0.5, 0.7, 1.0)
STRIDE = 4                             # stored prediction maps: every 4th pixel
SAMPLES = 2048                         # stored gradient samples per tensor
BN_BETA_SHIFT = 1.0                    # filler profile of these fixtures (stereo_toolbox_amd.utils.fill_state_dict): a well-conditioned
                                       # train step -- the reference's own fp32 gradients are 3e-5 (median) of a tensor's max from fp64
PCW = dict(B=2, H=128, W=256, D=64, LOSS_W=(0.5, 0.5, 0.5, 0.7, 1.0, 1.3))   # PCWNet_GC whole-model fixture (6 train-mode outputs)


def fill(module):
    """The fixtures' weights: deterministic filler with the shifted BatchNorm betas, loaded into `module`."""
    from stereo_toolbox_amd.utils import fill_state_dict
    sd = module.state_dict()
    fill_state_dict(sd, bn_beta_shift=BN_BETA_SHIFT)
    module.load_state_dict(sd)
    return module


def feature_maps(kind):
    """Synthetic 1/4-resolution feature maps of the two views for the isolated 3-D path: [gwc_left, gwc_right, concat_left,
    concat_right] (320 channels; concat: 12 for GwcNet_GC, 32 = ACVNet's `concatconv` outputs), uniform in (-1, 1)."""
    h, w, cc = H // 4, W // 4, (32 if kind == "acv" else 12)
    return [synthetic_tensor((B, 320, h, w), 11), synthetic_tensor((B, 320, h, w), 12),
            synthetic_tensor((B, cc, h, w), 13), synthetic_tensor((B, cc, h, w), 14)]


def sample(t):
    """Up to SAMPLES strided elements of a tensor (flattened): the rule both sides apply."""
    f = t.reshape(-1)
    return f[::max(1, f.numel() // SAMPLES)][:SAMPLES]
