"""Shapes, seeds and the weight filler shared by tests/golden/make_golden_igev_agg.py (build container, imports the reference)
and tests/test_igev_aggregation.py (runs everywhere: this module imports nothing from /root/reference)."""
from stereo_toolbox_amd.utils import fill_state_dict

MAXDISP, H4, W4, B = 64, 16, 32, 2          # 1/4-resolution volume [B, 8, 16, 16, 32]; levels 1/8, 1/16, 1/32 below it
FEAT_CH = (96, 64, 192, 160)
CLASSIFIER_GAIN = 40.0      # the filler's weights give a near-uniform softmax over D' (init_disp ~ 7.5 everywhere, which hides
                            # errors): the classifier is scaled up so that the regression output is peaky (std ~ 2 px)


def fill(sd):
    fill_state_dict(sd, seed=4321)
    sd["classifier.weight"].mul_(CLASSIFIER_GAIN)
    return sd
