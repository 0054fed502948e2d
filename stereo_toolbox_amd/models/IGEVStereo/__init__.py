from .aggregation import BasicConv, FeatureAtt, IGEVCostAggregation, hourglass  # noqa: F401
from .submodule import (build_gwc_volume, disparity_regression, groupwise_correlation, init_disparity,  # noqa: F401
                        init_gwc_volume)
