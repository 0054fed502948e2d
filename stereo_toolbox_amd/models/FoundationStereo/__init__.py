from .submodule import (build_concat_volume, build_gwc_volume, disparity_regression, groupwise_correlation,  # noqa: F401
                        init_comb_volume)
