"""stereo_toolbox_amd: the cost-volume hot path of xxxupeng/stereo_toolbox (PSMNet / GwcNet / ACVNet)
re-built for AMD MI355X (gfx950): hand-written HIP kernels behind the reference's Python API."""
__version__ = "0.1.0"
