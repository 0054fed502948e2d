from .gwcnet import GwcNet, GwcNet_G, GwcNet_GC  # noqa: F401
