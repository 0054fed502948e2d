from .cfnet import CFNet, cfnet  # noqa: F401
