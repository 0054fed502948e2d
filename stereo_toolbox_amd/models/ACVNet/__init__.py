from .acv import ACVNet  # noqa: F401
