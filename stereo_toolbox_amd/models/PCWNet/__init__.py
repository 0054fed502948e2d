from .pcwnet import PCWNet, PCWNet_G, PCWNet_GC  # noqa: F401
