from .stackhourglass import PSMNet  # noqa: F401
