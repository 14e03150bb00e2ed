from .spegnet import SPEGNet  # noqa: F401
