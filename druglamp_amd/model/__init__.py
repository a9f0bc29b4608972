from .model_interface import MInterface  # noqa: F401
