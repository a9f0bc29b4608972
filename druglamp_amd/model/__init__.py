from .model_interface import MInterface, load_reference_checkpoint  # noqa: F401
