"""MInterface (reference: model/model_interface.py:4-33): picks the model class by file/class name and
instantiates it with the constructor arguments found among `other_args`, then **config."""
import importlib
import inspect


class MInterface(object):
    def __init__(self, model_name, config):
        self.model_name = model_name
        self.config = config

    def load_model(self, **other_args):
        name = self.model_name
        camel_name = "".join(name.split("_"))
        try:
            Model = getattr(importlib.import_module("." + name, package=__package__), camel_name)
        except Exception:
            raise ValueError(f"Invalid Module File Name or Invalid Class Name {name}.{camel_name}!")
        return self.instancialize(Model, self.config, **other_args)

    def instancialize(self, Model, config, **other_args):
        class_args = inspect.getfullargspec(Model.__init__).args[1:]
        args1 = {a: other_args[a] for a in class_args if a in other_args}
        args1.update(**config)
        return Model(**args1)
