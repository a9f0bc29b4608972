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


def load_reference_checkpoint(model, checkpoint, strict: bool = True, allow_pickle: bool = False):
    """Load a checkpoint written by the reference's Lightning run (trainer.py:151-156 ModelCheckpoint on the ExpModule):
    a dict whose "state_dict" maps "exp_model.<key>" to tensors (ExpModule stores the model as self.exp_model,
    trainer.py:43) next to metric-object states.  `checkpoint` is a path or the loaded dict; a bare state_dict (with or
    without the prefix) is accepted too.  The lazily created SimSiam projectors are built first when the checkpoint holds
    them (the reference needs one SSL forward before it can load such a checkpoint).  Returns load_state_dict's result
    (missing / unexpected keys).  strict=True by default: the dict is already filtered to the `exp_model.` keys (Lightning's
    metric states never reach load_state_dict), so a missing or renamed key is an error, not a silently random weight.
    With strict=False (what the reference itself passes, trainer.py:134) anything missing besides the lazily built SimSiam
    projector keys still raises.  Files are read with weights_only=True; allow_pickle=True opts in to full unpickling
    when that fails."""
    import torch
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "read"):
        try:
            ck = torch.load(checkpoint, map_location="cpu", weights_only=True)
        except Exception:
            # Lightning checkpoints carry callback / hyper-parameter pickles next to the tensors; unpickling arbitrary
            # objects from a file is the caller's decision, not a silent default
            if not allow_pickle:
                raise
            if hasattr(checkpoint, "seek"):
                checkpoint.seek(0)
            ck = torch.load(checkpoint, map_location="cpu", weights_only=False)
    else:
        ck = checkpoint
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
    prefix = "exp_model."
    if any(k.startswith(prefix) for k in sd):
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    proj = [k for k in sd if ".projector." in k]
    if proj and hasattr(model, "ssl_model"):
        dims = {}
        for name in ("net", "llm_net"):
            w = sd.get("ssl_model.%s.projector.0.weight" % name)
            if w is not None:
                dims[name] = int(w.shape[1])
        if len(dims) == 2:
            model.ssl_model.build_projectors(dims["net"], dims["llm_net"], device=next(model.parameters()).device)
    res = model.load_state_dict(sd, strict=strict)
    hard = [k for k in res.missing_keys if ".projector." not in k]
    if hard:
        raise RuntimeError("load_reference_checkpoint: %d parameters of the model are not in the checkpoint (first: %s)" % (len(hard), hard[:4]))
    return res
