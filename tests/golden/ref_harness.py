"""Import the reference (Lzcstan/DrugLAMP at /root/reference) in THIS container, with empty stand-ins
for the third-party packages it imports at module scope but that are absent here (dgl, rdkit,
torch_geometric, lightning_utilities, yacs).  Harness-side only: used by make_golden.py to produce
the committed fixtures; nothing under /root/reference is copied, and nothing here runs on the GPU box.
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

REF = "/root/reference"


class _CfgNode(dict):
    """Minimal attribute-dict with yacs' clone()."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        out = _CfgNode()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, _CfgNode) else (list(v) if isinstance(v, list) else v)
        return out


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    if "dgl" not in sys.modules:
        fn = mod("dgl.function", copy_u=None, sum=None, copy_src=None)
        mod("dgl", function=fn)
    if "rdkit" not in sys.modules:
        chem = mod("rdkit.Chem")
        mod("rdkit", Chem=chem)
    if "torch_geometric" not in sys.modules:
        u = mod("torch_geometric.utils", from_smiles=None)
        mod("torch_geometric", utils=u)
    if "lightning_utilities" not in sys.modules:
        rz = mod("lightning_utilities.core.rank_zero", rank_zero_only=lambda f: f)
        core = mod("lightning_utilities.core", rank_zero=rz)
        mod("lightning_utilities", core=core)
    if "yacs" not in sys.modules:
        cfgm = mod("yacs.config", CfgNode=_CfgNode)
        mod("yacs", config=cfgm)


def import_reference():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present at %s (fixtures can only be generated in the build "
                           "container)" % REF)
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)


def default_cfg():
    """configs.get_cfg_defaults() merged with configs/DrugLAMP2C2P.yaml values that the models read."""
    import_reference()
    from configs import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.DECODER.BINARY = 1
    cfg.RS.MAX_MARGIN = 0.5
    cfg.RS.RESET_EPOCH = 100
    cfg.RS.INIT_EPOCH = 5
    cfg.RS.EPOCH_STEP = 5
    cfg.RS.SSL = True
    cfg.RS.CM = True
    cfg.SOLVER.LR, cfg.SOLVER.SSL_LR, cfg.SOLVER.CM_LR = 1e-4, 3e-5, 3e-5
    cfg.SOLVER.BATCH_SIZE, cfg.SOLVER.MAX_EPOCH = 16, 100
    return cfg
