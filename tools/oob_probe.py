"""Find out-of-bounds device accesses: run training steps with the caching allocator OFF (PYTORCH_NO_CUDA_MEMORY_CACHING=1: every
tensor is its own hipMalloc, so an access beyond a tensor faults instead of landing in a neighbour) and a synchronisation after
every druglamp_amd.ops call; the last name printed before the process aborts is the launch at fault.
    PYTORCH_NO_CUDA_MEMORY_CACHING=1 python tools/oob_probe.py [model] [epoch] [batch]"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops                                    # noqa: E402
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into  # noqa: E402
from druglamp_amd.model import MInterface  # noqa: E402
from druglamp_amd.synthetic import make_batch  # noqa: E402
from druglamp_amd.trainer import Trainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "DrugLAMP"
epoch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
quiet = os.environ.get("OOB_QUIET", "0") == "1"


def wrap(nm, fn):
    def w(*a, **k):
        if not quiet:
            print("ops." + nm, flush=True)
        out = fn(*a, **k)
        torch.cuda.synchronize()
        return out
    return w


for nm in dir(ops):
    f = getattr(ops, nm)
    if isinstance(f, types.FunctionType) and not nm.startswith("_") and f.__module__ == ops.__name__ and nm not in (
            "check", "guard_flags", "guard_text", "check_guard_flags", "manual_seed", "next_seed", "use_seed_offset", "seed_offset_tensor",
            "prof_tag", "deferred_reductions", "dynamic_tiles", "reset_tickets", "weight_prep_launches"):
        setattr(ops, nm, wrap(nm, f))

dev = torch.device("cuda", 0)
cfg = load_yaml_into(get_cfg_defaults(), name)
model = MInterface(name, cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16)
tr.set_lrs(1e-4, 3e-5, 1e-5)
for i in range(3):
    batch, meta = make_batch(B, dev, seed=50 + i, with_graph=True, llm_dtype=torch.bfloat16)
    print("== step", i, flush=True)
    out = tr.training_step(batch, meta=meta, cur_epoch=epoch)
    torch.cuda.synchronize()
    print({k: float(v) for k, v in out.items()}, flush=True)
tr.check_device_flags()
print("no fault")
