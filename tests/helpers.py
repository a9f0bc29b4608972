"""Shared helpers for the parity tests: fixture loading, deterministic weights/inputs (oracle/detgen)."""
import json
import os

import numpy as np
import torch

from oracle import detgen, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def sd_spec(g):
    return [(k, tuple(shape), dt) for k, shape, dt in json.loads(str(g["sd"]))]


def det_state_dict(g, salt=0, device="cpu"):
    spec = sd_spec(g)
    vals = detgen.fill_state_dict(spec, salt)
    sd = {k: torch.from_numpy(v).to(device) for k, v in vals.items()}
    # the reference shares ONE ProteinCNN between `protein_extractor` and `ssl_model.extractor`
    # (basic_model.py:79-86): load_state_dict writes the shared tensors twice and the later key wins.
    for k in list(sd):
        if k.startswith("ssl_model.extractor."):
            sd["protein_extractor." + k[len("ssl_model.extractor."):]] = sd[k]
    return sd


def T(name, shape, scale=1.0, salt=0):
    return torch.from_numpy(detgen.normalish(name, shape, salt) * np.float32(scale))


def model_inputs(tag, B, salt=0, **kw):
    return tuple(torch.from_numpy(a) for a in synth.model_inputs(tag, B, salt, **kw))


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not isinstance(b, torch.Tensor) else b).double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check_sub(x, g, key, tol):
    """compare a big tensor against its stored sub-sample (head/tail rows, row norms, checksum)."""
    x = x.detach().double().cpu()
    assert relerr(x[:, :4], g[key + "/head"]) <= tol, key + " head"
    assert relerr(x[:, -4:], g[key + "/tail"]) <= tol, key + " tail"
    assert relerr(x.norm(dim=-1), g[key + "/rownorm"]) <= tol, key + " rownorm"


def gradnorms(g, prefix="gradnorm"):
    return {k[len(prefix) + 1:]: float(g[k]) for k in g.files if k.startswith(prefix + "/")}
