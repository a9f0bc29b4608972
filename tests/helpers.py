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


def _log_margin(kind, value, a, b):
    """DL_PARITY_LOG=<file>: every relerr / elemerr evaluation of a test run is appended as one JSON line (test id, call site,
    both measures) — tools/parity_margins.py turns the file into the per-test margin table under profiles/."""
    path = os.environ.get("DL_PARITY_LOG")
    if not path:
        return
    import inspect
    test, line = "?", 0
    for fr in inspect.stack()[2:]:
        if fr.function.startswith("test_"):
            test, line = fr.function, fr.lineno
            break
    d = (a - b).abs()
    rec = {"test": os.environ.get("PYTEST_CURRENT_TEST", test).split(" ")[0], "line": line, "kind": kind, "value": value,
           "relerr": float(d.max() / (b.abs().max() + 1e-30)),
           "elemerr": float((d / (b.abs() + 1e-2 * b.abs().max() + 1e-30)).max()), "numel": int(b.numel())}
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b) if not isinstance(b, torch.Tensor) else b).detach().double().cpu()
    v = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    _log_margin("relerr", v, a, b)
    return v


def elemerr(a, b, floor=1e-2):
    """Element-wise relative error with an absolute floor: max_i |a_i - b_i| / (|b_i| + floor * max|b|).  `relerr` divides every
    difference by the LARGEST reference element, so one large element hides relative error on the small ones;
    here an element is judged against its own magnitude down to `floor` (1 %) of the largest."""
    a = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b) if not isinstance(b, torch.Tensor) else b).detach().double().cpu()
    v = float(((a - b).abs() / (b.abs() + floor * b.abs().max() + 1e-30)).max())
    _log_margin("elemerr", v, a, b)
    return v


def pmma_dropout_masks(tag, B, L, d, p):
    return {k: torch.from_numpy(v) for k, v in synth.pmma_dropout_masks(tag, B, L, d, p).items()}


def check_sub(x, g, key, tol, elem_tol=None):
    """compare a big tensor against its stored sub-sample (head/tail rows, row norms, checksum); elem_tol: also element by
    element (elemerr) on the stored rows."""
    x = x.detach().double().cpu()
    assert relerr(x[:, :4], g[key + "/head"]) <= tol, key + " head"
    assert relerr(x[:, -4:], g[key + "/tail"]) <= tol, key + " tail"
    assert relerr(x.norm(dim=-1), g[key + "/rownorm"]) <= tol, key + " rownorm"
    if elem_tol is not None:
        assert elemerr(x[:, :4], g[key + "/head"]) <= elem_tol and elemerr(x[:, -4:], g[key + "/tail"]) <= elem_tol, key + " element-wise"


def gradnorms(g, prefix="gradnorm"):
    return {k[len(prefix) + 1:]: float(g[k]) for k in g.files if k.startswith(prefix + "/")}
