"""ADVICE r4 bisect: the compact MolecularGCN (fp32) forward / backward with EVERY druglamp_amd.ops call's outputs recorded in
call order; run once per library build, then compare the two recordings to find the first op whose output differs.

    python tools/bn_bisect.py rec out_a.pt                                        # product build
    DL_USE_STUDY_LIB=libdruglamp_hip_noslpbn.so python tools/bn_bisect.py rec out_b.pt
    python tools/bn_bisect.py cmp out_a.pt out_b.pt
"""
import sys
import types

import torch

if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    print(len(a), len(b), "recorded calls")
    shown = zeros_shown = 0
    for i, ((na, ta), (nb, tb)) in enumerate(zip(a, b)):
        assert na == nb, (i, na, nb)
        for j, (x, y) in enumerate(zip(ta, tb)):
            if x.shape != y.shape:
                print(i, na, j, "SHAPE", x.shape, y.shape)
                continue
            x, y = x.double(), y.double()
            # ReLU kinks: elements that are exactly zero in one build and not in the other (a pre-activation within rounding of 0)
            zm = (x == 0) != (y == 0)
            if zm.any() and zeros_shown < 12:
                other = torch.where(x == 0, y, x)[zm].abs()
                print("call %4d %-22s out %d: %d elements zero in one build only; the other build's values there: max |v| %.3e (tensor max %.3e); "
                      "first at flat %d" % (i, na, j, int(zm.sum()), float(other.max()), float(x.abs().max()), int(zm.flatten().nonzero()[0])))
                zeros_shown += 1
            bad = ~(torch.isfinite(x) & torch.isfinite(y))
            d = (x - y).abs()
            d[bad] = 0
            rel = float(d.max() / (x.abs()[~bad].max() + 1e-30)) if (~bad).any() else 0.0
            nanm = int((torch.isnan(x) != torch.isnan(y)).sum())
            if rel > 1e-6 or nanm:
                idx = int(d.flatten().argmax())
                print("call %4d %-22s out %d shape %-18s rel diff %.3e  nan mismatch %d  at flat %d: %.6g vs %.6g" % (
                    i, na, j, tuple(x.shape), rel, nanm, idx, float(x.flatten()[idx]), float(y.flatten()[idx])))
                shown += 1
        if shown >= 25:
            break
    sys.exit(0)

sys.path.insert(0, ".")
from druglamp_amd import ops                                    # noqa: E402

rec = []


def wrap(name, fn):
    def w(*a, **k):
        out = fn(*a, **k)
        torch.cuda.synchronize()
        ts = [t for t in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(t)]
        rec.append((name + ("[act=%s]" % k["act"] if "act" in k else ""), [t.detach().float().cpu().clone() for t in ts]))
        return out
    return w


for nm in dir(ops):
    f = getattr(ops, nm)
    if isinstance(f, types.FunctionType) and not nm.startswith("_") and f.__module__ == ops.__name__ and nm not in (
            "check", "guard_flags", "guard_text", "check_guard_flags", "manual_seed", "next_seed", "use_seed_offset", "seed_offset_tensor",
            "prof_tag", "deferred_reductions", "dynamic_tiles", "reset_tickets", "weight_prep_launches"):
        setattr(ops, nm, wrap(nm, f))

import copy                                                      # noqa: E402
from druglamp_amd.model.basic_model import MolecularGCN         # noqa: E402
from druglamp_amd.synthetic import make_batch                   # noqa: E402
torch.manual_seed(1)
ref = MolecularGCN(75, 128, True, [128] * 3).to("cuda:0").train()
ref.compute_dtype = torch.float32
cmp_ = copy.deepcopy(ref)
cmp_.compact_padding, cmp_.compact_min_rows = True, 0
(feat_d, *_), _ = make_batch(6, "cuda:0", seed=9, with_graph=True)
h, adj = feat_d
cot = torch.randn(6, 512, 128, device="cuda:0")
o = cmp_((h, adj))
(o.float() * cot).sum().backward()
rec.append(("param_grads", [p.grad.detach().float().cpu().clone() for p in cmp_.parameters()]))
torch.save(rec, sys.argv[2])
print("recorded", len(rec), "calls ->", sys.argv[2])
