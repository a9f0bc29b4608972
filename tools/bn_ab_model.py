"""Runs the compact MolecularGCN forward/backward (fp32) while every dl_bn_* call is ALSO issued to a second build of the
library on the same inputs; prints the calls whose outputs differ.  usage: python tools/bn_ab_model.py libdruglamp_hip_slp.so"""
import ctypes as C
import os
import sys
sys.path.insert(0, ".")
import torch
from druglamp_amd import _lib, ops
L1 = _lib.lib()
L2 = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1]))
for name, (res, args) in _lib.SIGNATURES.items():
    fn = getattr(L2, name)
    fn.restype, fn.argtypes = res, args


class Both:
    """Proxy: bn calls go to L1, then to L2 with the output pointers redirected to clones; differences are printed."""
    OUT = {"dl_bn_stats": [7], "dl_bn_apply_fwd": [1], "dl_bn_bwd_reduce": [10], "dl_bn_bwd_apply": [8], "dl_bn_tail_fix": [0],
           "dl_bn_finalize": [4, 5, 6]}

    def __getattr__(self, name):
        f1 = getattr(L1, name)
        if name not in self.OUT:
            return f1
        f2 = getattr(L2, name)

        def call(*a):
            a = list(a)
            outs = self.OUT[name]
            # find the tensors behind the output pointers
            live = {t.data_ptr(): t for t in _live}
            before = {i: live[a[i]].clone() for i in outs if a[i] in live}
            rc = f1(*a)
            torch.cuda.synchronize()
            res1 = {i: live[a[i]].clone() for i in before}
            for i, t in before.items():
                live[a[i]].copy_(t)
            f2(*a)
            torch.cuda.synchronize()
            for i in before:
                d = (live[a[i]].double() - res1[i].double()).abs().max() / (res1[i].double().abs().max() + 1e-30)
                if float(d) > 1e-6:
                    print("DIFF %-18s out arg %d  rel %.3e  shape %s args %s" % (name, i, float(d), tuple(res1[i].shape), [x for x in a if isinstance(x, (int, float))][:8]))
                live[a[i]].copy_(res1[i])
            return rc
        return call


_live = []
_orig_empty, _orig_empty_like = torch.empty, torch.empty_like


def _track(t):
    _live.append(t)
    return t


torch.empty = lambda *a, **k: _track(_orig_empty(*a, **k))
torch.empty_like = lambda *a, **k: _track(_orig_empty_like(*a, **k))
_lib._lib = Both()
import copy
from druglamp_amd.model.basic_model import MolecularGCN
from druglamp_amd.synthetic import make_batch
torch.manual_seed(1)
ref = MolecularGCN(75, 128, True, [128] * 3).to("cuda:0").train()
cmp_ = copy.deepcopy(ref)
ref.compact_padding, cmp_.compact_padding = False, True
cmp_.compact_min_rows = 0
(feat_d, *_), _ = make_batch(6, "cuda:0", seed=9, with_graph=True)
h, adj = feat_d
cot = torch.randn(6, 512, 128, device="cuda:0")
for nm, m in (("full", ref), ("compact", cmp_)):
    print("==", nm)
    o = m((h, adj))
    (o.float() * cot).sum().backward()
g1, g2 = ref.init_transform.weight.grad, cmp_.init_transform.weight.grad
print("init_transform grad rel diff", float((g1 - g2).abs().max() / g1.abs().max()))
