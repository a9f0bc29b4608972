"""The north-star attention shape with its projections (BASELINE.json north_star: B x (Ld=64 query rows, Lp=512 key rows), d=256 =
4 heads of 64; tools/northstar_block.py measures it): the HIP block (LayerNorm, Q / K / V projections, fused attention, out
projection + residual, and every gradient) against the oracle's building blocks — the functions pinned to the reference's
PMMA by tests/test_oracle_golden.py (block.py:33-62, attention.py:38-42,90-127) — at the north-star fp32 tolerance."""
import importlib.util
import os

import pytest
import torch

from oracle import druglamp_oracle as O
from tests.helpers import elemerr, relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("northstar_block", os.path.join(ROOT, "tools", "northstar_block.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _oracle_block(ns, params, xd, xp, dy, nb):
    """The block from the oracle's own pieces, fp32 on the CPU."""
    sd = {"ln_q.weight": params["ln_q.w"], "ln_q.bias": params["ln_q.b"], "ln_kv.weight": params["ln_kv.w"], "ln_kv.bias": params["ln_kv.b"],
          "q.weight": params["wq"], "q.bias": params["bq"], "kv.weight": params["wkv"], "kv.bias": params["bkv"],
          "o.weight": params["wo"], "o.bias": params["bo"]}
    sd = {k: v.detach().float().cpu().requires_grad_(True) for k, v in sd.items()}
    xd_, xp_ = xd.detach().float().cpu().requires_grad_(True), xp.detach().float().cpu().requires_grad_(True)
    D = ns.D
    nq = O._ln(sd, "ln_q", xd_.view(nb, ns.LQ, D), ns.EPS)
    nkv = O._ln(sd, "ln_kv", xp_.view(nb, ns.LK, D), ns.EPS)
    q = O._heads(O._lin(sd, "q", nq), ns.H)
    kv = O._lin(sd, "kv", nkv)
    k, v = O._heads(kv[..., :D].contiguous(), ns.H), O._heads(kv[..., D:].contiguous(), ns.H)
    a, _ = O._sdpa(q, k, v)
    y = O._lin(sd, "o", a).reshape(nb * ns.LQ, D) + xd_
    (y * dy.detach().float().cpu()).sum().backward()
    g = {"wq": sd["q.weight"].grad, "bq": sd["q.bias"].grad, "wkv": sd["kv.weight"].grad, "bkv": sd["kv.bias"].grad,
         "wo": sd["o.weight"].grad, "bo": sd["o.bias"].grad, "ln_q.w": sd["ln_q.weight"].grad, "ln_q.b": sd["ln_q.bias"].grad,
         "ln_kv.w": sd["ln_kv.weight"].grad, "ln_kv.b": sd["ln_kv.bias"].grad, "xd": xd_.grad, "xp": xp_.grad}
    return y.detach(), g


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_northstar_block_against_the_oracle(dt, tol):
    ns = _tool()
    dev = torch.device("cuda:0")
    nb = 6
    params = ns.make_params(dev)
    xd, xp, dy = ns.make_inputs(dev, nb)
    blk = ns.Block(params, dt, nb)
    y = blk.forward(xd.to(dt), xp.to(dt))
    g = blk.backward(dy.to(dt))
    yr, gr = _oracle_block(ns, params, xd, xp, dy, nb)
    assert relerr(y, yr) <= tol and elemerr(y, yr) <= 10 * tol
    for k in gr:
        # key-projection bias: analytically zero gradient (softmax shift invariance), rounding noise on both sides
        if k == "bkv":
            assert relerr(g[k][ns.D:], gr[k][ns.D:]) <= 3 * tol, k
            assert float(g[k][:ns.D].abs().max()) <= 3 * tol * float(gr[k][ns.D:].abs().max()), k
        else:
            assert relerr(g[k], gr[k]) <= 3 * tol, k
