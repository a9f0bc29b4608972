"""LayerNorm kernels: repeatability (bitwise, over repeated launches) and accuracy vs fp64 for the model's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, D) in ((65536, 256), (65536, 512), (131072, 256), (2048, 256), (4100, 512), (1000, 256)):
    x = torch.randn(M, D, device=dev).bfloat16(); dy = torch.randn(M, D, device=dev).bfloat16(); dres = torch.randn(M, D, device=dev).bfloat16()
    g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    y0, mean0, rstd0 = ops.layernorm_fwd(x, g, b, 1e-6)
    dx0, dg0, db0 = ops.layernorm_bwd(dy, x, mean0, rstd0, g, dres=dres)
    bad = 0
    for it in range(30):
        y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
        dx, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, g, dres=dres)
        bad += int(not (torch.equal(y, y0) and torch.equal(mean, mean0) and torch.equal(rstd, rstd0) and torch.equal(dx, dx0)
                        and torch.equal(dg, dg0) and torch.equal(db, db0)))
    xd = x.double(); mu = xd.mean(-1, keepdim=True); var = xd.var(-1, unbiased=False, keepdim=True); xh = (xd - mu) / torch.sqrt(var + 1e-6)
    yr = xh * g.double() + b.double()
    gg = dy.double() * g.double()
    dxr = (gg - gg.mean(-1, keepdim=True) - xh * (gg * xh).mean(-1, keepdim=True)) / torch.sqrt(var + 1e-6) + dres.double()
    print("M=%6d D=%d  mismatching repeats %d/30   |y-ref| %.3e  |dx-ref| %.3e  |dg-ref| %.3e (rel)" % (
        M, D, bad, float((y0.double() - yr).abs().max()), float((dx0.double() - dxr).abs().max()),
        float((dg0.double() - (dy.double() * xh).sum(0)).abs().max() / (dy.double() * xh).sum(0).abs().max())), flush=True)
