"""A/B of the fp32 BatchNorm kernels between two builds of the library (DL_USE_STUDY_LIB): dumps every op's output on fixed
inputs; `python tools/bn_ab.py cmp a.pt b.pt` prints the largest relative differences (and against an fp64 restatement)."""
import sys
import torch

if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        if k.startswith("ref_"):
            continue
        d = (a[k].double() - b[k].double()).abs().max() / (a[k].double().abs().max() + 1e-30)
        ra = (a[k].double() - a["ref_" + k]).abs().max() / (a["ref_" + k].abs().max() + 1e-30)
        rb = (b[k].double() - b["ref_" + k]).abs().max() / (b["ref_" + k].abs().max() + 1e-30)
        print("%-14s a-vs-b %.2e   a-vs-fp64 %.2e   b-vs-fp64 %.2e" % (k, float(d), float(ra), float(rb)))
    sys.exit(0)

sys.path.insert(0, ".")
from druglamp_amd import ops
torch.manual_seed(0)
dev = "cuda:0"
B, LP, lead, C, w = 6, 136, 128, 128, 48
R = B * LP
x = torch.randn(R, C, device=dev)
dy = torch.randn(R, C, device=dev)
out = {}
s_lead = ops.bn_stats(x, LP, 0, lead)
s_tail = ops.bn_stats(x, LP, lead, LP - lead)
out["stats_lead"], out["stats_tail"] = s_lead.clone(), s_tail.clone()
xd = x.double().view(B, LP, C)
out["ref_stats_lead"] = torch.cat([xd[:, :lead].sum((0, 1)), (xd[:, :lead] ** 2).sum((0, 1))])
out["ref_stats_tail"] = torch.cat([xd[:, lead:].sum((0, 1)), (xd[:, lead:] ** 2).sum((0, 1))])
sums = s_lead + w * s_tail
n = B * (lead + w * (LP - lead))
mean, var, rstd = ops.bn_finalize(sums, n, 1e-5)
g = torch.rand(C, device=dev) + 0.5
bta = torch.randn(C, device=dev)
y = ops.bn_apply_fwd(x, mean, rstd, g, bta, 0, 0, 0)
out["apply"] = y.clone()
out["ref_apply"] = (x.double() - mean.double()) * rstd.double() * g.double() + bta.double()
s2 = ops.bn_bwd_reduce(dy, x, mean, rstd, 0, 0, 0)
yh = (x.double() - mean.double()) * rstd.double()
out["bwd_reduce"] = s2.clone()
out["ref_bwd_reduce"] = torch.cat([dy.double().sum(0), (dy.double() * yh).sum(0)])
dx = ops.bn_bwd_apply(dy, x, mean, rstd, g, s2, 1.0 / n, False, 0, 0, 0)
out["bwd_apply"] = dx.clone()
ref_dx = g.double() * rstd.double() * (dy.double() - s2[:C].double() / n - yh * s2[C:].double() / n)
out["ref_bwd_apply"] = ref_dx.clone()
ops.bn_tail_fix(dx, x, mean, rstd, g, s2, 1.0 / n, w, LP, lead)
out["tail_fix"] = dx.clone()
r2 = ref_dx.view(B, LP, C).clone()
corr = (w - 1) * g.double() * rstd.double() * (s2[:C].double() / n + yh.view(B, LP, C)[:, lead:] * s2[C:].double() / n)
r2[:, lead:] -= corr
out["ref_tail_fix"] = r2.view(R, C)
dxr = ops.bn_bwd_apply(dy, x, mean, rstd, g, s2, 1.0 / n, True, 0, 0, 0)
out["bwd_apply_relu"] = dxr.clone()
out["ref_bwd_apply_relu"] = ref_dx * (x.double() > 0)
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[1])
