"""LayerNorm / BatchNorm pass timings at the step's sizes: achieved algorithmic TB/s (bf16)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for M, D in [(65536, 256), (65536, 512), (131072, 256)]:
    x = torch.randn(M, D, device=dev).to(dt); dy = torch.randn(M, D, device=dev).to(dt); dres = torch.randn(M, D, device=dev).to(dt)
    g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    tf = t(lambda: ops.layernorm_fwd(x, g, b, 1e-6))
    tb = t(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres=dres))
    tb2 = t(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g))
    by = M * D * 2
    print("LN %6d x %4d: fwd %.1f us (%.2f TB/s)   bwd+dres %.1f us (%.2f TB/s)   bwd %.1f us (%.2f TB/s)" % (
        M, D, tf, 2 * by / tf / 1e6, tb, 4 * by / tb / 1e6, tb2, 3 * by / tb2 / 1e6), flush=True)
R, C = 256 * 2312, 128
y = torch.randn(R, C, device=dev).to(dt); dz = torch.randn(R, C, device=dev).to(dt)
ts = t(lambda: ops.bn_stats(y, 2312, 4, 2304))
sums = ops.bn_stats(y, 2312, 4, 2304)
mean, var, rstd = ops.bn_finalize(sums, 256 * 2304, 1e-5)
gm = torch.ones(C, device=dev); bt = torch.zeros(C, device=dev)
ta = t(lambda: ops.bn_apply_fwd(y, mean, rstd, gm, bt, 2312, 4, 2304))
tr = t(lambda: ops.bn_bwd_reduce(dz, y, mean, rstd, 2312, 4, 2304))
s2 = ops.bn_bwd_reduce(dz, y, mean, rstd, 2312, 4, 2304)
tba = t(lambda: ops.bn_bwd_apply(dz, y, mean, rstd, gm, s2, 1.0 / (256 * 2304), True, 2312, 4, 2304))
by = R * C * 2
print("BN %d x %d: stats %.1f us (%.2f TB/s)  apply %.1f us (%.2f TB/s)  bwd_reduce %.1f us (%.2f TB/s)  bwd_apply %.1f us (%.2f TB/s)" % (
    R, C, ts, by / ts / 1e6, ta, 2 * by / ta / 1e6, tr, 2 * by / tr / 1e6, tba, 3 * by / tba / 1e6))
