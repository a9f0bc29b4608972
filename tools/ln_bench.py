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
# ---- other elementwise passes of the step --------------------------------------------------------------------------
M, D = 65536, 512
x = torch.randn(M, D, device=dev).to(dt); pre = torch.randn(M, 4 * D, device=dev).to(dt); dyb = torch.randn(M, 4 * D, device=dev).to(dt)
td = t(lambda: ops.dropout_apply(x, 0.1, 5))
print("dropout_apply %d x %d: %.1f us (%.2f TB/s)" % (M, D, td, 2 * M * D * 2 / td / 1e6))
tg = t(lambda: ops.gelu_bwd(dyb[:, :1024].contiguous(), pre[:, :1024].contiguous()))
print("gelu_bwd (+2 copies) %d x 1024: %.1f us" % (M, tg))
a = torch.randn(2, M, 256, device=dev).to(dt)
ti = t(lambda: ops.interleave_streams(a)); ti2 = t(lambda: ops.interleave_streams(a.reshape(M, 512).contiguous().reshape(1, M, 512)[0], inverse=True))
print("interleave_streams 2 x %d x 256: %.1f us (%.2f TB/s), inverse %.1f us" % (M, ti, 2 * a.numel() * 2 / ti / 1e6, ti2))
p1 = torch.randn(M, 128, device=dev).to(dt); p2 = torch.randn(M, 128, device=dev).to(dt)
tc = t(lambda: ops.concat2(p1, p2))
print("concat2 %d x (128 | 128): %.1f us (%.2f TB/s)" % (M, tc, 2 * (p1.numel() + p2.numel()) * 2 / tc / 1e6))
v = torch.randn(256, 256, 256, device=dev).to(dt); lg = torch.randn(256, 256, 8, device=dev).to(dt)
out, gate = ops.token_gate_fwd(v, lg, 8, True)
tgf = t(lambda: ops.token_gate_fwd(v, lg, 8, True)); tgb = t(lambda: ops.token_gate_bwd(out, v, gate, 8, True))
print("token_gate fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (tgf, 2 * v.numel() * 2 / tgf / 1e6, tgb, 3 * v.numel() * 2 / tgb / 1e6))
xf = torch.randn(14_000_000, device=dev); gf = torch.randn_like(xf); m1 = torch.zeros_like(xf); m2 = torch.zeros_like(xf)
tad = t(lambda: ops.adamw_step(xf, gf, m1, m2, lr=1e-4, step=3))
print("adamw 14M params: %.1f us (%.2f TB/s)" % (tad, 28 * xf.numel() / tad / 1e6))
xc = torch.randn(M, 512, device=dev)
tcast = t(lambda: ops.cast(xc, dt))
print("cast f32->bf16 %d x 512: %.1f us (%.2f TB/s)" % (M, tcast, 6 * xc.numel() / tcast / 1e6))
z = torch.randn(256, 2312, 128, device=dev).to(dt)
tsp = t(lambda: ops.cnn_sitepool_fwd(z, 2304, 4, 9)); pooled = ops.cnn_sitepool_fwd(z, 2304, 4, 9); tsb = t(lambda: ops.cnn_sitepool_bwd(pooled, 2304, 4, 9))
print("cnn_sitepool fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (tsp, z.numel() * 2 / tsp / 1e6, tsb, z.numel() * 2 / tsb / 1e6))
xp = torch.randn(256, 2304, 640, device=dev).to(dt)
tfp = t(lambda: ops.fill_pool(xp, 9, dt))
print("fill_pool 256 x 2304 x 640: %.1f us (%.2f TB/s)" % (tfp, xp.numel() * 2 / tfp / 1e6))
pe = torch.randn(256, 256, device=dev).to(dt); xr = torch.randn(M, 256, device=dev).to(dt)
tar = t(lambda: ops.add_rowmod_dropout(xr, pe, 0.1, 3)); trs = t(lambda: ops.rowmod_sum(xr, 256))
print("add_rowmod_dropout %.1f us (%.2f TB/s)  rowmod_sum %.1f us (%.2f TB/s)" % (tar, 2 * xr.numel() * 2 / tar / 1e6, trs, xr.numel() * 2 / trs / 1e6))
