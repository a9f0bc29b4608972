"""Chained FFN forward (dl_ffn_fwd) against the dl_gemm pair it replaces, at the step's shapes."""
import sys
import torch
sys.path.insert(0, ".")
from druglamp_amd import ops
dev, dt = "cuda:0", torch.bfloat16


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for D, Hd, M in [(512, 2048, 65536), (256, 1024, 65536), (512, 2048, 8192), (256, 1024, 8192)]:
    x = torch.randn(M, D, device=dev).to(dt)
    res = torch.randn(M, D, device=dev).to(dt)
    w1 = (torch.randn(Hd, D, device=dev) * D ** -0.5).to(dt)
    w2 = (torch.randn(D, Hd, device=dev) * Hd ** -0.5).to(dt)
    b1, b2 = torch.randn(Hd, device=dev), torch.randn(D, device=dev)
    pre = torch.empty(M, Hd, device=dev, dtype=dt)
    for p in (0.0, 0.1):
        def pair():
            act = ops.gemm(x, w1, M=M, N=Hd, K=D, bias=b1, act=1, pre_out=pre, dropout_p=p, seed=1)
            return ops.gemm(act, w2, M=M, N=D, K=Hd, bias=b2, dropout_p=p, seed=2, residual=res)
        t0 = timeit(pair)
        t1 = timeit(lambda: ops.ffn_fwd(x, w1, b1, w2, b2, residual=res, dropout_p=p, seed1=1, seed2=2))
        fl = 4.0 * M * D * Hd
        print("D %4d Hd %4d M %6d p %.1f   pair %7.1f us (%.0f TF/s)   chained %7.1f us (%.0f TF/s)" % (D, Hd, M, p, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6), flush=True)
