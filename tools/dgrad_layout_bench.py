"""Data-gradient products of the row-wise layers: dx[M][N] = g[M][K] W, with W as the K-slow forward image ([K][N] rows, 'NT' in
tools/gemm_shapes.py) against its [N][K] transpose (both operands K-contiguous, 'NN').  usage: python tools/dgrad_layout_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
dt = torch.bfloat16
for (M, N, K) in [(65536, 648, 256), (65536, 256, 128), (65536, 128, 648), (65536, 128, 128), (65536, 256, 256), (34816, 128, 128),
                  (34816, 256, 128), (34816, 392, 256), (65536, 128, 256), (131072, 128, 256), (8192, 648, 256), (8192, 256, 128), (8192, 128, 128)]:
    g = torch.randn(M, K, device="cuda").to(dt)
    w = torch.randn(K, N, device="cuda").to(dt)          # forward image [out = K][in = N]
    wT = w.t().contiguous()
    t_nt = timeit(lambda: ops.gemm(g, w, M=M, N=N, K=K, w_kslow=True, ldw=N))
    t_nn = timeit(lambda: ops.gemm(g, wT, M=M, N=N, K=K))
    d = float((ops.gemm(g, w, M=M, N=N, K=K, w_kslow=True, ldw=N).float() - ops.gemm(g, wT, M=M, N=N, K=K).float()).abs().max())
    print("%7d x %4d x %4d   K-slow W %7.1f us   transposed W %7.1f us   max diff %.3g" % (M, N, K, t_nt, t_nn, d), flush=True)
