"""Does the operand row pitch matter (L2 / HBM channel camping on power-of-two pitches)?  Same product, padded pitches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (M, N, K) in [(65536, 512, 2048), (65536, 2048, 512), (65536, 1024, 256), (65536, 256, 1024)]:
    row = []
    for padx, padw, pado in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (32, 32, 0), (128, 128, 0)]:
        xb = torch.randn(M, K + padx, device=dev).to(torch.bfloat16); wb = torch.randn(N, K + padw, device=dev).to(torch.bfloat16)
        ob = torch.empty(M, N + pado, device=dev, dtype=torch.bfloat16)
        x, w, out = xb[:, :K], wb[:, :K], ob[:, :N]
        t = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out))
        row.append("x+%d w+%d o+%d: %.0f us (%.0f TF/s)" % (padx, padw, pado, t, 2.0 * M * N * K / t / 1e6))
    print((M, N, K), " | ".join(row), flush=True)
