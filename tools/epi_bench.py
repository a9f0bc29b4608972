"""Epilogue cost study for the fat-N / small-K GEMMs: same shape with different epilogues and debug flags."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
dt = torch.bfloat16
for (M, N, K) in [(65536, 2048, 512), (65536, 1024, 256), (65536, 512, 2048)]:
    x = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt)
    b = torch.randn(N, device=dev); pre = torch.empty(M, N, device=dev, dtype=dt); res = torch.randn(M, N, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    r = {}
    r["plain"] = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out))
    r["bias"] = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=b, out=out))
    r["relu"] = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=b, act=2, out=out))
    r["gelu+pre"] = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=b, act=1, pre_out=pre, out=out))
    r["gelu+pre+drop"] = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=b, act=1, pre_out=pre, dropout_p=0.1, seed=5, out=out))
    r["dgelu+drop"] = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, dact_pre=pre, dropout_p=0.1, seed=5, out=out))
    r["res"] = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=b, residual=res, out=out))
    r["copy2x"] = timeit(lambda: (out.copy_(res), pre.copy_(res)))
    print((M, N, K), "  ".join("%s %.0f" % kv for kv in r.items()), flush=True)
