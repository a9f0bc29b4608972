"""Weight-gradient GEMM (both operands K-slow, split-K, fused bias gradient) at the shapes of one training step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
dt = torch.bfloat16
tot = 0
for (M, N, K, cnt) in [(128, 128, 131072, 6), (128, 128, 65536, 4), (256, 128, 131072, 2), (128, 256, 131072, 1), (256, 256, 65536, 5), (256, 512, 65536, 4), (512, 512, 65536, 2), (768, 256, 65536, 4), (1024, 256, 65536, 6), (256, 1024, 65536, 4), (2048, 512, 65536, 2), (512, 2048, 65536, 2)]:
    dy = torch.randn(K, M, device="cuda").to(dt); x = torch.randn(K, N, device="cuda").to(dt); db = torch.empty(M, device="cuda")
    t = timeit(lambda: ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0, x_colsum=db))
    tot += t * cnt
    print((M, N, K), "%.1f us" % t, flush=True)
print("weighted total %.0f us" % tot)
