"""Weight-gradient GEMMs (both operands K-slow, fp32 output through split-K slabs) at the contraction lengths of small
per-GPU batches: time of dl_gemm (+ its slab reduction) for explicit split counts against the automatic plan."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
dt = torch.bfloat16
Bs = [int(a) for a in sys.argv[1:]] or [32, 64]
for B in Bs:
    K = B * 256
    print("per-GPU batch %d -> contraction K = %d" % (B, K))
    for (M, N) in [(1024, 256), (256, 1024), (768, 256), (256, 256), (256, 512), (2048, 512), (512, 2048), (1536, 512), (512, 512), (128, 128)]:
        dy = (torch.randn(K, M, device=dev) * 0.5).to(dt); x = (torch.randn(K, N, device=dev) * 0.5).to(dt)
        line = "  %5d x %5d:" % (M, N)
        for sp in (0, 1, 2, 4, 8, 16, 32):
            f = lambda sp=sp: ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=sp)
            try:
                line += "  %s %.1f" % ("auto" if sp == 0 else "s%d" % sp, timeit(f))
            except RuntimeError as e:
                line += "  s%d err" % sp
        print(line, flush=True)
