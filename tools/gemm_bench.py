import os, sys, time, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
shapes = [(65536, 768, 256), (65536, 256, 256), (65536, 1024, 256), (65536, 256, 1024), (65536, 2048, 512), (65536, 512, 2048), (65536, 512, 512), (589824, 128, 384)]
dt = torch.bfloat16
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt)
    t = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K))
    by = (M * K + M * N) * 2
    line = "NT %s: %.1f us %.0f TF/s %.2f TB/s(alg)" % ((M, N, K), t * 1e6, 2 * M * N * K / t / 1e12, by / t / 1e12)
    if len(sys.argv) > 1:
        tt = timeit(lambda: x @ w.t()); line += "  | hipblaslt %.1f us" % (tt * 1e6)
    dy = torch.randn(M, N, device=dev).to(dt)
    t = timeit(lambda: ops.gemm(dy, w, M=M, N=K, K=N, w_kslow=True, ldw=K))
    line += " | dgrad %.1f us %.0f TF/s" % (t * 1e6, 2 * M * N * K / t / 1e12)
    t = timeit(lambda: ops.gemm(dy, x, M=N, N=K, K=M, x_kslow=True, w_kslow=True, ldx=N, ldw=K, out_dtype=torch.float32, split_k=0))
    line += " | wgrad %.1f us %.0f TF/s" % (t * 1e6, 2 * M * N * K / t / 1e12)
    print(line, flush=True)
