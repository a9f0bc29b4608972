import sys, time, torch
sys.path.insert(0, ".")
from druglamp_amd import ops
dt = torch.bfloat16
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for M in (8192, 16384):
    for (N, K) in [(256, 1024), (1024, 256), (768, 256), (256, 256), (256, 512), (512, 256)]:
        xs = [(torch.randn(M, K, device="cuda") * 0.5).to(dt) for _ in range(2)]
        ws = [(torch.randn(N, K, device="cuda") * 0.1).to(dt) for _ in range(2)]
        out = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(2)]
        one = t(lambda: ops.gemm(xs[0], ws[0], M=M, N=N, K=K, out=out[0]))
        two = t(lambda: (ops.gemm(xs[0], ws[0], M=M, N=N, K=K, out=out[0]), ops.gemm(xs[1], ws[1], M=M, N=N, K=K, out=out[1])))
        pair = t(lambda: ops.gemm_pair(xs, ws, M=M, N=N, K=K, out=out))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                ops.gemm(xs[0], ws[0], M=M, N=N, K=K, out=out[0]); ops.gemm(xs[1], ws[1], M=M, N=N, K=K, out=out[1])
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            for _ in range(20):
                ops.gemm_pair(xs, ws, M=M, N=N, K=K, out=out)
        tg = t(lambda: g.replay(), 10) / 20; tg2 = t(lambda: g2.replay(), 10) / 20
        print("M=%d N=%d K=%d: one %.1f  two %.1f  pair %.1f us | in a graph: two %.1f  pair %.1f" % (M, N, K, one, two, pair, tg, tg2), flush=True)
