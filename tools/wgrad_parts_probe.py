"""Small-output weight gradients on the 128-tile split-K kernel: full launch, no slab stores (DL_GEMM_DBG=1), no operand
feed after the first stage (2), neither (3); the reduction launch is not part of the timings (deferred).  Study library."""
import os, sys, time, torch
os.environ["DL_USE_STUDY_LIB"] = "1"
sys.path.insert(0, ".")
from druglamp_amd import ops
dt = torch.bfloat16
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("%-24s" % "shape" + "".join("%12s" % c for c in ("full", "no stores", "no feed", "neither", "HBM floor")))
for (M, N, K) in [(1024, 256, 65536), (256, 1024, 65536), (768, 256, 65536), (256, 512, 65536), (256, 256, 65536), (128, 128, 131072), (512, 512, 65536)]:
    dy = (torch.randn(K, M, device="cuda") * 0.5).to(dt); x = (torch.randn(K, N, device="cuda") * 0.5).to(dt)
    row = []
    for dbg in ("0", "1", "2", "3"):
        os.environ["DL_GEMM_DBG"] = dbg
        def f():
            with ops.deferred_reductions():
                ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0)
                ops._pending.clear()            # timing study: the reduction is dropped
        row.append(t(f))
    print("%-24s" % str((M, N, K)) + "".join("%12.1f" % v for v in row) + "%12.1f" % ((M + N) * K * 2 / 5.0e6), flush=True)
