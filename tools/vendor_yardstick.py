"""Yardstick only (nothing on the product path calls a vendor GEMM): torch.matmul (hipBLASLt / rocBLAS) against dl_gemm on the
step's plain GEMM shapes, bf16, fp32 accumulate."""
import sys, time, torch
sys.path.insert(0, ".")
from druglamp_amd import ops
dt = torch.bfloat16
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("%-28s%14s%14s" % ("forward shape (M, N, K)", "dl_gemm us", "vendor us"))
for (M, N, K) in [(65536, 512, 2048), (65536, 2048, 512), (65536, 1536, 512), (65536, 256, 1024), (65536, 1024, 256), (65536, 512, 256), (65536, 256, 256), (591864, 128, 1152)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * 0.1).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    a = t(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out))
    b = t(lambda: torch.matmul(x, w.t(), out=out))
    print("%-28s%14.1f%14.1f   (%.0f vs %.0f TFLOP/s)" % (str((M, N, K)), a, b, 2.0 * M * N * K / a / 1e6, 2.0 * M * N * K / b / 1e6), flush=True)
print("%-28s%14s%14s" % ("weight gradient (M, N, K)", "dl_gemm us", "vendor us"))
for (M, N, K) in [(512, 2048, 65536), (2048, 512, 65536), (1024, 256, 65536), (256, 256, 65536), (128, 128, 131072), (128, 1152, 591864)]:
    dy = (torch.randn(K, M, device="cuda") * 0.5).to(dt); x = (torch.randn(K, N, device="cuda") * 0.5).to(dt)
    a = t(lambda: ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0))
    b = t(lambda: torch.matmul(dy.t(), x))
    print("%-28s%14.1f%14.1f   (%.0f vs %.0f TFLOP/s; vendor output bf16, ours fp32)" % (str((M, N, K)), a, b, 2.0 * M * N * K / a / 1e6, 2.0 * M * N * K / b / 1e6), flush=True)
