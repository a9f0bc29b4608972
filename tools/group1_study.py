"""Single weight-gradient products: the product path of dl_gemm (128-tile two-buffer kernel with split-K, or the deep-ring tile
where big_tt_plan takes it) against a ONE-member dl_gemm_group (deep-ring 128 x 256 tile, slab count from the group plan).
Times include the second-stage reduction in both cases."""
import os, sys, time, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops, _lib
dt = torch.bfloat16
L = _lib.lib()
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
shapes = [(128, 128, 131072), (256, 256, 65536), (256, 512, 65536), (128, 128, 65536), (256, 128, 131072), (128, 256, 131072), (128, 384, 591870),
          (32, 128, 591872), (8, 1024, 65536), (128, 648, 65536), (648, 128, 65536), (128, 80, 131072), (1024, 256, 65536), (768, 256, 65536),
          (128, 128, 16384), (256, 256, 8192), (8, 1024, 8192), (128, 1152, 73976), (128, 384, 73982)]
print("%-24s %10s %10s %8s" % ("shape", "dl_gemm", "group(1)", "splits"))
for M, N, K in shapes:
    a = (torch.randn(K, M, device="cuda") * 0.5).to(dt); b = (torch.randn(K, N, device="cuda") * 0.5).to(dt)
    db = torch.empty(M, dtype=torch.float32, device="cuda")
    keep = ops.group_wgrad_max_k, ops.group_wgrad_small_mn
    ops.group_wgrad_max_k, ops.group_wgrad_small_mn = 0, 0
    t0 = t(lambda: ops.gemm(a, b, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0, x_colsum=db))
    ref = ops.gemm(a, b, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0, x_colsum=db).clone()
    ops.group_wgrad_max_k, ops.group_wgrad_small_mn, ops.group_wgrad_min_n = 1 << 30, 1 << 30, 1
    def grouped():
        with ops.deferred_reductions():
            return ops.gemm(a, b, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0, x_colsum=db)
    try:
        t1 = t(grouped)
        out = grouped(); torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max())
    except Exception as e:
        t1, err = float("nan"), str(e)[:60]
    ops.group_wgrad_max_k, ops.group_wgrad_small_mn = keep
    print("%-24s %10.1f %10.1f   err %s" % (str((M, N, K)), t0, t1, err if isinstance(err, str) else "%.1e" % err), flush=True)
