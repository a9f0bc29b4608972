"""One FFN-up GEMM with the GELU + pre-activation + dropout epilogue, repeated: target of rocprofv3 --pmc passes.
usage: gemm_epi_one.py [M N K reps]   (DL_USE_STUDY_LIB / DL_GEMM_BIGCFG select the tile configuration)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
a = [int(v) for v in sys.argv[1:]]
M, N, K = (a + [65536, 2048, 512])[:3] if len(a) >= 3 else (65536, 2048, 512)
reps = a[3] if len(a) > 3 else 10
dt = torch.bfloat16
x = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * 0.1).to(dt); b = torch.randn(N, device="cuda")
pre = torch.empty(M, N, device="cuda", dtype=dt); out = torch.empty(M, N, device="cuda", dtype=dt)
for _ in range(reps):
    ops.gemm(x, w, M=M, N=N, K=K, out=out, bias=b, act=1, pre_out=pre, dropout_p=0.1, seed=3)
torch.cuda.synchronize()
