"""One GEMM shape, repeated: the target of rocprofv3 --pmc passes (tools/gemm_pmc.sh).
usage: gemm_one.py M N K [reps] [tt]   (tt: weight-gradient layout, both operands K-slow, fp32 out, split-K)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
M, N, K = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
tt = len(sys.argv) > 5 and sys.argv[5] == "tt"
if tt:
    x = torch.randn(K, M, device=dev).to(torch.bfloat16); w = torch.randn(K, N, device=dev).to(torch.bfloat16)
    for _ in range(reps):
        ops.gemm(x, w, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0)
else:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(reps):
        ops.gemm(x, w, M=M, N=N, K=K, out=out)
torch.cuda.synchronize()
