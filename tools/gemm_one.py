"""One GEMM shape, repeated: the target of rocprofv3 --pmc passes (tools/gemm_pmc.sh).  usage: gemm_one.py M N K [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
M, N, K = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm(x, w, M=M, N=N, K=K, out=out)
torch.cuda.synchronize()
