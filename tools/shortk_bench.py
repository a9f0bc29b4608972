import torch, time, sys
sys.path.insert(0, ".")
from druglamp_amd import ops
dt=torch.bfloat16
def t(f,n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for (M,N,K,kw) in [(65536,768,256,"b"), (65536,1024,256,"bgpd"), (65536,256,256,"br"), (65536,256,256,""), (65536,512,256,""), (65536,1024,256,"G"), (65536,1024,64,"G"), (65536,512,128,"b")]:
    x=(torch.randn(M,K,device="cuda")*0.5).to(dt); w=(torch.randn(N,K,device="cuda")*0.1).to(dt); b=torch.randn(N,device="cuda")
    res=torch.randn(M,N,device="cuda").to(dt); pre=torch.empty(M,N,device="cuda",dtype=dt); out=torch.empty(M,N,device="cuda",dtype=dt)
    k=dict()
    if "b" in kw: k["bias"]=b
    if "g" in kw: k["act"]=1
    if "p" in kw: k["pre_out"]=pre
    if "d" in kw: k.update(dropout_p=0.1, seed=3)
    if "r" in kw: k["residual"]=res
    if "G" in kw: k.update(dact_pre=res, dropout_p=0.1, seed=3)
    print((M,N,K,kw), "big %.1f us   128-tile %.1f us" % (t(lambda: ops.gemm(x,w,M=M,N=N,K=K,out=out,**k)), t(lambda: ops.gemm(x,w,M=M,N=N,K=K,out=out,algo=1,**k))), flush=True)
