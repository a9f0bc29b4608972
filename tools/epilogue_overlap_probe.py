"""Where the GELU-epilogue GEMM's time goes on the 256x256 tile (one workgroup per CU) and on 256x128 tiles with two
workgroups per CU: full kernel, no stores (DL_GEMM_DBG=1), no operand feed (2), neither (3).  Study library."""
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
print("%-30s %-8s" % ("shape", "tile") + "".join("%12s" % c for c in ("full", "no stores", "no feed", "neither")))
for (M, N, K, kw) in [(65536, 2048, 512, ""), (65536, 2048, 512, "bgp"), (65536, 2048, 512, "bgpd"), (65536, 2048, 512, "G"), (65536, 512, 2048, ""), (65536, 512, 2048, "brd")]:
    x = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * 0.1).to(dt); b = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(dt); pre = torch.empty(M, N, device="cuda", dtype=dt); out = torch.empty(M, N, device="cuda", dtype=dt)
    k = dict()
    if "b" in kw: k["bias"] = b
    if "g" in kw: k["act"] = 1
    if "p" in kw: k["pre_out"] = pre
    if "d" in kw: k.update(dropout_p=0.1, seed=3)
    if "r" in kw: k["residual"] = res
    if "G" in kw: k.update(dact_pre=res, dropout_p=0.1, seed=3)
    for cfg, name in (("0", "256x256"), ("1", "2x256x128")):
        row = []
        for dbg in ("0", "1", "2", "3"):
            os.environ["DL_GEMM_BIGCFG"] = cfg; os.environ["DL_GEMM_DBG"] = dbg
            row.append(t(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out, **k)))
        print("%-30s %-8s" % (str((M, N, K, kw)), name) + "".join("%12.1f" % v for v in row), flush=True)
