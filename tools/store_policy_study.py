import os, sys, time, torch
os.environ["DL_USE_STUDY_LIB"] = "1"
sys.path.insert(0, "/root/repo")
from druglamp_amd import ops
dt = torch.bfloat16
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
names = ["plain", "sc1", "sc0sc1", "nt", "sc0sc1nt", "sc0"]
print("%-28s" % "shape" + "".join("%10s" % n for n in names), flush=True)
for (M, N, K, kw) in [(65536, 2048, 512, ""), (65536, 1536, 512, "b"), (65536, 512, 2048, ""), (65536, 512, 512, "br"), (65536, 1024, 256, "")]:
    x = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * 0.1).to(dt); b = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(dt); out = torch.empty(M, N, device="cuda", dtype=dt)
    k = dict()
    if "b" in kw: k["bias"] = b
    if "r" in kw: k["residual"] = res
    row = []; ref = None
    for sm in range(6):
        os.environ["DL_GEMM_DBG"] = str(sm << 4)
        row.append(t(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out, **k)))
        if ref is None: ref = out.clone()
        else: assert torch.equal(ref, out), names[sm]
    print("%-28s" % str((M, N, K, kw)) + "".join("%10.1f" % v for v in row), flush=True)
