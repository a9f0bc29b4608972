import os, sys, time, torch
os.environ["DL_USE_STUDY_LIB"] = "1"
sys.path.insert(0, ".")
from druglamp_amd import ops
dt = torch.bfloat16
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("%-28s" % "shape" + "".join("%11s" % c for c in ("full", "no stores", "no feed", "neither", "HBM@5TB/s", "128-tile")))
for (M, N, K, kw) in [(65536, 512, 256, ""), (65536, 768, 256, "b"), (65536, 256, 256, ""), (65536, 256, 512, "b"), (65536, 256, 768, ""), (65536, 256, 1024, ""), (65536, 512, 512, "")]:
    x = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * 0.1).to(dt); b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=dt)
    k = dict(bias=b) if "b" in kw else {}
    row = []
    for dbg in ("0", "1", "2", "3"):
        os.environ["DL_GEMM_DBG"] = dbg
        row.append(t(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out, **k)))
    os.environ["DL_GEMM_DBG"] = "0"
    row.append((M * K + M * N) * 2 / 5.0e6)
    row.append(t(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out, algo=1, **k)))
    print("%-28s" % str((M, N, K, kw)) + "".join("%11.1f" % v for v in row), flush=True)
