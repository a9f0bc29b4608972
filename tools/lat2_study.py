"""Short-K / narrow-N forward shapes with many tiles: gemm_kernel (two-buffer ring, next tile not requested before the
epilogue) vs the deep-ring 128x128 forms of gemm_big_kernel made eligible for many-tile shapes (study library:
DL_GEMM_LATCFG=3 = 64-byte rows x 4 stages, two workgroups per CU; =1 = 128-byte rows x 4 stages, one per CU)."""
import os, sys, time, torch
os.environ["DL_USE_STUDY_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
dt = torch.bfloat16
shapes = [(131072, 128, 128, "bg"), (131072, 128, 128, ""), (65536, 128, 128, "b"), (131072, 256, 128, "b"), (65536, 128, 256, "b"), (131072, 128, 256, "b"),
          (591864, 128, 1152, "bg"), (591864, 128, 1152, ""), (591867, 128, 768, "bg"), (591870, 128, 384, "bg"), (65536, 128, 648, "bgp")]
for (M, N, K, epi) in shapes:
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt); w = (torch.randn(N, K, device=dev) * 0.1).to(dt); b = torch.randn(N, device=dev)
    kw = dict(bias=b) if "b" in epi else {}
    if "g" in epi and "p" not in epi: kw["act"] = 2                                  # ReLU epilogue
    if "p" in epi: kw.update(act=1, pre_out=torch.empty(M, N, device=dev, dtype=dt))  # GELU + pre-activation copy
    f = lambda: ops.gemm(x, w, M=M, N=N, K=K, **kw)
    line = "%-28s" % ((M, N, K, epi),)
    ref = None
    for cfg, mink, mx in (("0", "512", "256"), ("3", "64", "100000000"), ("1", "64", "100000000")):
        os.environ["DL_GEMM_LATCFG"] = cfg; os.environ["DL_GEMM_LATMINK"] = mink; os.environ["DL_GEMM_LATMAX"] = mx
        out = f().float(); torch.cuda.synchronize()
        if ref is None: ref = out
        line += "  cfg%s %6.1f us (diff %.1e)" % (cfg, timeit(f), (out - ref).abs().max().item())
    print(line, flush=True)
