"""Large-tile GEMM path vs the 128x128 path: bitwise comparison per epilogue + timing."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
dt = torch.bfloat16
shapes = [(65536, 2048, 512), (65536, 1024, 256), (65536, 512, 2048), (65536, 256, 1024), (65536, 768, 256), (591864, 128, 1152),
          (131072, 128, 128), (65536, 648, 128), (65536, 256, 256), (65536, 1536, 512), (65000, 520, 96)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (M, N, K) in shapes:
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt); w = (torch.randn(N, K, device=dev) * 0.1).to(dt)
    b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).to(dt)
    pre_in = torch.randn(M, N, device=dev).to(dt)
    cases = {
        "plain": dict(), "bias": dict(bias=b), "relu": dict(bias=b, act=2),
        "gelu+pre+drop": dict(bias=b, act=1, pre_out=True, dropout_p=0.1, seed=5),
        "dgelu+drop": dict(dact_pre=pre_in, dropout_p=0.1, seed=5),
        "res+drop": dict(bias=b, residual=res, dropout_p=0.1, seed=7),
    }
    line = "%s" % ((M, N, K),)
    for name, kw in cases.items():
        outs, ts = {}, {}
        for mode in ("0", "1"):
            k2 = dict(kw, algo=1 if mode == "0" else 0)
            pre = None
            if k2.get("pre_out"):
                pre = torch.zeros(M, N, device=dev, dtype=dt); k2["pre_out"] = pre
            out = torch.zeros(M, N, device=dev, dtype=dt)
            ops.gemm(x, w, M=M, N=N, K=K, out=out, **k2)
            torch.cuda.synchronize()
            outs[mode] = (out.clone(), None if pre is None else pre.clone())
            ts[mode] = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out, **k2), 10)
        ok = all(torch.equal(outs["0"][0], outs[m][0]) and (outs["0"][1] is None or torch.equal(outs["0"][1], outs[m][1])) for m in ("1",))
        if not ok:
            d1 = (outs["0"][0].float() - outs["1"][0].float()).abs().max().item()
            line += "  %s MISMATCH(%.3g)" % (name, d1)
        line += "  %s %.0f/%.0f" % (name, ts["0"], ts["1"])
    print(line, flush=True)
