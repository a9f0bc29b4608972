"""Few-tile GEMM shapes of the strong-scaling batches (M = 8192 / 16384 / 32768 rows): gemm_kernel's two-buffer ring
(DL_GEMM_LATCFG=0) against the deep-ring 128x128 forms of gemm_big_kernel (1: 128-byte rows x 4 stages, 2: x 3 stages,
3: 64-byte rows x 4 stages, two workgroups per CU).  Study library; prints us per call and the max |diff| vs form 0."""
import os, sys, time, torch
os.environ["DL_USE_STUDY_LIB"] = "1"
sys.path.insert(0, ".")
from druglamp_amd import ops
dt = torch.bfloat16
def t(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
shapes = [(256, 1024, "brd"), (1024, 256, "bgpd"), (1024, 256, "G"), (256, 1024, ""), (768, 256, "b"), (512, 2048, "brd"),
          (2048, 512, "bgpd"), (2048, 512, "G"), (512, 2048, ""), (256, 768, ""), (256, 512, "b"), (512, 1536, ""),
          (256, 256, "br"), (1536, 512, "b"), (512, 512, "br"), (512, 256, ""), (256, 256, ""), (128, 1152, "bg")]
Ms = [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768]
for M in Ms:
    print("M = %d%32s" % (M, "") + "".join("%9s" % ("cfg%d" % c) for c in range(4)) + "   max|diff|")
    for (N, K, kw) in shapes:
        x = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * 0.1).to(dt); b = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda").to(dt); pre = torch.empty(M, N, device="cuda", dtype=dt)
        k = dict()
        if "b" in kw: k["bias"] = b
        if "g" in kw: k["act"] = 1
        if "p" in kw: k["pre_out"] = pre
        if "d" in kw: k.update(dropout_p=0.1, seed=3)
        if "r" in kw: k["residual"] = res
        if "G" in kw: k.update(dact_pre=res, dropout_p=0.1, seed=3)
        row, outs = [], []
        for cfg in range(4):
            os.environ["DL_GEMM_LATCFG"] = str(cfg)
            out = torch.empty(M, N, device="cuda", dtype=dt)
            row.append(t(lambda: ops.gemm(x, w, M=M, N=N, K=K, out=out, **k)))
            outs.append(out.float())
        d = max(float((o - outs[0]).abs().max()) for o in outs[1:])
        print("%-38s" % str((M, N, K, kw)) + "".join("%9.1f" % v for v in row) + "   %.2e" % d, flush=True)
