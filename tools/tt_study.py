"""Large-tile weight-gradient kernel variants (study library: DL_USE_STUDY_LIB=1, DL_GEMM_TTCFG) — time per variant, with
the stores / the operand feed switched off (DL_GEMM_DBG bits 1 / 2), and the error against the product kernel."""
import os, sys, time, torch
os.environ.setdefault("DL_USE_STUDY_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
dt = torch.bfloat16
shapes = [(2048, 512, 65536), (512, 2048, 65536), (1536, 512, 65536), (128, 1152, 591864), (128, 768, 591867), (1000, 648, 10007)]
if os.environ.get("DL_GEMM_TTWIDE") == "1":
    shapes = [(1024, 256, 65536), (256, 1024, 65536), (768, 256, 65536), (256, 512, 65536), (512, 512, 65536), (256, 256, 65536), (128, 384, 591870), (256, 648, 65536), (256, 392, 131072)]
cfgs = [int(c) for c in os.environ.get("TT_CFGS", "0,1,2,3,4").split(",")]
for (M, N, K) in shapes:
    dy = (torch.randn(K, M, device=dev) * 0.5).to(dt); x = (torch.randn(K, N, device=dev) * 0.5).to(dt)
    db = torch.empty(M, device=dev)
    f = lambda: ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0, x_colsum=db)
    os.environ["DL_GEMM_TTCFG"] = "0"; os.environ["DL_GEMM_DBG"] = "0"
    ref = f().clone(); refb = db.clone(); torch.cuda.synchronize()
    line = "%-22s" % ((M, N, K),)
    for c in cfgs:
        os.environ["DL_GEMM_TTCFG"] = str(c)
        ts = []
        for dbg in (0, 1, 2, 3):
            os.environ["DL_GEMM_DBG"] = str(dbg)
            if dbg == 0:
                out = f().clone(); torch.cuda.synchronize()
                err = (out - ref).abs().max().item(); errb = (db - refb).abs().max().item()
            ts.append(timeit(f))
        os.environ["DL_GEMM_DBG"] = "0"
        print("%s cfg %d: full %6.1f us (%4.0f TF/s)  no-store %6.1f  no-feed %6.1f  neither %6.1f   diff %.1e / %.1e" % (
            line, c, ts[0], 2.0 * M * N * K / ts[0] / 1e6, ts[1], ts[2], ts[3], err, errb), flush=True)
