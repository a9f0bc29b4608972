"""Large-tile weight-gradient path vs the 128-tile split-K path: error vs fp64 reference (sampled) + timing."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
dt = torch.bfloat16
shapes = [(1536, 512, 65536), (1024, 256, 65536), (512, 2048, 65536), (2048, 512, 65536), (768, 256, 65536), (256, 1024, 65536),
          (256, 512, 65536), (512, 512, 65536), (256, 648, 65536), (256, 392, 131072), (128, 1152, 591864), (128, 768, 591867),
          (128, 384, 591870), (256, 256, 65536), (648, 256, 65536), (1000, 200, 10007)]
for (M, N, K) in shapes:
    dy = (torch.randn(K, M, device=dev) * 0.5).to(dt); x = (torch.randn(K, N, device=dev) * 0.5).to(dt)
    outs, ts = {}, {}
    for mode in ("0", "1"):
        f = lambda mode=mode: ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0,
                                       algo=1 if mode == "0" else 0)
        outs[mode] = f().clone(); torch.cuda.synchronize()
        ts[mode] = timeit(f)
    ref = dy[:, :64].double().t() @ x.double()
    scale = ref.abs().max().item()
    e0 = (outs["0"][:64].double() - ref).abs().max().item() / scale; e1 = (outs["1"][:64].double() - ref).abs().max().item() / scale
    d = (outs["0"] - outs["1"]).abs().max().item() / scale
    print("%-22s old %.0f us (%.0f TF/s)  big %.0f us (%.0f TF/s)   err old %.2e big %.2e  diff %.2e" % (
        (M, N, K), ts["0"], 2.0 * M * N * K / ts["0"] / 1e6, ts["1"], 2.0 * M * N * K / ts["1"] / 1e6, e0, e1, d), flush=True)
