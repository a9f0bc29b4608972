"""NT-Xent (InfoNCE) kernels at the reference-true shapes: n = B * 512 node rows per half, d = 128
(/root/reference/model/self_supervised_learning.py:35-41,168-182), B = 16 (the reference's batch) and B = 256 (the
metric's batch).  Prints time, TFLOP/s against the dense MFMA peak of the dtype (bf16 2.5 PF, fp32 157 TF) and the
algorithmic HBM bytes against 8 TB/s.  --once: a single forward + backward per shape (for rocprofv3 / PMC passes)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
once = "--once" in sys.argv
dev = torch.device("cuda:0")
def timeit(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
shapes = [(16 * 512, 128), (256 * 512, 128)]
if "--small" in sys.argv: shapes = shapes[:1]
for dt, peak in ((torch.bfloat16, 2500.0), (torch.float32, 157.3)):
    for n, d in shapes:
        if dt == torch.float32 and n > 16 * 512 and "--all" not in sys.argv:
            continue                      # fp32 at B = 256: 17.6 TFLOP on a 157 TF/s pipe (> 0.1 s per pass); --all runs it
        q = (torch.randn(n, d, device=dev) * 0.3).to(dt); k = (torch.randn(n, d, device=dev) * 0.3).to(dt)
        loss, lse, _ = ops.ntxent_fwd_ex(q, k, q, k, 0, 0, n, 0.1)
        reps = 1 if once else (20 if n <= 16 * 512 else 3)
        tf = timeit(lambda: ops.ntxent_fwd_ex(q, k, q, k, 0, 0, n, 0.1), reps)
        tb = timeit(lambda: ops.ntxent_bwd_ex(q, k, q, k, 0, 0, n, 0.1, lse, lse, 1.0 / (2 * n)), reps)
        es = q.element_size()
        flops_f = 2.0 * (2 * n) ** 2 * d; flops_b = 2 * flops_f
        # algorithmic bytes: every row read once as a resident row; the streamed side is re-read by each of the
        # (2n / rows-per-workgroup) workgroups from L2 — the HBM floor is ONE read of both sides + the outputs
        bytes_f = 2 * (2 * n) * d * es + 2 * (2 * n) * 4; bytes_b = 2 * (2 * n) * d * es + (2 * n) * d * 4
        print("%-9s n=%7d d=%d  fwd %9.3f ms %7.1f TF/s (%.3f of MFMA peak; HBM floor %.4f ms = %.5f of the time)   "
              "bwd %9.3f ms %7.1f TF/s (%.3f; HBM floor %.4f ms)   loss %.5f" % (
                  str(dt).replace("torch.", ""), n, d, tf * 1e3, flops_f / tf / 1e12, flops_f / tf / 1e12 / peak,
                  bytes_f / 8e12 * 1e3, bytes_f / 8e12 / tf, tb * 1e3, flops_b / tb / 1e12, flops_b / tb / 1e12 / peak,
                  bytes_b / 8e12 * 1e3, float(loss)), flush=True)
