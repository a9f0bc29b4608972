"""Site pooling of the compact ProteinCNN output: through the row map (dl_cnn_sitepool_rows_fwd / _bwd) against the
expansion + dense pooling pair (ExpandRowsFn + SitePoolFn), forward and backward, at the default batch's sizes.
HIP-event time per call, one stream.  Run on the GPU box: python tools/sitepool_rows_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops, synthetic                              # noqa: E402
from druglamp_amd.functional import ExpandRowsFn, SitePoolFn, SitePoolRowsFn   # noqa: E402
from druglamp_amd.protein_plan import PlanDev, plan_of                # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    _, meta = synthetic.make_batch(B, "cpu", seed=0, with_graph=False)
    L, S, C = 2304, 9, 128
    plan = plan_of([m["Prot_Len"] for m in meta], L)
    pd = PlanDev(plan, dev)
    pd.fill(plan)
    z = torch.randn(pd.rows, C, device=dev).bfloat16()
    g = torch.randn(B, L // S, C, device=dev).bfloat16()
    fwd_rows = lambda: ops.cnn_sitepool_rows_fwd(z, pd.row_of, B, L, S)
    bwd_rows = lambda: ops.cnn_sitepool_rows_bwd(g, pd.rep, pd.row_of, L, S)
    fwd_exp = lambda: ops.cnn_sitepool_fwd(ops.rows_gather(z, pd.row_of).view(B, L, C), L, 0, S)
    bwd_exp = lambda: ops.rows_sum_strided(ops.cnn_sitepool_bwd(g, L, 0, S).view(B * L, C), pd.rep)
    a, b_ = fwd_rows(), fwd_exp()
    assert torch.equal(a, b_), "forward differs"
    da, db = bwd_rows(), bwd_exp()
    err = (da.float() - db.float()).abs().max().item()
    print("rows %d of %d positions; bwd max |diff| vs expand form %.3g (bf16 rounding of the dense intermediate)" % (pd.rows, B * L, err))
    for name, f in (("fwd through map", fwd_rows), ("fwd expand+pool", fwd_exp), ("bwd through map", bwd_rows), ("bwd pool+sum", bwd_exp)):
        print("%-18s %8.1f us" % (name, timed(f)))


if __name__ == "__main__":
    main()
