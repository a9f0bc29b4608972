"""Attention kernels at the path's three shapes: resident vs streaming forms (algo argument of dl_attn_fwd / dl_attn_bwd) — time + max diff."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
dt = torch.bfloat16
# (name, P, H, S, shift, Lq, Lk, hd)
long_only = "--long" in sys.argv
cases = [("pmma paired", 512, 4, 2, 256, 256, 256, 64), ("pmma self", 256, 4, 1, 0, 256, 256, 128), ("pgca", 256, 1, 1, 0, 256, 512, 128),
         ("north-star x-attn (B=256, Ld=64, Lp=512, d=256)", 256, 4, 1, 0, 64, 512, 64), ("same, 2 heads of 128", 256, 2, 1, 0, 64, 512, 128)]
for name, P, H, S, shift, Lq, Lk, hd in ([] if long_only else cases):
    d = H * hd
    qkv = (torch.randn(P * Lq, 3 * d, device=dev) * 0.5).to(dt)
    kv = (torch.randn(P * Lk, 3 * d, device=dev) * 0.5).to(dt)
    q, k, v = qkv[:, :d], kv[:, d:2 * d], kv[:, 2 * d:]
    qs, ks = (Lq * 3 * d, hd, 3 * d), (Lk * 3 * d, hd, 3 * d)
    do = (torch.randn(S, P * Lq, d, device=dev) * 0.1).to(dt)
    res = {}
    for mode in ("0", "1"):
        algo = 1 if mode == "0" else 0          # DL_ATTN_ALGO_STREAM / AUTO
        o = torch.zeros(S, P * Lq, d, device=dev, dtype=dt)
        f = lambda: ops.attn_fwd(q, k, v, n_problems=P, n_heads=H, n_segments=S, partner_shift=shift, Lq=Lq, Lk=Lk, head_dim=hd,
                                 scale=hd ** -0.5, q_strides=qs, k_strides=ks, v_strides=ks, out=o, o_strides=(Lq * d, hd, d), o_ss=P * Lq * d, algo=algo)
        lse = f(); torch.cuda.synchronize()
        tf = timeit(f)
        dq = torch.zeros(P * Lq, d, device=dev, dtype=dt); dk = torch.zeros(P * Lk, d, device=dev, dtype=dt); dv = torch.zeros_like(dk)
        fb = lambda: ops.attn_bwd(q, k, v, o, do, lse, n_problems=P, n_heads=H, n_segments=S, partner_shift=shift, Lq=Lq, Lk=Lk, head_dim=hd,
                                  scale=hd ** -0.5, q_strides=qs, k_strides=ks, v_strides=ks, o_strides=(Lq * d, hd, d), o_ss=P * Lq * d,
                                  do_strides=(Lq * d, hd, d), do_ss=P * Lq * d, dq=dq, dq_strides=(Lq * d, hd, d), dk=dk,
                                  dk_strides=(Lk * d, hd, d), dv=dv, dv_strides=(Lk * d, hd, d), algo=algo)
        fb(); torch.cuda.synchronize()
        tb = timeit(fb)
        res[mode] = (o.clone(), lse.clone(), dq.clone(), dk.clone(), dv.clone(), tf, tb)
    fl = 4.0 * S * P * H * Lq * Lk * hd
    diffs = [float((res["0"][i].float() - res["1"][i].float()).abs().max()) for i in range(5)]
    print("%-12s fwd %.0f -> %.0f us (%.0f -> %.0f TF/s)   bwd %.0f -> %.0f us (%.0f -> %.0f TF/s)   maxdiff o/lse/dq/dk/dv %s" % (
        name, res["0"][5], res["1"][5], fl / res["0"][5] / 1e6, fl / res["1"][5] / 1e6, res["0"][6], res["1"][6],
        2.5 * fl / res["0"][6] / 1e6, 2.5 * fl / res["1"][6] / 1e6, ["%.2g" % x for x in diffs]), flush=True)
