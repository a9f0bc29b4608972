"""Attention backward at head_dim 64 with Lq, Lk <= 256 (bf16): the one-pass kernel (dl_attn_bwd algo = DL_ATTN_ALGO_ONE_PASS) against the
dQ + dK/dV kernel pair (algo = DL_ATTN_ALGO_TWO_PASS) — time per call (HIP events) and max |difference| of dQ / dK / dV,
at the model's paired shape and a few others.  python tools/attn_bwd_onepass.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops   # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def run(name, P, H, S, shift, Lq, Lk, hd=64):
    d = H * hd
    torch.manual_seed(0)
    qkv = (torch.randn(P * Lq, 3 * d, device=dev) * 0.5).to(dt)
    kv = (torch.randn(P * Lk, 3 * d, device=dev) * 0.5).to(dt)
    q, k, v = qkv[:, :d], kv[:, d:2 * d], kv[:, 2 * d:]
    qs, ks = (Lq * 3 * d, hd, 3 * d), (Lk * 3 * d, hd, 3 * d)
    do = (torch.randn(S, P * Lq, d, device=dev) * 0.1).to(dt)
    o = torch.zeros(S, P * Lq, d, device=dev, dtype=dt)
    lse = ops.attn_fwd(q, k, v, n_problems=P, n_heads=H, n_segments=S, partner_shift=shift, Lq=Lq, Lk=Lk, head_dim=hd, scale=hd ** -0.5,
                       q_strides=qs, k_strides=ks, v_strides=ks, out=o, o_strides=(Lq * d, hd, d), o_ss=P * Lq * d)
    res = {}
    for algo in (3, 2):
        dq = torch.zeros(P * Lq, d, device=dev, dtype=dt)
        dk = torch.zeros(P * Lk, d, device=dev, dtype=dt)
        dv = torch.zeros_like(dk)
        fb = lambda: ops.attn_bwd(q, k, v, o, do, lse, n_problems=P, n_heads=H, n_segments=S, partner_shift=shift, Lq=Lq, Lk=Lk,
                                  head_dim=hd, scale=hd ** -0.5, q_strides=qs, k_strides=ks, v_strides=ks, o_strides=(Lq * d, hd, d),
                                  o_ss=P * Lq * d, do_strides=(Lq * d, hd, d), do_ss=P * Lq * d, dq=dq, dq_strides=(Lq * d, hd, d), dk=dk,
                                  dk_strides=(Lk * d, hd, d), dv=dv, dv_strides=(Lk * d, hd, d), algo=algo)
        t = timed(fb)
        res[algo] = (dq.float(), dk.float(), dv.float(), t)
    fl = 10.0 * S * P * H * Lq * Lk * hd        # algorithmic: S, dP, dV, dK, dQ
    diffs = ["%.2g / %.2g" % (float((res[3][i] - res[2][i]).abs().max()), float(res[2][i].abs().max())) for i in range(3)]
    print("%-28s two-pass %7.1f us  one-pass %7.1f us (%.0f -> %.0f TF/s, five products)   max|diff| / max|ref| dq dk dv: %s" % (
        name, res[2][3], res[3][3], fl / res[2][3] / 1e6, fl / res[3][3] / 1e6, diffs), flush=True)


if __name__ == "__main__":
    Bs = [int(x) for x in sys.argv[1:]] or [256, 32]
    for B in Bs:
        run("pmma paired B=%d" % B, 2 * B, 4, 2, B, 256, 256)
    run("one segment P=256", 256, 4, 1, 0, 256, 256)
    run("ragged 100 x 77, paired", 64, 2, 2, 32, 100, 77)
