"""The north-star attention shape WITH its projections (BASELINE.json north_star: PMMA-style cross attention at B=256,
Ld=64 query rows, Lp=512 key rows, d=256 = 4 heads of 64; SURVEY 8d: the ">= 40 % of MFMA peak" target applies to the block
with the Q / K / V / out projections counted — 564 FLOP per byte — not to the bare core, which is HBM-bound at 57 FLOP/B):

    forward   nq = LN(xd), nkv = LN(xp);  q = nq Wq^T + bq;  [k | v] = nkv [Wk; Wv]^T + b;  a = softmax(q k^T / 8) v per head;
              y = a Wo^T + bo + xd                                    (block.py:33-62 / attention.py:90-127 with Lq != Lk)
    backward  every data and weight gradient of the above (dxd, dxp, dWq, dWkv, dWo, biases, LayerNorm parameters)

built from the SAME entry points the model's blocks use (dl_layernorm_*, dl_gemm, dl_attn_fwd / _bwd through druglamp_amd.ops),
one launch per product — there is no projection-fused attention kernel in this library, and this file says what that costs.

    python3 tools/northstar_block.py            host-timed fwd / bwd, both rooflines, parity of the bf16 and fp32 runs vs fp64 torch
    python3 tools/northstar_block.py --once     a few passes only (for rocprofv3 --kernel-trace --stats / --pmc runs)

Algorithmic work per batch of 256 pairs: forward 47.2 GFLOP (projections 38.7 + core 8.6), backward 98.8 GFLOP (2 x the
projection flops + 2.5 x the core: five tile products against the forward's two); algorithmic bytes = the tensors that MUST
cross HBM: xd, xp read, y written (forward); xd, xp, dy read, dxd, dxp written + the saved activations the backward re-reads as
this library keeps them (nq, nkv, q, kv, a, lse): stated per direction in the output."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops  # noqa: E402

B, H, LQ, LK, HD = 256, 4, 64, 512, 64
D = H * HD
EPS = 1e-6


def make_params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)                        # noqa: E731
    p = {"ln_q.w": 1 + 0.1 * r(D), "ln_q.b": 0.1 * r(D), "ln_kv.w": 1 + 0.1 * r(D), "ln_kv.b": 0.1 * r(D),
         "wq": r(D, D) / D ** 0.5, "bq": 0.1 * r(D), "wkv": r(2 * D, D) / D ** 0.5, "bkv": 0.1 * r(2 * D),
         "wo": r(D, D) / D ** 0.5, "bo": 0.1 * r(D)}
    return {k: v.to(dev) for k, v in p.items()}


def make_inputs(dev, nb=B, seed=1):
    g = torch.Generator().manual_seed(seed)
    xd = torch.randn(nb * LQ, D, generator=g).to(dev)
    xp = torch.randn(nb * LK, D, generator=g).to(dev)
    dy = (torch.randn(nb * LQ, D, generator=g) * 0.1).to(dev)
    return xd, xp, dy


class Block:
    """The block on the HIP entry points, compute dtype `dt` (bf16 or fp32); weights are fp32 masters, images made once."""

    def __init__(self, params, dt, nb=B):
        self.dt, self.nb = dt, nb
        c = lambda t: t.to(dt).contiguous()                             # noqa: E731
        self.p = params
        self.wq, self.wkv, self.wo = c(params["wq"]), c(params["wkv"]), c(params["wo"])
        self.wqT, self.wkvT, self.woT = c(params["wq"].t()), c(params["wkv"].t()), c(params["wo"].t())   # [in][out]: K-contiguous data gradients

    def forward(self, xd, xp):
        nb, p = self.nb, self.p
        Mq, Mk = nb * LQ, nb * LK
        self.xd, self.xp = xd, xp
        self.nq, self.mq, self.rq = ops.layernorm_fwd(xd, p["ln_q.w"], p["ln_q.b"], EPS)
        self.nkv, self.mk, self.rk = ops.layernorm_fwd(xp, p["ln_kv.w"], p["ln_kv.b"], EPS)
        self.q = ops.gemm(self.nq, self.wq, M=Mq, N=D, K=D, bias=p["bq"])
        self.kv = ops.gemm(self.nkv, self.wkv, M=Mk, N=2 * D, K=D, bias=p["bkv"])
        self.a = torch.empty((Mq, D), dtype=self.dt, device=xd.device)
        self.qs, self.ks, self.os = (LQ * D, HD, D), (LK * 2 * D, HD, 2 * D), (LQ * D, HD, D)
        self.lse = ops.attn_fwd(self.q, self.kv, self.kv[:, D:], n_problems=nb, n_heads=H, n_segments=1, partner_shift=0, Lq=LQ, Lk=LK,
                                head_dim=HD, scale=HD ** -0.5, q_strides=self.qs, k_strides=self.ks, v_strides=self.ks, out=self.a,
                                o_strides=self.os, o_ss=0)
        return ops.gemm(self.a, self.wo, M=Mq, N=D, K=D, bias=p["bo"], residual=xd)

    def backward(self, dy):
        nb, p = self.nb, self.p
        Mq, Mk = nb * LQ, nb * LK
        f32 = torch.float32
        g = {}
        g["bo"] = torch.empty(D, dtype=f32, device=dy.device)
        g["wo"] = ops.gemm(dy, self.a, M=D, N=D, K=Mq, x_kslow=True, w_kslow=True, ldx=D, ldw=D, out_dtype=f32, split_k=0, x_colsum=g["bo"])
        da = ops.gemm(dy, self.woT, M=Mq, N=D, K=D)
        dq = torch.empty_like(self.q)
        dkv = torch.empty_like(self.kv)
        ops.attn_bwd(self.q, self.kv, self.kv[:, D:], self.a, da, self.lse, n_problems=nb, n_heads=H, n_segments=1, partner_shift=0,
                     Lq=LQ, Lk=LK, head_dim=HD, scale=HD ** -0.5, q_strides=self.qs, k_strides=self.ks, v_strides=self.ks,
                     o_strides=self.os, o_ss=0, do_strides=self.os, do_ss=0, dq=dq, dq_strides=self.qs, dk=dkv, dk_strides=self.ks,
                     dv=dkv[:, D:], dv_strides=self.ks)
        g["bq"] = torch.empty(D, dtype=f32, device=dy.device)
        g["wq"] = ops.gemm(dq, self.nq, M=D, N=D, K=Mq, x_kslow=True, w_kslow=True, ldx=D, ldw=D, out_dtype=f32, split_k=0, x_colsum=g["bq"])
        g["bkv"] = torch.empty(2 * D, dtype=f32, device=dy.device)
        g["wkv"] = ops.gemm(dkv, self.nkv, M=2 * D, N=D, K=Mk, x_kslow=True, w_kslow=True, ldx=2 * D, ldw=D, out_dtype=f32, split_k=0,
                            x_colsum=g["bkv"])
        dnq = ops.gemm(dq, self.wqT, M=Mq, N=D, K=D)
        dnkv = ops.gemm(dkv, self.wkvT, M=Mk, N=D, K=2 * D)
        g["xd"], g["ln_q.w"], g["ln_q.b"] = ops.layernorm_bwd(dnq, self.xd, self.mq, self.rq, p["ln_q.w"], dres=dy)      # + the residual path
        g["xp"], g["ln_kv.w"], g["ln_kv.b"] = ops.layernorm_bwd(dnkv, self.xp, self.mk, self.rk, p["ln_kv.w"])
        return g


def torch_reference(params, xd, xp, dy, nb, dtype=torch.float64):
    """The same block in plain torch at `dtype` (autograd): y and the gradient dict."""
    import torch.nn.functional as F
    p = {k: v.detach().to(dtype).requires_grad_(True) for k, v in params.items()}
    xd_, xp_ = xd.detach().to(dtype).requires_grad_(True), xp.detach().to(dtype).requires_grad_(True)
    nq = F.layer_norm(xd_, (D,), p["ln_q.w"], p["ln_q.b"], EPS)
    nkv = F.layer_norm(xp_, (D,), p["ln_kv.w"], p["ln_kv.b"], EPS)
    q = F.linear(nq, p["wq"], p["bq"]).view(nb, LQ, H, HD).permute(0, 2, 1, 3)
    kv = F.linear(nkv, p["wkv"], p["bkv"])
    k = kv[:, :D].reshape(nb, LK, H, HD).permute(0, 2, 1, 3)
    v = kv[:, D:].reshape(nb, LK, H, HD).permute(0, 2, 1, 3)
    a = torch.softmax(q @ k.transpose(-1, -2) * HD ** -0.5, -1) @ v
    y = F.linear(a.permute(0, 2, 1, 3).reshape(nb * LQ, D), p["wo"], p["bo"]) + xd_
    (y * dy.to(dtype)).sum().backward()
    g = {k: v.grad for k, v in p.items()}
    g["xd"], g["xp"] = xd_.grad, xp_.grad
    return y.detach(), g


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    dev = torch.device("cuda:0")
    once = "--once" in sys.argv
    params = make_params(dev)
    # parity first (8 pairs, every gradient): bf16 at the bf16 tolerance, fp32 at the north-star 1e-4
    xd8, xp8, dy8 = make_inputs(dev, 8)
    yr, gr = torch_reference(params, xd8, xp8, dy8, 8)
    for dt, tol in ((torch.float32, 1e-4), (torch.bfloat16, 3e-2)):
        blk = Block(params, dt, 8)
        y = blk.forward(xd8.to(dt), xp8.to(dt))
        g = blk.backward(dy8.to(dt))
        worst = max([("y", _rel(y, yr))] + [(k, _rel(g[k], gr[k])) for k in gr], key=lambda t: t[1])
        print("parity %-8s vs fp64 torch on 8 pairs: y %.2e, worst gradient %s %.2e (tolerance %.0e)%s"
              % (str(dt).replace("torch.", ""), _rel(y, yr), worst[0], worst[1], tol, "" if worst[1] <= 3 * tol else "   <-- ABOVE TOLERANCE"))
    dt = torch.bfloat16
    blk = Block(params, dt, B)
    xd, xp, dy = (t.to(dt) for t in make_inputs(dev, B))

    def timeit(fn, n):
        for _ in range(1 if once else 5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    n = 3 if once else 50
    tf = timeit(lambda: blk.forward(xd, xp), n)
    tb = timeit(lambda: blk.backward(dy), n)
    proj = 2.0 * B * (LQ * D * D + LK * D * 2 * D + LQ * D * D)
    core = 4.0 * B * H * LQ * LK * HD
    fl_f, fl_b = proj + core, 2 * proj + 2.5 * core
    es = 2
    by_f_min = (B * LQ * D * 2 + B * LK * D) * es                       # xd, xp in; y out
    by_f_lib = by_f_min + (B * LQ * D * 3 + B * LK * D * 3) * es * 2 - 0   # + nq, q, a and nkv, kv written then read (one round trip each)
    by_b_min = (B * LQ * D * 3 + B * LK * D * 2) * es                   # xd, xp, dy in; dxd, dxp out
    by_b_lib = by_b_min + (B * LQ * D * 3 + B * LK * D * 3) * es + (B * LQ * D * 3 + B * LK * D * 3) * es * 2   # saved nq, q, a, nkv, kv read; da, dq, dnq, dkv, dnkv round trips
    print("north-star BLOCK B=%d Ld=%d Lp=%d d=%d (H=%d x %d), bf16, projections counted" % (B, LQ, LK, D, H, HD))
    for name, t, fl, bmin, blib in (("forward ", tf, fl_f, by_f_min, by_f_lib), ("backward", tb, fl_b, by_b_min, by_b_lib),
                                    ("fwd+bwd ", tf + tb, fl_f + fl_b, by_f_min + by_b_min, by_f_lib + by_b_lib)):
        print("  %s: %7.1f us  %6.0f TFLOP/s  mfma_frac %.3f of 2500  | algorithmic bytes %.0f MB (%.0f FLOP/B) -> hbm_frac %.3f of 8 TB/s; "
              "as this library stores its intermediates %.0f MB -> %.2f TB/s = %.3f"
              % (name, t, fl / t / 1e6, fl / t / 1e6 / 2500, bmin / 1e6, fl / bmin, bmin / t / 1e6 / 8.0, blib / 1e6, blib / t / 1e6, blib / t / 1e6 / 8.0))
    print("  [%.1f GFLOP forward (projections %.1f + core %.1f), %.1f GFLOP backward]" % (fl_f / 1e9, proj / 1e9, core / 1e9, fl_b / 1e9))


if __name__ == "__main__":
    main()
