"""Battery of kernel-vs-torch checks that keeps going after a failure and prints one line per case.
Run on the GPU box:  python tools/gpu_probe.py > gpurun_out/probe.log 2>&1
"""
import math
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
RESULTS = []


def rel_err(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def report(name, err, tol):
    ok = err <= tol and not math.isnan(err)
    RESULTS.append((name, err, tol, ok))
    print("%-70s err=%.3e tol=%.1e %s" % (name, err, tol, "OK" if ok else "FAIL"), flush=True)


def case(fn):
    try:
        fn()
    except Exception:
        print("EXC in", fn.__name__)
        traceback.print_exc()
        RESULTS.append((fn.__name__, float("nan"), 0, False))


def tols(dt):
    return 2e-5 if dt == torch.float32 else 2e-2


# ---------------------------------------------------------------- GEMM
def gemm_cases():
    for dt in (torch.float32, torch.bfloat16):
        for (M, N, K) in [(128, 128, 128), (256, 384, 256), (200, 136, 96), (32, 32, 32), (1000, 520, 264)]:
            x = torch.randn(M, K, device=dev).to(dt)
            w = torch.randn(N, K, device=dev).to(dt) / math.sqrt(K)
            b = torch.randn(N, device=dev)
            ref = x.double() @ w.double().t() + b.double()
            y = ops.gemm(x, w, M=M, N=N, K=K, bias=b)
            report("gemm NT %s %s" % (dt, (M, N, K)), rel_err(y, ref), tols(dt))
            # gelu + pre
            pre = torch.empty(M, N, device=dev, dtype=dt)
            y = ops.gemm(x, w, M=M, N=N, K=K, bias=b, act=1, pre_out=pre)
            report("gemm NT gelu %s %s" % (dt, (M, N, K)), rel_err(y, torch.nn.functional.gelu(ref)), tols(dt))
            report("gemm NT pre %s %s" % (dt, (M, N, K)), rel_err(pre, ref), tols(dt))
            # residual
            r = torch.randn(M, N, device=dev).to(dt)
            y = ops.gemm(x, w, M=M, N=N, K=K, bias=b, residual=r)
            report("gemm NT res %s %s" % (dt, (M, N, K)), rel_err(y, ref + r.double()), tols(dt))
            # dgrad: dX[M,K] = dY[M,N] @ W[N,K]
            dy = torch.randn(M, N, device=dev).to(dt)
            ref_dx = dy.double() @ w.double()
            dx = ops.gemm(dy, w, M=M, N=K, K=N, w_kslow=True, ldw=K)
            report("gemm dgrad %s %s" % (dt, (M, N, K)), rel_err(dx, ref_dx), tols(dt))
            # dgrad with dgelu epilogue
            prek = torch.randn(M, K, device=dev).to(dt)
            pk = prek.double()
            gp = 0.5 * (1 + torch.erf(pk / math.sqrt(2))) + pk * torch.exp(-0.5 * pk * pk) / math.sqrt(2 * math.pi)
            dx = ops.gemm(dy, w, M=M, N=K, K=N, w_kslow=True, ldw=K, dact_pre=prek)
            report("gemm dgrad+dgelu %s %s" % (dt, (M, N, K)), rel_err(dx, ref_dx * gp), tols(dt))
            # wgrad: dW[N,K] = dY^T X   (split-K auto and forced 1)
            ref_dw = dy.double().t() @ x.double()
            for sk in (0, -1, 3):
                dw = ops.gemm(dy, x, M=N, N=K, K=M, x_kslow=True, w_kslow=True, ldx=N, ldw=K,
                              out_dtype=torch.float32, split_k=sk)
                report("gemm wgrad split=%d %s %s" % (sk, dt, (M, N, K)), rel_err(dw, ref_dw), tols(dt))
    # dropout statistics + determinism + row-mod residual
    M, N, K = 512, 256, 64
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    y0 = ops.gemm(x, w, M=M, N=N, K=K)
    y1 = ops.gemm(x, w, M=M, N=N, K=K, dropout_p=0.1, seed=123)
    y2 = ops.gemm(x, w, M=M, N=N, K=K, dropout_p=0.1, seed=123)
    keep = (y1 != 0)
    frac = float(keep.float().mean())
    report("gemm dropout keep-frac (0.9)", abs(frac - 0.9), 5e-3)
    report("gemm dropout determinism", float((y1 - y2).abs().max()), 0.0)
    report("gemm dropout scaling", rel_err(y1[keep], (y0 / 0.9)[keep]), 1e-5)
    y3 = ops.dropout_apply(y0, 0.1, 123)
    report("dropout_apply same mask as gemm epilogue", float((y3 - y1).abs().max()), 1e-5)
    pe = torch.randn(64, N, device=dev)
    y4 = ops.gemm(x, w, M=M, N=N, K=K, residual=pe, res_row_mod=64, res_before_dropout=True)
    report("gemm rowmod residual", rel_err(y4, y0 + pe.repeat(M // 64, 1)), 1e-5)
    # colsum
    for dt in (torch.float32, torch.bfloat16):
        a = torch.randn(1111, 264, device=dev).to(dt)
        report("colsum %s" % dt, rel_err(ops.colsum(a), a.double().sum(0)), 1e-5 if dt == torch.float32 else 1e-5)


# ---------------------------------------------------------------- LayerNorm
def ln_cases():
    for dt in (torch.float32, torch.bfloat16):
        for (M, D) in [(64, 256), (130, 512), (17, 32)]:
            x = torch.randn(M, D, device=dev).to(dt)
            g = torch.randn(D, device=dev)
            b = torch.randn(D, device=dev)
            xr = x.double().requires_grad_(True)
            gr = g.double().requires_grad_(True)
            br = b.double().requires_grad_(True)
            yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
            y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
            report("ln fwd %s %s" % (dt, (M, D)), rel_err(y, yr), tols(dt))
            dy = torch.randn(M, D, device=dev).to(dt)
            dres = torch.randn(M, D, device=dev).to(dt)
            yr.backward(dy.double())
            dx, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, g, dres=dres)
            report("ln bwd dx %s %s" % (dt, (M, D)), rel_err(dx, xr.grad + dres.double()), tols(dt))
            report("ln bwd dgamma %s %s" % (dt, (M, D)), rel_err(dg, gr.grad), tols(dt))
            report("ln bwd dbeta %s %s" % (dt, (M, D)), rel_err(db, br.grad), tols(dt))


# ---------------------------------------------------------------- attention
def attn_ref(q, k, v, scale):
    s = torch.einsum("phqd,phkd->phqk", q, k) * scale
    p = torch.softmax(s, -1)
    return torch.einsum("phqk,phkd->phqd", p, v), s


def attn_cases():
    for dt in (torch.float32, torch.bfloat16):
        for (P, H, Lq, Lk, hd, nseg) in [(2, 4, 256, 256, 64, 1), (2, 2, 100, 77, 64, 1), (3, 1, 256, 512, 128, 1),
                                         (2, 4, 64, 64, 128, 1), (4, 4, 128, 128, 64, 2), (2, 4, 256, 256, 64, 2),
                                         (2, 2, 40, 40, 128, 2),
                                         (3, 4, 64, 512, 64, 1)]:          # north-star cross-attention tile (Ld=64, Lp=512, 4 heads of 64)
            d = H * hd
            scale = 1.0 / math.sqrt(hd)
            # layout [P][L][3d] fused qkv (like the projection GEMM writes it)
            L = max(Lq, Lk)
            qkv = torch.randn(P, L, 3 * d, device=dev).to(dt)
            q = qkv[:, :Lq, 0:d]
            k = qkv[:, :Lk, d:2 * d]
            v = qkv[:, :Lk, 2 * d:3 * d]
            st = (L * 3 * d, hd, 3 * d)
            out = torch.zeros(P, Lq, nseg * d, device=dev, dtype=dt)
            raw = torch.zeros(P, H, Lq, Lk, device=dev) if nseg == 1 else None
            shift = P // 2 if nseg == 2 else 0
            lse = ops.attn_fwd(q, k, v, n_problems=P, n_heads=H, n_segments=nseg, partner_shift=shift, Lq=Lq, Lk=Lk,
                               head_dim=hd, scale=scale, q_strides=st, k_strides=st, v_strides=st, out=out,
                               o_strides=(Lq * nseg * d, hd, nseg * d), o_ss=d, raw_logits=raw)
            qd = q.double().reshape(P, Lq, H, hd).permute(0, 2, 1, 3).requires_grad_(True)
            kd = k.double().reshape(P, Lk, H, hd).permute(0, 2, 1, 3).requires_grad_(True)
            vd = v.double().reshape(P, Lk, H, hd).permute(0, 2, 1, 3).requires_grad_(True)
            o0, s0 = attn_ref(qd, kd, vd, scale)
            refs = [o0]
            if nseg == 2:
                qpart = torch.roll(qd, -shift, 0)  # partner(p) = (p + shift) % P
                o1, _ = attn_ref(qpart, kd, vd, scale)
                refs.append(o1)
            tag = "%s P%d H%d Lq%d Lk%d hd%d seg%d" % (str(dt).split(".")[1], P, H, Lq, Lk, hd, nseg)
            for sidx, oref in enumerate(refs):
                got = out[:, :, sidx * d:(sidx + 1) * d].reshape(P, Lq, H, hd).permute(0, 2, 1, 3)
                report("attn fwd seg%d %s" % (sidx, tag), rel_err(got, oref), tols(dt))
            lse_ref = torch.logsumexp(s0, -1)
            report("attn lse %s" % tag, rel_err(lse[0], lse_ref), 1e-5 if dt == torch.float32 else 1e-2)
            if raw is not None:
                report("attn raw logits %s" % tag, rel_err(raw, s0), 1e-5 if dt == torch.float32 else 2e-2)
            # backward
            do = torch.randn(P, Lq, nseg * d, device=dev).to(dt)
            dqkv = torch.zeros(P, L, 3 * d, device=dev, dtype=dt)
            dq = dqkv[:, :Lq, 0:d]
            dk = dqkv[:, :Lk, d:2 * d]
            dv = dqkv[:, :Lk, 2 * d:3 * d]
            ost = (Lq * nseg * d, hd, nseg * d)
            loss = 0
            for sidx, oref in enumerate(refs):
                dor = do[:, :, sidx * d:(sidx + 1) * d].double().reshape(P, Lq, H, hd).permute(0, 2, 1, 3)
                loss = loss + (oref * dor).sum()
            loss.backward()
            # algo 0 = the library's choice; 3 = DL_ATTN_ALGO_ONE_PASS (these shapes are too small for AUTO to pick the one-pass
            # kernel, which wants a workgroup per CU): dQ, dK, dV from one evaluation of P and dS, both shares of a paired dQ
            for algo in ((0, 3) if dt == torch.bfloat16 and hd == 64 and Lk <= 256 else (0,)):
                dqkv.zero_()
                ops.attn_bwd(q, k, v, out, do, lse, n_problems=P, n_heads=H, n_segments=nseg, partner_shift=shift, Lq=Lq,
                             Lk=Lk, head_dim=hd, scale=scale, q_strides=st, k_strides=st, v_strides=st, o_strides=ost,
                             o_ss=d, do_strides=ost, do_ss=d, dq=dq, dq_strides=st, dk=dk, dk_strides=st, dv=dv,
                             dv_strides=st, algo=algo)
                t2 = tag + (" one-pass" if algo == 3 else "")
                report("attn bwd dq %s" % t2, rel_err(dq.reshape(P, Lq, H, hd).permute(0, 2, 1, 3), qd.grad), tols(dt) * 2)
                report("attn bwd dk %s" % t2, rel_err(dk.reshape(P, Lk, H, hd).permute(0, 2, 1, 3), kd.grad), tols(dt) * 2)
                report("attn bwd dv %s" % t2, rel_err(dv.reshape(P, Lk, H, hd).permute(0, 2, 1, 3), vd.grad), tols(dt) * 2)


# ---------------------------------------------------------------- token gate / elementwise / adamw
def misc_cases():
    for dt in (torch.float32, torch.bfloat16):
        B, L, D, H = 3, 40, 64, 8
        v = torch.randn(B, L, D, device=dev).to(dt)
        logits = torch.randn(B, L, H, device=dev).to(dt)
        vr = v.double().requires_grad_(True)
        lr_ = logits.double().requires_grad_(True)
        attn = torch.softmax(lr_, 1).transpose(1, 2)  # (B, H, L)
        outr = (attn.contiguous().view(B * H, L).unsqueeze(-1) * vr.contiguous().view(B * H, L, D // H)).view(B, L, D) + vr
        out, gate = ops.token_gate_fwd(v, logits, H, True)
        report("token_gate fwd %s" % dt, rel_err(out, outr), tols(dt))
        do = torch.randn(B, L, D, device=dev).to(dt)
        outr.backward(do.double())
        dv, dl = ops.token_gate_bwd(do, v, gate, H, True)
        report("token_gate bwd dv %s" % dt, rel_err(dv, vr.grad), tols(dt))
        report("token_gate bwd dlogits %s" % dt, rel_err(dl, lr_.grad), tols(dt) * 2)
        x = torch.randn(4 * 16, 32, device=dev).to(dt)
        pe = torch.randn(16, 32, device=dev).to(dt)
        y = ops.add_rowmod_dropout(x, pe, 0.0, 0)
        report("add_rowmod %s" % dt, rel_err(y, x.double() + pe.double().repeat(4, 1)), tols(dt))
        report("rowmod_sum %s" % dt, rel_err(ops.rowmod_sum(x, 16), x.double().view(4, 16, 32).sum(0)), 1e-5)
        report("cast roundtrip %s" % dt, rel_err(ops.cast(ops.cast(x, torch.float32), dt), x), 0.0)
    # AdamW vs torch
    n = 1003
    p = torch.randn(n, device=dev)
    pt = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([pt], lr=1e-3)
    m = torch.zeros(n, device=dev)
    v2 = torch.zeros(n, device=dev)
    for step in range(1, 4):
        g = torch.randn(n, device=dev)
        pt.grad = g.clone()
        opt.step()
        ops.adamw_step(p, g, m, v2, lr=1e-3, step=step)
    report("adamw 3 steps vs torch", rel_err(p, pt.detach()), 1e-6)


def loss_cases():
    n, D = 300, 128
    x = torch.randn(n, D, device=dev)
    y = torch.randn(n, D, device=dev)
    xr = x.double().requires_grad_(True)
    lr_ = 2 - 2 * (torch.nn.functional.normalize(xr, dim=-1) * torch.nn.functional.normalize(y.double(), dim=-1)).sum(-1)
    report("cos_rowloss fwd", rel_err(ops.cos_rowloss_fwd(x, y), lr_), 1e-5)
    lr_.mean().backward()
    report("cos_rowloss bwd", rel_err(ops.cos_rowloss_bwd(x, y, 1.0 / n), xr.grad), 1e-5)
    # nt-xent
    for (nn_, d) in [(8, 64), (100, 128), (512, 128)]:
        q = torch.randn(nn_, d, device=dev) * 0.3
        k = torch.randn(nn_, d, device=dev) * 0.3
        qr = q.double().requires_grad_(True)
        kr = k.double().requires_grad_(True)
        projs = torch.cat((qr, kr))
        logits = projs @ projs.t()
        N2 = 2 * nn_
        mask = torch.eye(N2, device=dev).bool()
        logits = logits[~mask].reshape(N2, N2 - 1) / 0.1
        labels = torch.cat((torch.arange(nn_, device=dev) + nn_ - 1, torch.arange(nn_, device=dev)))
        lref = torch.nn.functional.cross_entropy(logits, labels, reduction="sum") / N2
        loss, lse = ops.ntxent_fwd(q, k, 0.1)
        report("ntxent fwd n=%d d=%d" % (nn_, d), rel_err(loss, lref.reshape(1)), 1e-5)
        lref.backward()
        dq, dk = ops.ntxent_bwd(q, k, 0.1, lse, 1.0)
        report("ntxent bwd dq n=%d d=%d" % (nn_, d), rel_err(dq, qr.grad), 1e-4)
        report("ntxent bwd dk n=%d d=%d" % (nn_, d), rel_err(dk, kr.grad), 1e-4)
    # triplet
    n_p, n_d, dim = 20, 29, 256
    P = torch.nn.functional.normalize(torch.randn(n_p, dim, device=dev), dim=-1)
    Dm = torch.nn.functional.normalize(torch.randn(n_d, dim, device=dev), dim=-1)
    gt = torch.randint(0, 2, (n_p, n_d), device=dev).to(torch.int8)
    gt[0] = 0   # anchor with no positives -> anchor-as-positive branch
    gt[1] = 1   # anchor with no negatives -> contributes nothing
    margin = 0.3
    Pr = P.double().requires_grad_(True)
    Dr = Dm.double().requires_grad_(True)
    cosm = torch.nn.functional.cosine_similarity(Pr[:, None, :], Dr[None, :, :], dim=-1)
    dist = 1 - torch.sigmoid(cosm)
    tot, ntri = 0, 0
    for i in range(n_p):
        pos = [j for j in range(n_d) if gt[i, j] == 1]
        neg = [j for j in range(n_d) if gt[i, j] == 0]
        if pos and neg:
            h = dist[i, pos][:, None] - dist[i, neg][None, :] + margin
            tot = tot + torch.clamp(h, min=0).sum()
            ntri += len(pos) * len(neg)
        elif neg:
            dself = 1 - torch.sigmoid(torch.nn.functional.cosine_similarity(Pr[i:i + 1], Pr[i:i + 1]))
            h = dself - dist[i, neg] + margin
            tot = tot + torch.clamp(h, min=0).sum()
            ntri += len(neg)
    lref = tot / max(ntri, 1)
    loss, nt, buf = ops.triplet_sigcos_fwd(P, Dm, gt, margin)
    report("triplet fwd", rel_err(loss, lref.reshape(1)), 1e-5)
    report("triplet n_tri", abs(float(nt) - ntri), 0.0)
    lref.backward()
    dp, dd = ops.triplet_sigcos_bwd(P, Dm, gt, margin, buf, nt, 1.0)
    report("triplet bwd dp", rel_err(dp, Pr.grad), 1e-4)
    report("triplet bwd dd", rel_err(dd, Dr.grad), 1e-4)


# ---------------------------------------------------------------- timing
def bench_cases():
    def timeit(fn, n=20):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    for dt in (torch.bfloat16, torch.float32):
        for (M, N, K) in [(65536, 768, 256), (65536, 1024, 256), (65536, 256, 1024), (65536, 2048, 512), (65536, 512, 2048)]:
            x = torch.randn(M, K, device=dev).to(dt)
            w = torch.randn(N, K, device=dev).to(dt)
            t = timeit(lambda: ops.gemm(x, w, M=M, N=N, K=K))
            tt = timeit(lambda: x @ w.t())
            print("BENCH gemm NT %s %s: %.1f us  %.1f TF/s   (torch matmul %.1f us %.1f TF/s)" %
                  (str(dt).split(".")[1], (M, N, K), t * 1e6, 2 * M * N * K / t / 1e12, tt * 1e6, 2 * M * N * K / tt / 1e12), flush=True)
            dy = torch.randn(M, N, device=dev).to(dt)
            t = timeit(lambda: ops.gemm(dy, w, M=M, N=K, K=N, w_kslow=True, ldw=K))
            print("BENCH gemm dgrad %s %s: %.1f us  %.1f TF/s" % (str(dt).split(".")[1], (M, N, K), t * 1e6, 2 * M * N * K / t / 1e12), flush=True)
            t = timeit(lambda: ops.gemm(dy, x, M=N, N=K, K=M, x_kslow=True, w_kslow=True, ldx=N, ldw=K, out_dtype=torch.float32, split_k=0))
            print("BENCH gemm wgrad %s %s: %.1f us  %.1f TF/s" % (str(dt).split(".")[1], (M, N, K), t * 1e6, 2 * M * N * K / t / 1e12), flush=True)
    for dt in (torch.bfloat16,):
        for (P, H, Lq, Lk, hd, nseg) in [(512, 4, 256, 256, 64, 2), (256, 4, 256, 256, 128, 1), (256, 1, 256, 512, 128, 1)]:
            d = H * hd
            L = max(Lq, Lk)
            qkv = torch.randn(P, L, 3 * d, device=dev).to(dt)
            q, k, v = qkv[:, :Lq, 0:d], qkv[:, :Lk, d:2 * d], qkv[:, :Lk, 2 * d:]
            st = (L * 3 * d, hd, 3 * d)
            out = torch.zeros(P, Lq, nseg * d, device=dev, dtype=dt)
            ost = (Lq * nseg * d, hd, nseg * d)
            kw = dict(n_problems=P, n_heads=H, n_segments=nseg, partner_shift=P // 2 if nseg == 2 else 0, Lq=Lq, Lk=Lk,
                      head_dim=hd, scale=1 / math.sqrt(hd), q_strides=st, k_strides=st, v_strides=st)
            lse = ops.attn_fwd(q, k, v, out=out, o_strides=ost, o_ss=d, **kw)
            t = timeit(lambda: ops.attn_fwd(q, k, v, out=out, o_strides=ost, o_ss=d, **kw))
            fl = 4.0 * nseg * P * H * Lq * Lk * hd
            print("BENCH attn fwd P%d H%d Lq%d Lk%d hd%d seg%d: %.1f us %.1f TF/s" % (P, H, Lq, Lk, hd, nseg, t * 1e6, fl / t / 1e12), flush=True)
            do = torch.randn_like(out)
            dqkv = torch.zeros_like(qkv)
            t = timeit(lambda: ops.attn_bwd(q, k, v, out, do, lse, o_strides=ost, o_ss=d, do_strides=ost, do_ss=d,
                                            dq=dqkv[:, :Lq, 0:d], dq_strides=st, dk=dqkv[:, :Lk, d:2 * d], dk_strides=st,
                                            dv=dqkv[:, :Lk, 2 * d:], dv_strides=st, **kw))
            print("BENCH attn bwd P%d H%d Lq%d Lk%d hd%d seg%d: %.1f us %.1f TF/s" % (P, H, Lq, Lk, hd, nseg, t * 1e6, 2.5 * fl / t / 1e12), flush=True)
    x = torch.randn(65536, 512, device=dev).to(torch.bfloat16)
    g = torch.ones(512, device=dev)
    b = torch.zeros(512, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6))
    print("BENCH ln fwd 65536x512 bf16: %.1f us %.2f TB/s" % (t * 1e6, 2 * x.numel() * 2 / t / 1e12))


if __name__ == "__main__":
    print("device:", torch.cuda.get_device_name(0))
    which = sys.argv[1:] or ["gemm", "ln", "attn", "misc", "loss", "bench"]
    table = {"gemm": gemm_cases, "ln": ln_cases, "attn": attn_cases, "misc": misc_cases, "loss": loss_cases,
             "bench": bench_cases}
    for w in which:
        case(table[w])
    nfail = sum(1 for r in RESULTS if not r[3])
    print("SUMMARY: %d cases, %d failed" % (len(RESULTS), nfail))
    for r in RESULTS:
        if not r[3]:
            print("  FAILED:", r[0], r[1])
