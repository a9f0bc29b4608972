"""Bitwise repeatability of kernels while a second process hammers the same GPU (races on counted waits / LDS rings show up
as rare mismatching launches only under contention).  usage: contention_repeat.py [reps]
CR_LOAD=process (default): the load is a second PROCESS (its queues are time-sliced against ours: waves are saved / restored);
CR_LOAD=thread: the same load from a second thread of THIS process on its own stream (kernels of one process share the chip
without queue time-slicing); CR_LOAD=none: idle GPU.  Round 6 uses the three to ask what the round-3 LayerNorm-backward fault
(stale lanes 48-63 with the SLP-vectorised build, only under load) depends on."""
import os, sys, time, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROLE = os.environ.get("CR_ROLE", "main")
from druglamp_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(1 if ROLE == "main" else 2)
dt = torch.bfloat16
def tt(M, N, K):
    dy = (torch.randn(K, M, device=dev) * 0.5).to(dt); x = (torch.randn(K, N, device=dev) * 0.5).to(dt); db = torch.empty(M, device=dev)
    return lambda: (ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0, x_colsum=db), db.clone())
def nn(M, N, K):
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt); w = (torch.randn(N, K, device=dev) * 0.1).to(dt); b = torch.randn(N, device=dev)
    return lambda: (ops.gemm(x, w, M=M, N=N, K=K, bias=b),)
def ln(M, D):
    x = torch.randn(M, D, device=dev).to(dt); dy = torch.randn(M, D, device=dev).to(dt); g = torch.randn(D, device=dev); b = torch.randn(D, device=dev)
    def f():
        y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
        dx, dg, dbb = ops.layernorm_bwd(dy, x, mean, rstd, g, dres=dy)
        return y, dx, dg, dbb, mean, rstd
    return f
only_ln = os.environ.get("CR_ONLY_LN") == "1"
def nn_gelu(M, N, K):
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt); w = (torch.randn(N, K, device=dev) * 0.1).to(dt); b = torch.randn(N, device=dev)
    pre = torch.empty(M, N, device=dev, dtype=dt)
    return lambda: (ops.gemm(x, w, M=M, N=N, K=K, bias=b, act=1, pre_out=pre, dropout_p=0.1, seed=7), pre.clone())
def nn_dgelu(M, N, K):
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt); w = (torch.randn(N, K, device=dev) * 0.1).to(dt)
    pre = torch.randn(M, N, device=dev).to(dt)
    return lambda: (ops.gemm(x, w, M=M, N=N, K=K, dact_pre=pre, dropout_p=0.1, seed=7),)
def nn_res(M, N, K):
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt); w = (torch.randn(N, K, device=dev) * 0.1).to(dt); b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(dt)
    return lambda: (ops.gemm(x, w, M=M, N=N, K=K, bias=b, residual=r, dropout_p=0.1, seed=9),)
def ntx(n, d, dtype):
    q = (torch.randn(n, d, device=dev) * 0.3).to(dtype); k = (torch.randn(n, d, device=dev) * 0.3).to(dtype)
    def f():
        loss, lse, rl = ops.ntxent_fwd_ex(q, k, q, k, 0, 0, n, 0.1)
        dq, dk = ops.ntxent_bwd_ex(q, k, q, k, 0, 0, n, 0.1, lse, lse, 1.0 / (2 * n))
        return loss, lse, dq, dk
    return f
def misc():
    x = torch.randn(256, 2304, 640, device=dev).to(dt)
    h = torch.randn(591872, 128, device=dev).to(dt)
    def f():
        outs = list(ops.fill_pool(x, 9, dt))
        outs.append(ops.dropout_apply(h, 0.1, 11))
        return tuple(outs)
    return f
def attn_bwd(P, H, S, L, algo):
    hd = 64; d = H * hd; shift = P // 2 if S == 2 else 0
    qkv = (torch.randn(P * L, 3 * d, device=dev) * 0.5).to(dt)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    st = (L * 3 * d, hd, 3 * d)
    do = (torch.randn(S, P * L, d, device=dev) * 0.1).to(dt)
    o = torch.zeros(S, P * L, d, device=dev, dtype=dt)
    kw = dict(n_problems=P, n_heads=H, n_segments=S, partner_shift=shift, Lq=L, Lk=L, head_dim=hd, scale=hd ** -0.5, q_strides=st, k_strides=st, v_strides=st)
    lse = ops.attn_fwd(q, k, v, out=o, o_strides=(L * d, hd, d), o_ss=P * L * d, **kw)
    def f():
        dqkv = torch.empty_like(qkv)
        ops.attn_bwd(q, k, v, o, do, lse, o_strides=(L * d, hd, d), o_ss=P * L * d, do_strides=(L * d, hd, d), do_ss=P * L * d, dq=dqkv, dq_strides=st,
                     dk=dqkv[:, d:], dk_strides=st, dv=dqkv[:, 2 * d:], dv_strides=st, algo=algo, **kw)
        return (dqkv,)
    return f
def bn(R, C, weighted):
    y = torch.relu(torch.randn(R, C, device=dev) * 0.3 + 0.1).to(dt); dz = (torch.randn(R, C, device=dev) * 1e-3).to(dt)
    rw = torch.randint(-1, 4, (R,), device=dev).float() if weighted else None
    n = int(rw.clamp(min=0).sum().item()) if weighted else R
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    def f():
        mean, var, rstd = ops.bn_stats_finalize(y, 0, 0, 0, n, 1e-5, 0.0, None, None, rw)
        z = ops.bn_apply_fwd(y, mean, rstd, gam, bet, 0, 0, 0, rw)
        s2 = ops.bn_bwd_reduce(dz, y, mean, rstd, 0, 0, 0, rw)
        dy = ops.bn_bwd_apply(dz, y, mean, rstd, gam, s2, 1.0 / n, True, 0, 0, 0, rw)
        return mean, var, rstd, z, s2, dy
    return f
def rows_pool(B):
    from druglamp_amd import synthetic
    from druglamp_amd.protein_plan import PlanDev, plan_of
    _, meta = synthetic.make_batch(B, "cpu", seed=0, with_graph=False)
    plan = plan_of([m["Prot_Len"] for m in meta], 2304)
    pd = PlanDev(plan, dev); pd.fill(plan)
    z = torch.randn(pd.rows, 128, device=dev).to(dt); g = torch.randn(B, 256, 128, device=dev).to(dt)
    x32 = torch.randn(4096, 640, device=dev)
    def run():
        ops.protein_plan_build(pd)                       # (round 5: the row tables themselves are built on the device)
        return (pd.buf.clone(), ops.cnn_sitepool_rows_fwd(z, pd.row_of, B, 2304, 9), ops.cnn_sitepool_rows_bwd(g, pd.rep, pd.row_of, 2304, 9),
                ops.rows_gather(z, pd.row_of), ops.cast(x32, dt))
    return run
def make_cases():
  return {"BatchNorm stats+finalize / apply / bwd reduce / bwd apply 300000x128 (wide kernels)": bn(300000, 128, False),
         "BatchNorm with row weights 165888x128": bn(165888, 128, True), "BatchNorm 70000x96 (generic kernels)": bn(70000, 96, False),
         "row tables built on the device, site pooling through the row map fwd / bwd, rows_gather, cast": rows_pool(128),
         "attention backward one-pass, paired 384 x 4 x 256^2 (LDS-DMA ring, counted waits)": attn_bwd(384, 4, 2, 256, 3),
         "attention backward one-pass, one segment 300 x 4 x 200^2": attn_bwd(300, 4, 1, 200, 3),
         "tt 2048x512x65536": tt(2048, 512, 65536), "tt 1024x256x65536": tt(1024, 256, 65536), "tt 128x768x591867": tt(128, 768, 591867),
         "tt 256x256x65536 (128-tile)": tt(256, 256, 65536), "nn 65536x512x2048": nn(65536, 512, 2048), "nn 65536x768x256": nn(65536, 768, 256),
         "nn gelu+pre+dropout 65536x2048x512": nn_gelu(65536, 2048, 512), "nn gelu'(pre)+dropout 65536x2048x512": nn_dgelu(65536, 2048, 512),
         "nn bias+dropout+residual 65536x512x2048": nn_res(65536, 512, 2048), "nn gelu 128-tile 65536x128x648": nn_gelu(65536, 128, 648),
         "fill_pool + dropout_apply": misc(), "ntxent bf16 n=8192": ntx(8192, 128, dt), "ntxent f32 n=2048": ntx(2048, 128, torch.float32),
         "ln 65536x256": ln(65536, 256), "ln 65536x512": ln(65536, 512), "ln 65536x384 (8-byte kernels)": ln(65536, 384)}
cases = make_cases()
if only_ln and ROLE == "main":
    cases = {k: v for k, v in cases.items() if k.startswith("ln")}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
if ROLE == "load":
    t_end = time.time() + float(os.environ.get("CR_SECONDS", "60"))
    fs = list(cases.values())
    while time.time() < t_end:
        for f in fs: f()
        torch.cuda.synchronize()
    sys.exit(0)
LOAD = os.environ.get("CR_LOAD", "process")
p = None
stop = [False]
if LOAD == "process":
    env = dict(os.environ, CR_ROLE="load", CR_SECONDS="90")
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env)
    time.sleep(8)
elif LOAD == "thread":
    import threading
    lcases = list(make_cases().values())           # the loader's own tensors
    side = torch.cuda.Stream()
    def loader():
        with torch.cuda.stream(side):
            while not stop[0]:
                for f in lcases:
                    if stop[0]: break
                    f()
                side.synchronize()
    th = threading.Thread(target=loader, daemon=True); th.start()
    time.sleep(3)
print("load:", LOAD, flush=True)
for name, f in cases.items():
    ref = [t.clone() for t in f()]; torch.cuda.synchronize()
    bad = 0
    for _ in range(reps):
        out = f(); torch.cuda.synchronize()
        eq = [torch.equal(a, b) for a, b in zip(out, ref)]
        bad += int(not all(eq))
        if not all(eq) and bad <= 2:
            for i, (a, b) in enumerate(zip(out, ref)):
                if not eq[i]:
                    d = (a.float() - b.float()).abs()
                    print("    output %d: %d elements differ, max |diff| %.3e, first at flat index %d" % (
                        i, int((d > 0).sum()), float(d.max()), int((d.flatten() > 0).nonzero()[0])), flush=True)
                    if d.dim() == 2:
                        rows = (d > 0).any(1).nonzero().flatten().tolist()
                        print("      rows:", rows[:24], " columns of the first row:", (d[rows[0]] > 0).nonzero().flatten().tolist()[:40], flush=True)
                    else:
                        print("      columns:", (d > 0).nonzero().flatten().tolist(), flush=True)
    print("%-32s mismatching launches %d / %d" % (name, bad, reps), flush=True)
stop[0] = True
if p is not None:
    p.terminate(); p.wait()
