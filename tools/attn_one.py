"""One attention shape, forward + backward repeated: target of rocprofv3 --pmc passes (tools/attn_pmc.sh).
usage: attn_one.py self|paired|pgca [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "self"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P, H, S, shift, Lq, Lk, hd = {"paired": (512, 4, 2, 256, 256, 256, 64), "self": (256, 4, 1, 0, 256, 256, 128), "pgca": (256, 1, 1, 0, 256, 512, 128)}[which]
dev, dt = torch.device("cuda:0"), torch.bfloat16
d = H * hd
qkv = (torch.randn(P * Lq, 3 * d, device=dev) * 0.5).to(dt); kv = (torch.randn(P * Lk, 3 * d, device=dev) * 0.5).to(dt)
q, k, v = qkv[:, :d], kv[:, d:2 * d], kv[:, 2 * d:]
qs, ks = (Lq * 3 * d, hd, 3 * d), (Lk * 3 * d, hd, 3 * d)
do = (torch.randn(S, P * Lq, d, device=dev) * 0.1).to(dt)
o = torch.zeros(S, P * Lq, d, device=dev, dtype=dt)
dq = torch.zeros(P * Lq, d, device=dev, dtype=dt); dk = torch.zeros(P * Lk, d, device=dev, dtype=dt); dv = torch.zeros_like(dk)
for _ in range(reps):
    lse = ops.attn_fwd(q, k, v, n_problems=P, n_heads=H, n_segments=S, partner_shift=shift, Lq=Lq, Lk=Lk, head_dim=hd, scale=hd ** -0.5,
                       q_strides=qs, k_strides=ks, v_strides=ks, out=o, o_strides=(Lq * d, hd, d), o_ss=P * Lq * d)
    ops.attn_bwd(q, k, v, o, do, lse, n_problems=P, n_heads=H, n_segments=S, partner_shift=shift, Lq=Lq, Lk=Lk, head_dim=hd, scale=hd ** -0.5,
                 q_strides=qs, k_strides=ks, v_strides=ks, o_strides=(Lq * d, hd, d), o_ss=P * Lq * d, do_strides=(Lq * d, hd, d), do_ss=P * Lq * d,
                 dq=dq, dq_strides=(Lq * d, hd, d), dk=dk, dk_strides=(Lk * d, hd, d), dv=dv, dv_strides=(Lk * d, hd, d))
torch.cuda.synchronize()
