"""The north-star attention micro-shape (BASELINE.json: PMMA-style cross attention at B=256, Ld=64 query rows, Lp=512 keys,
d=256 = 4 heads of 64), forward and backward, with both rooflines:
  algorithmic flops  4 * B*H * Lq * Lk * hd (fwd), 2.5x that again for bwd (14 * ... in dl_prof's count)
  algorithmic bytes  Q + K + V + O (fwd); Q, K, V, O, dO read + dQ, dK, dV written (bwd); bf16
Run under rocprofv3 --kernel-trace --stats (and the FETCH_SIZE / WRITE_SIZE pmc passes) to get the per-kernel durations
and HBM traffic that go with the host-timed numbers printed here."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops
dev = torch.device("cuda:0")
dt = torch.bfloat16
B, H, Lq, Lk, hd = 256, 4, 64, 512, 64
d = H * hd
q = (torch.randn(B * Lq, d, device=dev) * 0.5).to(dt)
kv = (torch.randn(B * Lk, 2 * d, device=dev) * 0.5).to(dt)
k, v = kv[:, :d], kv[:, d:]
o = torch.zeros(B * Lq, d, device=dev, dtype=dt)
do = (torch.randn(B * Lq, d, device=dev) * 0.1).to(dt)
dq = torch.zeros_like(q); dkv = torch.zeros_like(kv)
qs, ks, os_ = (Lq * d, hd, d), (Lk * 2 * d, hd, 2 * d), (Lq * d, hd, d)
fwd = lambda: ops.attn_fwd(q, k, v, n_problems=B, n_heads=H, n_segments=1, partner_shift=0, Lq=Lq, Lk=Lk, head_dim=hd, scale=hd ** -0.5,
                           q_strides=qs, k_strides=ks, v_strides=ks, out=o, o_strides=os_, o_ss=0)
lse = fwd()
bwd = lambda: ops.attn_bwd(q, k, v, o, do, lse, n_problems=B, n_heads=H, n_segments=1, partner_shift=0, Lq=Lq, Lk=Lk, head_dim=hd,
                           scale=hd ** -0.5, q_strides=qs, k_strides=ks, v_strides=ks, o_strides=os_, o_ss=0, do_strides=os_, do_ss=0,
                           dq=dq, dq_strides=qs, dk=dkv, dk_strides=ks, dv=dkv[:, d:], dv_strides=ks)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
tf, tb = timeit(fwd), timeit(bwd)
fl = 4.0 * B * H * Lq * Lk * hd
by_f = (2 * B * Lq * d + 2 * B * Lk * d) * 2
by_b = (4 * B * Lq * d + 4 * B * Lk * d) * 2
print("north-star attention B=%d Ld=%d Lp=%d d=%d (H=%d x %d), bf16" % (B, Lq, Lk, d, H, hd))
print("  forward : %.1f us  %.0f TFLOP/s (mfma_frac %.3f of 2500)  %.2f TB/s algorithmic (hbm_frac %.3f of 8.0)  [%.1f MB, %.2f GFLOP, %.0f FLOP/B]"
      % (tf, fl / tf / 1e6, fl / tf / 1e6 / 2500, by_f / tf / 1e6, by_f / tf / 1e6 / 8.0, by_f / 1e6, fl / 1e9, fl / by_f))
print("  backward: %.1f us  %.0f TFLOP/s (mfma_frac %.3f)  %.2f TB/s algorithmic (hbm_frac %.3f)  [%.1f MB, %.2f GFLOP]"
      % (tb, 2.5 * fl / tb / 1e6, 2.5 * fl / tb / 1e6 / 2500, by_b / tb / 1e6, by_b / tb / 1e6 / 8.0, by_b / 1e6, 2.5 * fl / 1e9))
# parity of this shape against fp64
qd = q.double().reshape(B, Lq, H, hd).permute(0, 2, 1, 3)[:8]; kd = k.double().reshape(B, Lk, H, hd).permute(0, 2, 1, 3)[:8]
vd = v.double().reshape(B, Lk, H, hd).permute(0, 2, 1, 3)[:8]
ref = torch.softmax(qd @ kd.transpose(-1, -2) * hd ** -0.5, -1) @ vd
got = o.double().reshape(B, Lq, H, hd).permute(0, 2, 1, 3)[:8]
print("  forward max |o - fp64 ref| / max |ref| on 8 problems: %.2e" % float((got - ref).abs().max() / ref.abs().max()))
