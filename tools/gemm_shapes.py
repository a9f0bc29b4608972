"""Per-shape GEMM table for one training step: wraps ops.gemm with HIP-event timing (serialised) and
aggregates by (M, N, K, layout, epilogue).  usage: python tools/gemm_shapes.py [--batch 256]"""
import argparse
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from druglamp_amd import ops  # noqa: E402
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into  # noqa: E402
from druglamp_amd.model import MInterface  # noqa: E402
from druglamp_amd.synthetic import make_batch  # noqa: E402
from druglamp_amd.trainer import Trainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
model.branch_streams = False       # ONE stream: an event pair on a side stream brackets whatever else shares the chip with the launch
                                   # (round 4's table showed the adaptors' 40-us products at 110-140 us for that reason)
trainer = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16)
batch, meta = make_batch(args.batch, dev, seed=100, with_graph=True, llm_dtype=torch.bfloat16)
for _ in range(3):
    trainer.training_step(batch, meta=meta, cur_epoch=1)
torch.cuda.synchronize()

recs = []
orig = ops.gemm


def timed(x, w, **kw):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    queued = len(ops._wgroup) if ops._wgroup is not None else -1
    a.record()
    out = orig(x, w, **kw)
    b.record()
    if queued >= 0 and len(ops._wgroup) > queued:        # left to the block's grouped launch (dl_gemm_group): timed there
        recs.append(((kw["M"], kw["N"], kw["K"], "TT", "grp", str(out.dtype)[6:]), None, None))
        return out
    epi = ("b" if kw.get("bias") is not None else "") + ("r" if kw.get("residual") is not None else "") + \
          ("g" if kw.get("act") else "") + ("p" if kw.get("pre_out") is not None else "") + \
          ("G" if kw.get("dact_pre") is not None else "") + ("d" if kw.get("dropout_p", 0) > 0 else "") + \
          ("+" if kw.get("accumulate") else "")
    lay = ("T" if kw.get("x_kslow") else "N") + ("T" if kw.get("w_kslow") else "N")
    recs.append(((kw["M"], kw["N"], kw["K"], lay, epi, str(out.dtype)[6:]), a, b))
    return out


arecs = []
def wrap_attn(name):
    f = getattr(ops, name)
    def g(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = f(*a, **kw); e1.record()
        arecs.append(((name, kw["n_problems"], kw["n_heads"], kw["n_segments"], kw["Lq"], kw["Lk"], kw["head_dim"],
                       str(kw.get("q_strides")), str(kw.get("k_strides")), kw.get("raw_logits") is not None), e0, e1))
        return out
    setattr(ops, name, g)
wrap_attn("attn_fwd"); wrap_attn("attn_bwd")
ops.gemm = timed
orig_flush = ops.flush_wgrads


def timed_flush():
    if not ops._wgroup:
        return orig_flush()
    members = [(t[0].M, t[0].N, t[0].K) for t in ops._wgroup]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    orig_flush()
    b.record()
    # one row per grouped launch: M column = number of members, N = their outputs / 1024, K = rows
    recs.append(((len(members), sum(m * n for m, n, _ in members) // 1024, members[0][2], "TT", "GROUP", "float32"), a, b))
    grecs.append((members, a, b))


grecs = []
ops.flush_wgrads = timed_flush
import druglamp_amd.functional as Fn  # noqa: E402
if hasattr(Fn, "ops"):
    Fn.arecs = []
def wrap_attn(name):
    f = getattr(ops, name)
    def g(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = f(*a, **kw); e1.record()
        arecs.append(((name, kw["n_problems"], kw["n_heads"], kw["n_segments"], kw["Lq"], kw["Lk"], kw["head_dim"],
                       str(kw.get("q_strides")), str(kw.get("k_strides")), kw.get("raw_logits") is not None), e0, e1))
        return out
    setattr(ops, name, g)
wrap_attn("attn_fwd"); wrap_attn("attn_bwd")
ops.gemm = timed
trainer.training_step(batch, meta=meta, cur_epoch=1)
torch.cuda.synchronize()
agg = defaultdict(lambda: [0.0, 0])
for key, a, b in recs:
    agg[key][0] += a.elapsed_time(b) if a is not None else 0.0
    agg[key][1] += 1
tot = sum(v[0] for v in agg.values())
print("total gemm %.2f ms in %d calls (%d of them members of %d grouped launches: rows 'grp' carry no time, rows 'GROUP' = one launch each, "
      "M = members, N = outputs / 1024)" % (tot, len(recs) - len(grecs), sum(len(g[0]) for g in grecs), len(grecs)))
print("%9s %6s %8s %3s %-6s %-8s %4s %9s %8s %7s" % ("M", "N", "K", "lay", "epi", "out", "n", "us/call", "TF/s", "ms"))
for key, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    M, N, K, lay, epi, od = key
    if epi == "GROUP":
        tf = 2.0 * N * 1024 * K * n / (ms * 1e-3) / 1e12
    else:
        tf = 2.0 * M * N * K * n / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    print("%9d %6d %8d %3s %-6s %-8s %4d %9.1f %8.1f %7.3f" % (M, N, K, lay, epi, od, n, ms / n * 1e3, tf, ms))

agg = defaultdict(lambda: [0.0, 0])
for key, a, b in arecs:
    agg[key][0] += a.elapsed_time(b); agg[key][1] += 1
for key, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-90s n=%d  %.1f us/call" % (key, n, ms / n * 1e3))
