"""ProteinCNN on distinct rows, network-level check on the CPU in fp64 (round 4): the compact computation the HIP path
performs — convolutions over the plan's segments as ONE sequence, BatchNorm statistics weighted by the multiplicities, halo
rows re-zeroed, rows expanded through `row_of` — equals the reference's full computation (model/basic_model.py:155-180 over
sequences tiled by utils.py:392-412) in outputs, batch statistics and every parameter gradient."""
import numpy as np
import torch
import torch.nn.functional as F

from druglamp_amd.protein_plan import ProteinPlan, sample_template, HALO
from druglamp_amd.data import repeat_integer_label
torch.manual_seed(0)
S = 2304; C = 128
def full_cnn(emb, convs, bns, ids, fill):
    v = F.embedding(ids, emb, padding_idx=0)
    v = torch.cat((v, fill.unsqueeze(-1)), -1).transpose(2, 1)
    stats = []
    for (w, b), (g, be) in zip(convs, bns):
        v = F.relu(F.conv1d(v, w, b, padding="same"))
        mean = v.mean((0, 2)); var = v.var((0, 2), unbiased=False)
        stats.append((mean, var))
        v = (v - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5) * g[None, :, None] + be[None, :, None]
    return v.transpose(1, 2), stats     # (B, S, C) channel-last
def compact_cnn(emb, convs, bns, ids, fill, plan):
    src = torch.from_numpy(plan.src.astype(np.int64)); w = torch.from_numpy(plan.w).double()
    ok = src >= 0
    idf = ids.reshape(-1)[src.clamp(min=0)]
    x = torch.cat((F.embedding(idf, emb, padding_idx=0), fill.reshape(-1)[src.clamp(min=0)].unsqueeze(-1)), -1)
    x = x * ok.unsqueeze(-1)
    v = x.t().unsqueeze(0)              # (1, C, R)
    n = plan.n
    stats = []
    wp = w.clamp(min=0)
    for (cw, cb), (g, be) in zip(convs, bns):
        v = F.relu(F.conv1d(v, cw, cb, padding="same"))
        mean = (v[0] * wp).sum(1) / n
        var = (((v[0] - mean[:, None]) ** 2) * wp).sum(1) / n
        stats.append((mean, var))
        v = (v - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5) * g[None, :, None] + be[None, :, None]
        v = v * (w >= 0)[None, None, :]
    z = v[0].t()                        # (R, C)
    return z[torch.from_numpy(plan.row_of.astype(np.int64))].reshape(plan.B, plan.S, -1), stats
def run(lengths):
    B = len(lengths)
    ids = torch.zeros(B, S, dtype=torch.long); fill = torch.zeros(B, S, dtype=torch.float64)
    for b, L in enumerate(lengths):
        codes = np.random.randint(0, 26, L)     # zeros allowed (unknown letters)
        ids[b] = torch.from_numpy(repeat_integer_label(codes, S)).long()
        E = (S // (L + 2)) * (L + 2)
        fill[b, E:] = 1.0
    mk = lambda *s: (torch.randn(*s, dtype=torch.float64) * 0.2).requires_grad_(True)
    emb = mk(27, C - 1); convs = [(mk(C, C, k), mk(C)) for k in (3, 6, 9)]; bns = [(mk(C), mk(C)) for _ in range(3)]
    params = [emb] + [t for p in convs for t in p] + [t for p in bns for t in p]
    proj = torch.randn(B, S, C, dtype=torch.float64)
    zf, sf = full_cnn(emb, convs, bns, ids, fill)
    gf = torch.autograd.grad((zf * proj).sum(), params)
    plan = ProteinPlan(lengths, S, bucket=64)
    assert abs(plan.w.clip(min=0).sum() - B * S) < 1e-6, (plan.w.clip(min=0).sum(), B * S)
    zc, sc = compact_cnn(emb, convs, bns, ids, fill, plan)
    gc = torch.autograd.grad((zc * proj).sum(), params)
    eo = (zf - zc).abs().max().item()
    eg = max(((a - b).abs().max() / (a.abs().max() + 1e-30)).item() for a, b in zip(gf, gc))
    es = max(((a - b).abs().max()).item() for s1, s2 in zip(sf, sc) for a, b in zip(s1, s2))
    assert eo < 1e-9 and eg < 1e-9 and es < 1e-10
def test_compact_protein_cnn_equals_the_full_computation_in_fp64():
    np.random.seed(1)
    run([98, 398])
    run([1022, 100, 511])
    run([13, 1150, 1151, 2302, 766, 767])      # tiny period, reps == 2, reps == 1 (plain layout), period divides S (768 * 3)
    run([254, 190, 46, 574])                   # periods 256 / 192 / 48 / 576 divide 2304: no zero tail
    for L in (18, 21, 34, 1136, 1138):
        run([L])
