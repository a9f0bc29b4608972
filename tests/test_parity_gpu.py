"""GPU parity: the HIP hot path (through the C ABI, via the reference-shaped modules) against the
golden fixtures of the real reference and against the CPU oracle.  fp32 mode must hold the
north-star tolerance (1e-4); bf16 mode is checked at 3e-2 (bf16 has 8 mantissa bits)."""
import numpy as np
import pytest
import torch

from tests.helpers import T, check_sub, det_state_dict, elemerr, gradnorms, load, relerr

pytestmark = pytest.mark.gpu

F32_TOL = 1e-4
BF16_TOL = 3e-2
# Element-wise bound of the fp32 legs (helpers.elemerr: every element against its OWN magnitude, down to 1 % of the largest):
# `relerr` divides by the largest reference element, so a large element could hide relative error on small ones.
# Measured on MI355X (profiles/r6_parity_margins.txt): <= 8.7e-5 over the PMMA / PGCA / MHLA goldens incl. gradients.
F32_ELEM_TOL = 3e-4


def check_gradnorms(named_params, g, tol):
    """Per-parameter gradient norms against the reference's.  Key-projection biases and the MHLA gate
    bias have an analytically ZERO gradient (softmax shift invariance): there both sides hold only
    rounding noise, which must merely be small next to the real gradients."""
    ref = gradnorms(g)
    sd = dict(named_params)
    scale = max(ref.values())
    for k, n_ref in ref.items():
        got = float(sd[k].grad.double().norm())
        if k.endswith(("key.bias", "key_mol.bias", "lin2.bias")) or (k == "in_proj_bias"):
            assert abs(got - n_ref) <= tol * 10 * scale, (k, got, n_ref)
        else:
            assert abs(got - n_ref) <= tol * 10 * max(n_ref, 1e-4), (k, got, n_ref)


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


class _Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def pmma_config(L, dropout=0.0):
    return _Cfg(hidden_size=256, mol_len=L, feat_len=L,
                transformer=_Cfg(num_heads=4, num_p_plus_s_layers=4, attention_dropout_rate=0, dropout_rate=dropout))


def build_pmma(g, L, dtype, vis=False):
    from druglamp_amd.model.PMMA import PairedMultimodelAttention
    m = PairedMultimodelAttention(pmma_config(L), vis=vis)
    sd = det_state_dict(g)
    missing = m.load_state_dict(sd, strict=True)
    m = m.to(_dev()).eval()
    m.compute_dtype = dtype
    return m


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
def test_pmma_mid_forward_backward(dtype, tol):
    g = load("pmma_mid")
    m = build_pmma(g, 64, dtype)
    prot = T("pmma_mid.prot", (2, 64, 256)).to(_dev()).requires_grad_(True)
    mol = T("pmma_mid.mol", (2, 64, 256)).to(_dev()).requires_grad_(True)
    enc, w, gw = m(prot, mol)
    assert w == [] and gw == []
    assert relerr(enc, g["encoded"]) <= tol
    (enc * T("pmma_mid.G", tuple(enc.shape)).to(_dev())).sum().backward()
    assert relerr(prot.grad, g["dprot"]) <= tol * 3
    assert relerr(mol.grad, g["dmol"]) <= tol * 3
    if dtype == torch.float32:
        assert elemerr(enc, g["encoded"]) <= F32_ELEM_TOL and elemerr(prot.grad, g["dprot"]) <= F32_ELEM_TOL
        assert elemerr(mol.grad, g["dmol"]) <= F32_ELEM_TOL
    sd = dict(m.named_parameters())
    assert relerr(sd["encoder.layer_with_mol.0.attn.query.weight"].grad[:8, :16], g["dW_l0_query"]) <= tol * 3
    assert relerr(sd["encoder.layer_with_mol.3.ffn.fc2.weight"].grad[:8, :16], g["dW_l3_fc2"]) <= tol * 3
    assert relerr(sd["encoder.layer_with_mol.1.attn.fc_mol.bias"].grad, g["db_l1_fc_mol"]) <= tol * 3
    assert relerr(sd["embeddings.pe_mol"].grad[0, :4, :16], g["dpe_mol"]) <= tol * 3
    check_gradnorms(m.named_parameters(), g, tol)
    assert sd["embeddings.embedding.weight"].grad is None     # dead Linear, as in the reference


def test_pmma_dropout_on_against_the_oracle_fed_the_kernels_own_masks():
    """Dropout-ON arithmetic against the oracle (it had property tests only).  The HIP path never stores a mask: it
    is a counter hash of (site seed, element index) evaluated inside the GEMM epilogues / dl_add_rowmod_dropout and again in
    backward.  Here the fp32 PMMA runs in TRAINING mode (p = 0.1, the shipped value); the masks it used are then replayed
    site by site through dl_dropout_apply on a tensor of ones (same seeds, in the order the forward drew them) and handed to
    the oracle's training-mode PMMA, itself pinned to the reference in train mode by tests/golden/pmma_drop.npz.  Output and
    gradients must agree at the fp32 tolerance: a mask applied at the wrong place (before GELU, before the positional add),
    with the wrong scale, or differently in forward and backward fails here."""
    from druglamp_amd import ops
    from druglamp_amd.model.PMMA import PairedMultimodelAttention
    from oracle import druglamp_oracle as O
    g = load("pmma_drop")
    B, L, d, p = 2, 64, 256, float(g["p"])
    m = PairedMultimodelAttention(pmma_config(L, dropout=p), vis=False)
    m.load_state_dict(det_state_dict(g), strict=True)
    m = m.to(_dev()).train()
    m.compute_dtype = torch.float32
    prot = T("pmma_drop.prot", (B, L, d)).to(_dev()).requires_grad_(True)
    mol = T("pmma_drop.mol", (B, L, d)).to(_dev()).requires_grad_(True)
    ops.manual_seed(77)
    enc, _, _ = m(prot, mol)
    (enc * T("pmma_drop.G", tuple(enc.shape)).to(_dev())).sum().backward()
    # replay the masks: one seed per site, drawn in the forward's order (PMMA.py: mol embedding, prot embedding; per block and
    # stream (fc1, fc2) — functional.TransformerBlockFn)
    ops.manual_seed(77)

    def mask(width):
        one = torch.ones((B * L, width), dtype=torch.float32, device=_dev())
        return ops.dropout_apply(one, p, ops.next_seed()).reshape(B, L, width).cpu()
    masks = {"emb_mol": mask(d), "emb_prot": mask(d)}
    for i in range(4):
        w = d if i < 2 else 2 * d
        for s_ in range(2 if i < 2 else 1):
            masks["l%d.s%d.fc1" % (i, s_)] = mask(4 * w)
            masks["l%d.s%d.fc2" % (i, s_)] = mask(w)
    for k, v in masks.items():
        vals = torch.unique(v)
        assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.0 / (1.0 - p)) <= 1e-4, (k, vals)
        assert 0.07 < float((v == 0).float().mean()) < 0.13, k
    assert not torch.equal(masks["l0.s0.fc2"], masks["l0.s1.fc2"]) and not torch.equal(masks["emb_mol"], masks["emb_prot"])
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    pc = prot.detach().cpu().requires_grad_(True)
    mc = mol.detach().cpu().requires_grad_(True)
    ref = O.pmma_forward(sd, pc, mc, dropout_masks=masks)
    (ref * T("pmma_drop.G", tuple(ref.shape))).sum().backward()
    assert relerr(O.pmma_forward(sd, pc, mc).detach(), ref.detach()) > 1e-2            # (the masks matter)
    assert relerr(enc, ref.detach()) <= F32_TOL and elemerr(enc, ref.detach()) <= 10 * F32_TOL
    assert relerr(prot.grad, pc.grad) <= 3 * F32_TOL and relerr(mol.grad, mc.grad) <= 3 * F32_TOL
    named = dict(m.named_parameters())
    for k in ("encoder.layer_with_mol.0.ffn.fc1.weight", "encoder.layer_with_mol.1.ffn_mol.fc2.weight", "encoder.layer_with_mol.3.ffn.fc2.weight",
              "encoder.layer_with_mol.2.ffn.fc1.bias", "embeddings.pe_prot", "embeddings.pe_mol", "embeddings.mol_embeddings.weight",
              "encoder.layer_with_mol.0.attn.query_mol.weight"):
        assert relerr(named[k].grad, sd[k].grad) <= 3 * F32_TOL, k
    # a second forward draws fresh masks
    enc2, _, _ = m(prot, mol)
    assert relerr(enc2, enc) > 1e-2


def test_pmma_vis_maps():
    g = load("pmma_mid")
    m = build_pmma(g, 64, torch.float32, vis=True)
    prot = T("pmma_mid.prot", (2, 64, 256)).to(_dev())
    mol = T("pmma_mid.mol", (2, 64, 256)).to(_dev())
    with torch.no_grad():
        enc, w, gw = m(prot, mol)
    assert relerr(w[0][:, :, :4, :8], g["w0"]) <= F32_TOL
    assert relerr(gw[0][:, :, :4, :8], g["gw0"]) <= F32_TOL
    assert relerr(w[3][:, :, :4, :8], g["w3"]) <= F32_TOL


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
def test_pmma_full_forward(dtype, tol):
    g = load("pmma_full")
    m = build_pmma(g, 256, dtype)
    prot = T("pmma_full.prot", (2, 256, 256)).to(_dev()).requires_grad_(True)
    mol = T("pmma_full.mol", (2, 256, 256)).to(_dev()).requires_grad_(True)
    et = F32_ELEM_TOL if dtype == torch.float32 else None
    enc, _, _ = m(prot, mol)
    check_sub(enc, g, "encoded", tol, et)
    (enc * T("pmma_full.G", tuple(enc.shape)).to(_dev())).sum().backward()
    check_sub(prot.grad, g, "dprot", tol * 3, et)
    check_sub(mol.grad, g, "dmol", tol * 3, et)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("tag,shape", [("pgca_small", (48, 80, 3)), ("pgca_full", (256, 512, 2))])
def test_pgca(tag, shape, dtype, tol):
    from druglamp_amd.model.PGCA import GuidedCrossAttention
    Lq, Lk, B = shape
    g = load(tag)
    m = GuidedCrossAttention(embed_dim=128, num_heads=1)
    m.load_state_dict(det_state_dict(g), strict=True)
    m = m.to(_dev()).eval()
    m.compute_dtype = dtype
    q = T(tag + ".q", (Lq, B, 128)).to(_dev()).requires_grad_(True)
    kv = T(tag + ".kv", (Lk, B, 128)).to(_dev()).requires_grad_(True)
    out, raw = m(q, kv, kv)
    assert tuple(raw.shape) == (B, 1, Lq, Lk)
    assert relerr(out, g["out"]) <= tol
    assert relerr(raw[:, :, :8, :16], g["raw"]) <= tol
    assert relerr(raw.double().norm(dim=-1), g["rawnorm"]) <= tol
    (out * T(tag + ".G", tuple(out.shape)).to(_dev())).sum().backward()
    assert relerr(q.grad, g["dq"]) <= tol * 3
    assert relerr(kv.grad, g["dkv"]) <= tol * 3
    if dtype == torch.float32:
        assert elemerr(out, g["out"]) <= F32_ELEM_TOL and elemerr(raw[:, :, :8, :16], g["raw"]) <= F32_ELEM_TOL
        assert elemerr(q.grad, g["dq"]) <= F32_ELEM_TOL and elemerr(kv.grad, g["dkv"]) <= F32_ELEM_TOL
    check_gradnorms(m.named_parameters(), g, tol)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
def test_pgca_over_distinct_key_rows_equals_the_module_over_all_rows(dtype, tol):
    """Round 5: GuidedCrossAttention(key_tail=(8, w)) over block + 8 distinct key rows against the same module over the 512 rows
    the reference attends over (block rows + the 8 tail rows repeated w times): output, query gradient, the key gradient
    (a tail row's = the sum over its copies) and every parameter gradient.  The weights are the PGCA golden's."""
    from druglamp_amd.model.PGCA import GuidedCrossAttention
    Lq, B, blk, tail = 256, 4, 128, 8
    w = (512 - blk) // tail
    g = load("pgca_full")
    ms = []
    for _ in range(2):
        m = GuidedCrossAttention(embed_dim=128, num_heads=1)
        m.load_state_dict(det_state_dict(g), strict=True)
        m = m.to(_dev()).eval()
        m.compute_dtype = dtype
        ms.append(m)
    q0 = T("pgca_c.q", (Lq, B, 128)).to(_dev())
    kc0 = T("pgca_c.kv", (blk + tail, B, 128)).to(_dev())
    G = T("pgca_c.G", (Lq, B, 128)).to(_dev())
    qa, ka = q0.clone().requires_grad_(True), kc0.clone().requires_grad_(True)
    out_c, raw_c = ms[0](qa, ka, ka, need_weights=False, key_tail=(tail, w))
    assert raw_c is None
    (out_c * G).sum().backward()
    qb, kb = q0.clone().requires_grad_(True), kc0.clone().requires_grad_(True)
    kfull = torch.cat([kb[:blk], kb[blk:].unsqueeze(0).expand(w, tail, B, 128).reshape(w * tail, B, 128)], 0)
    out_f, _ = ms[1](qb, kfull, kfull)
    (out_f * G).sum().backward()
    assert relerr(out_c, out_f) <= tol
    assert relerr(qa.grad, qb.grad) <= tol * 3 and relerr(ka.grad, kb.grad) <= tol * 3
    for (n, a), (_, b) in zip(ms[0].named_parameters(), ms[1].named_parameters()):
        assert relerr(a.grad, b.grad) <= tol * 3, n
    with pytest.raises(ValueError):
        ms[0](qa, ka, ka, key_tail=(tail, w))              # raw logits are those of all 512 rows


@pytest.mark.parametrize("dtype,tol", [(torch.float32, F32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("tag,shape", [("mhla_toy", (32, 64, 2, 5)), ("mhla_full", (256, 1024, 2, 256))])
def test_mhla(tag, shape, dtype, tol):
    from druglamp_amd.model.PMMA import MultiHeadLinearAttention
    d, dd, B, L = shape
    g = load(tag)
    m = MultiHeadLinearAttention(d_model=d, d_diff=dd, nhead=8, dropout=0, activation="gelu")
    m.load_state_dict(det_state_dict(g), strict=True)
    m = m.to(_dev()).eval()
    m.compute_dtype = dtype
    v = T(tag + ".v", (B, L, d)).to(_dev()).requires_grad_(True)
    out = m(v)
    assert relerr(out, g["out"]) <= tol
    (out * T(tag + ".G", tuple(out.shape)).to(_dev())).sum().backward()
    assert relerr(v.grad, g["dv"]) <= tol * 3
    if dtype == torch.float32:
        assert elemerr(out, g["out"]) <= F32_ELEM_TOL and elemerr(v.grad, g["dv"]) <= F32_ELEM_TOL
    check_gradnorms(m.named_parameters(), g, tol)


def test_no_cpu_fallback():
    """The product path must refuse CPU tensors instead of silently computing somewhere else."""
    g = load("pmma_mid")
    from druglamp_amd.model.PMMA import PairedMultimodelAttention
    m = PairedMultimodelAttention(pmma_config(64), vis=False).eval()
    with pytest.raises(RuntimeError):
        m(T("pmma_mid.prot", (2, 64, 256)), T("pmma_mid.mol", (2, 64, 256)))
