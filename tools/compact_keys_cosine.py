"""Spread of the whole-model gradient cosine between PGCA over the distinct drug rows and over all 512 rows in the bf16 pipeline
(tests/test_model_gpu.py::test_cross_attention_over_the_distinct_drug_rows_whole_model, batch 4): the same computation for several
seeds, so that the test's threshold can be read against the spread instead of one draw.  usage: python tools/compact_keys_cosine.py"""
import copy, os, sys, torch
sys.path.insert(0, os.environ.get("DL_TREE", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.protein_plan import BatchHints
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer
DEV = torch.device("cuda:0")
dt = torch.bfloat16
for seed in range(8):
    torch.manual_seed(3 + seed)
    cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
    ref = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(DEV).train()
    ref.pmma.p_drop = 0.0; ref.pmma.embeddings.p_drop = 0.0
    ref.set_compute_dtype(dt)
    ref.drug_extractor.compact_min_rows = 0
    batch, meta = make_batch(int(os.environ.get("CK_BATCH", "4")), DEV, seed=11 + seed, with_graph=True, llm_dtype=dt)
    blk = Trainer.padding_hints_of(meta, batch)["drug_tokens"]
    cmp_ = copy.deepcopy(ref)
    ref.compact_keys, cmp_.compact_keys = False, True
    feat_d, feat_p, labels, llm_d, llm_p = batch
    outs = []
    for m in (ref, cmp_):
        score = m(feat_d, feat_p, llm_d, llm_p, hints=BatchHints(drug_tokens=blk, raw_attention=False))[-1]
        (score.float().view(-1) * (torch.tensor([1.0, -2.0, 0.5, 3.0], device=DEV).repeat(int(os.environ.get("CK_BATCH", "4")) // 4))).sum().backward()
        outs.append(score.float())
    pa = [(n, a.grad, b.grad) for (n, a), (_, b) in zip(ref.named_parameters(), cmp_.named_parameters()) if a.grad is not None]
    va, vb = torch.cat([a.double().flatten() for _, a, _ in pa]), torch.cat([b.double().flatten() for _, _, b in pa])
    cos = float(torch.dot(va, vb) / (va.norm() * vb.norm()))
    worst = min(float(torch.dot(b.double().flatten(), a.double().flatten()) / (a.double().norm() * b.double().norm() + 1e-30))
                for n, a, b in sorted(pa, key=lambda t: -float(t[1].norm()))[:40])
    rel = float((outs[1] - outs[0]).abs().max() / outs[0].abs().max())
    print("seed %d: score relerr %.4f  whole-gradient cosine %.5f  worst of the 40 largest tensors %.4f" % (seed, rel, cos, worst), flush=True)
