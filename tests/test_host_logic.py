"""CPU-only checks of the host side: C-ABI export surface, state_dict ABI vs the reference's key
list, config surface, schedules, label matrices, parameter-run logic."""
import json
import math
import os
import re

import numpy as np
import pytest
import torch

from tests.helpers import load, sd_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from druglamp_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "druglamp_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dl_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 34
    L = _lib.lib()                                   # dlopen + argtypes; no compute without a GPU
    for name in sorted(declared):
        assert hasattr(L, name), "libdruglamp_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "%s has no ctypes signature" % name
    assert L.dl_version() >= 100
    # argument validation happens before any launch and reports through dl_last_error
    assert L.dl_gemm(None, None) != 0
    assert b"null" in L.dl_last_error()


@pytest.mark.parametrize("kind", ["DrugLAMP", "DrugLAMP2C2P", "DrugLAMPwoLLM"])
def test_state_dict_abi_matches_reference(kind):
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface
    cfg = load_yaml_into(get_cfg_defaults(), kind)
    m = MInterface(kind, cfg).load_model(n_drug_feature=384, n_prot_feature=640, unused_dataset_attr=1)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("drug_extractor.")}
    ref = {k: s for k, s, _ in sd_spec(load("model_" + kind))}
    assert set(mine) == set(ref)
    assert all(mine[k] == ref[k] for k in ref)
    assert sum(p.numel() for p in m.parameters()) == 14026568
    # the shared ProteinCNN (basic_model.py:79-86)
    assert m.ssl_model.extractor is m.protein_extractor
    with pytest.raises(ValueError):
        MInterface("NoSuchModel", cfg).load_model()


def test_pmma_state_dict_abi():
    from druglamp_amd.configs import get_model_defaults
    from druglamp_amd.model.PMMA import PairedMultimodelAttention
    g = load("pmma_full")
    m = PairedMultimodelAttention(get_model_defaults(128), vis=False)
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    ref = {k: s for k, s, _ in sd_spec(g)}
    assert mine == ref
    assert sum(p.numel() for p in m.parameters()) == 10252800


def test_config_surface():
    from druglamp_amd.configs import get_cfg_defaults, get_model_defaults, load_yaml_into
    c = load_yaml_into(get_cfg_defaults(), "DrugLAMP2C2P")
    assert c.SOLVER.BATCH_SIZE == 16 and c.SOLVER.LR == 1e-4 and c.SOLVER.SSL_LR == 3e-5 and c.SOLVER.CM_LR == 3e-5
    assert c.RS.SSL is True and c.RS.CM is True and c.RS.INIT_EPOCH == 5 and c.RS.EPOCH_STEP == 5
    assert c.RS.MAX_MARGIN == 0.5 and c.RS.RESET_EPOCH == 100 and c.DECODER.BINARY == 1
    assert c.PROTEIN.SEQ_LEN == 2304 and c.PROTEIN.SITE_LEN == 9 and c.DRUG.NODE_IN_FEATS == 75
    assert load_yaml_into(get_cfg_defaults(), "DrugLAMP").RS.CM is False
    m = get_model_defaults(128)
    assert m.hidden_size == 256 and m.mol_len == m.feat_len == 256
    assert m.transformer.num_heads == 4 and m.transformer.dropout_rate == 0.1 and m.mlha_dropout == 0
    with pytest.raises(KeyError):
        get_cfg_defaults().merge_from_dict({"NOPE": 1})


def test_margin_schedule_and_label_matrix():
    from druglamp_amd.model.cross_modality import MarginSchedule, label_matrix
    g = load("losses")
    sch = MarginSchedule(m_ori=0.5, n_re=100)
    got = [sch.margin]
    for _ in range(205):
        sch.step()
        got.append(sch.margin)
    assert np.allclose(got, g["margins"], rtol=0, atol=1e-12)
    meta = [{"Prot_ID": "a", "Drug_ID": 1, "Y": 1.0}, {"Prot_ID": "b", "Drug_ID": 1, "Y": 0.0},
            {"Prot_ID": "a", "Drug_ID": 2, "Y": 0.0}, {"Prot_ID": "a", "Drug_ID": 1, "Y": 0.0}]
    pidx, didx, gt = label_matrix(meta)
    assert pidx == [3, 1] and didx == [3, 2]            # LAST occurrence per id, first-seen order
    assert gt.tolist() == [[0, 0], [0, 0]]              # later duplicate overwrites; unobserved (b,2) = 0
    assert label_matrix(meta, use_cm=False)[2].tolist() == [[0, 0], [0, -1]]


def test_lr_schedule():
    from druglamp_amd.trainer import CosineAnnealingWarmupRestarts
    s = CosineAnnealingWarmupRestarts(100, 1e-4, 1e-8, 20)
    assert s.lr == 1e-8                                  # first epoch runs at min_lr
    lrs = [s.step() for _ in range(100)]
    assert abs(lrs[0] - (1e-8 + (1e-4 - 1e-8) / 20)) < 1e-15
    assert abs(lrs[19] - 1e-4) < 1e-12                   # step 20 = end of warm-up = max_lr
    assert abs(lrs[59] - (1e-8 + (1e-4 - 1e-8) * (1 + math.cos(math.pi * 40 / 80)) / 2)) < 1e-15
    assert abs(lrs[99] - 1e-8) < 1e-15                   # cycle restart


def test_mlm_mask_counts():
    from druglamp_amd.model.self_supervised_learning import get_mask_subset_with_prob, mask_with_tokens
    torch.manual_seed(0)
    seq = torch.randint(1, 26, (4, 200)).double()
    seq[0, 150:] = 0
    seq[3, 10:] = 0
    valid = ~mask_with_tokens(seq, (0,))
    m = get_mask_subset_with_prob(valid, 0.15)
    assert (m & ~valid).sum() == 0
    for b in range(4):
        assert int(m[b].sum()) == math.ceil(0.15 * int(valid[b].sum()))


def test_flat_params_runs_cpu():
    from druglamp_amd.trainer import FlatParams
    ps = [torch.nn.Parameter(torch.randn(s)) for s in [(3, 5), (7,), (4, 4), (2,), (6, 2)]]
    vals = [p.detach().clone() for p in ps]
    flat = FlatParams(ps)
    assert all(torch.equal(p.detach(), v) for p, v in zip(ps, vals))
    assert all(p.data_ptr() % 16 == 0 for p in ps)
    for i in (0, 1, 3):
        ps[i].grad = torch.full_like(ps[i], float(i + 1))
    idx = flat.pack_grads()
    assert idx == [0, 1, 3]
    runs = flat.runs(idx, lambda i: 0)
    assert [(s, e) for s, e, _ in runs] == [(0, 16 + 8), (flat.offsets[3], flat.offsets[3] + 4)]
    assert float(flat.grads[:15].sum()) == 15.0 and float(flat.grads[16:23].sum()) == 14.0
    ps[0].data.add_(1.0)                                  # views write through to the arena
    assert torch.equal(flat.arena[:15].view(3, 5), ps[0].detach())


def test_c_abi_argument_validation_without_gpu():
    """Every entry point validates shapes / alignment / workspaces BEFORE launching anything and reports
    through the status code + dl_last_error(); none of this needs a device."""
    import ctypes as C
    from druglamp_amd import _lib
    L = _lib.lib()
    buf = (C.c_char * 4096)()
    p = C.addressof(buf)
    p16 = (p + 15) // 16 * 16
    a = _lib.GemmArgs()
    a.X, a.W, a.C = p16, p16, p16
    a.M, a.N, a.K = 64, 64, 0                       # bad shape
    a.ldx = a.ldw = a.ldc = 64
    assert L.dl_gemm(C.byref(a), None) == -2 and b"bad shape" in L.dl_last_error()
    a.K = 60                                        # bf16 needs K % 8 == 0
    a.in_dtype = a.out_dtype = _lib.DL_BF16
    a.ldx = a.ldw = 64
    assert L.dl_gemm(C.byref(a), None) == -3
    a.K = 64
    a.X = p16 + 2                                   # misaligned operand
    assert L.dl_gemm(C.byref(a), None) == -3 and b"aligned" in L.dl_last_error()
    a.X = p16
    a.K = 512                                       # 8 k-steps: a 4-way split keeps 4 non-empty slabs
    a.split_k, a.out_dtype = 4, _lib.DL_F32         # split-K without workspace
    assert L.dl_gemm(C.byref(a), None) == -4 and b"workspace" in L.dl_last_error()
    assert L.dl_gemm_workspace_bytes(C.byref(a)) == 4 * 64 * 64 * 4
    a.K, a.split_k = 192, 8                         # 3 k-steps: the plan is trimmed to 3 non-empty slabs
    assert L.dl_gemm_workspace_bytes(C.byref(a)) == 3 * 64 * 64 * 4
    # weight-gradient form with the fused bias gradient: slabs + one [splits][M] strip; needs x_kslow, w_kslow, split_k = 0
    a.x_colsum = p16
    assert L.dl_gemm(C.byref(a), None) == -6 and b"x_colsum" in L.dl_last_error()
    a.x_kslow = a.w_kslow = 1
    a.split_k, a.K = 0, 64 * 64
    sp = L.dl_gemm_workspace_bytes(C.byref(a)) // ((64 * 64 + 64) * 4)
    assert sp >= 1 and L.dl_gemm_workspace_bytes(C.byref(a)) == sp * (64 * 64 + 64) * 4
    a.x_colsum, a.x_kslow, a.w_kslow = None, 0, 0
    f = _lib.AttnFwdArgs()
    f.Q = f.K = f.V = f.O = p16
    f.n_problems, f.n_heads, f.n_segments, f.Lq, f.Lk, f.head_dim, f.dtype = 1, 1, 1, 16, 16, 48, _lib.DL_BF16
    assert L.dl_attn_fwd(C.byref(f), None) == -6 and b"head_dim" in L.dl_last_error()
    f.head_dim, f.q_rs = 64, 3                      # stride not a multiple of 16 bytes
    assert L.dl_attn_fwd(C.byref(f), None) == -3
    assert L.dl_layernorm_fwd(p16, 6, p16, p16, p16, 6, None, None, 4, 6, 1e-6, 0, None) == -2
    assert L.dl_ntxent_fwd(p16, p16, 8, 48, 0.1, p16, p16, p16, 4096, None) == -6
    assert L.dl_cast(p16, 0, p16, 7, 16, None) == -1
    assert L.dl_adamw_step(p16, p16, p16, p16, 16, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0, 1.0, None, 0, None) == -1


def test_gemm_group_plan_without_gpu():
    """dl_gemm_group_plan is host logic: which groups of weight-gradient products the library takes, and the slab count per
    member (one common count so that tiles x slabs is about one round of 256 workgroups, at least eight 64-row k-steps per
    slab, trimmed so that no slab is empty)."""
    import ctypes as C
    from druglamp_amd import _lib
    L = _lib.lib()
    p16 = 1 << 20

    def member(M, N, K, **kw):
        a = _lib.GemmArgs()
        a.X = a.W = a.C = p16
        a.M, a.N, a.K = M, N, K
        a.ldx, a.ldw, a.ldc = M, N, N
        a.x_kslow = a.w_kslow = 1
        a.in_dtype, a.out_dtype = _lib.DL_BF16, _lib.DL_F32
        a.split_k = 0
        for k, v in kw.items():
            setattr(a, k, v)
        return a

    def plan(members):
        arr = (_lib.GemmArgs * len(members))(*members)
        sp = (C.c_int32 * len(members))()
        rc = L.dl_gemm_group_plan(arr, len(members), sp)
        return rc, list(sp)

    # a paired d = 256 block at 8192 rows: 56 tiles of 128 x 256 -> 4 slabs each
    block = [member(256, 1024, 8192), member(1024, 256, 8192), member(256, 256, 8192), member(256, 512, 8192), member(768, 256, 8192)] * 2
    rc, sp = plan(block)
    assert rc == 0 and sp == [4] * 10
    # few k-steps: at least eight per slab (K = 1024 = 16 steps -> 2 slabs); 1000 rows = 16 steps (the last one partial)
    assert plan([member(128, 128, 1024), member(128, 128, 1000)]) == (0, [2, 2])
    # large members over >= 16384 rows take 256-row tiles: 2048 x 512 -> 16 tiles, 512 x 2048 -> 16, one slab count for both
    rc, sp = plan([member(2048, 512, 65536), member(512, 2048, 65536)])
    assert rc == 0 and sp == [8, 8]
    # not a weight-gradient product / odd sizes / too many members: DL_ERR_UNSUPPORTED, the caller issues dl_gemm calls
    assert plan([member(256, 256, 8192, x_kslow=0)])[0] == -6
    assert plan([member(256, 252, 8192)])[0] == -6
    assert plan([member(256, 256, 8192, accumulate=1)])[0] == -6
    assert plan([member(256, 256, 8192)] * 17)[0] == -6 and b"dl_gemm_group_plan" in L.dl_last_error()
    # dl_gemm_group validates the workspaces before it launches anything
    arr = (_lib.GemmArgs * 2)(member(256, 256, 8192), member(256, 256, 8192))
    assert L.dl_gemm_group(arr, 2, None) == -4 and b"workspace" in L.dl_last_error()


def test_cm_labels_pack_the_label_matrix_into_static_shapes():
    """CMLabels (model/cross_modality.py): label_matrix() of a batch padded to (B, B) with the padding marked ignored (-1),
    unique-row indices padded with row 0, row masks and counts — one packed buffer, refilled in place (what a captured CM
    step reads).  CPU build of the same code path (no pinned staging without a GPU)."""
    from druglamp_amd.model.cross_modality import CMLabels, CrossModality, label_matrix
    import copy, pickle
    meta = [{"Prot_ID": "p%d" % (i % 3), "Drug_ID": "d%d" % (i % 5), "Y": float(i % 2)} for i in range(8)]
    lab = CMLabels(8, "cpu")
    ptrs = (lab.idx.data_ptr(), lab.mask.data_ptr(), lab.n.data_ptr(), lab.gt.data_ptr())
    for use_cm in (True, False, True):
        for _ in range(5):                                  # more fills than staging slots
            lab.fill(meta, use_cm)
        pi, di, gt = label_matrix(meta, use_cm)
        assert lab.idx[0, :len(pi)].tolist() == pi and lab.idx[1, :len(di)].tolist() == di
        assert lab.idx[0, len(pi):].sum() == 0 and lab.idx[1, len(di):].sum() == 0
        assert lab.n.tolist() == [float(len(pi)), float(len(di))]
        assert lab.mask[0, :, 0].tolist() == [1.0] * len(pi) + [0.0] * (8 - len(pi))
        assert (lab.gt[:len(pi), :len(di)].numpy() == gt).all()
        assert (lab.gt[len(pi):] == -1).all() and (lab.gt[:, len(di):] == -1).all()
    assert ptrs == (lab.idx.data_ptr(), lab.mask.data_ptr(), lab.n.data_ptr(), lab.gt.data_ptr())      # refilled in place
    with pytest.raises(ValueError):
        lab.fill(meta[:5])
    cm = CrossModality(hidden_size=128, max_margin=0.5, n_re=10)
    cm._label_blocks[(8, "cpu")] = lab
    assert copy.deepcopy(cm)._label_blocks == {} and pickle.loads(pickle.dumps(cm))._label_blocks == {}


def test_padding_hints_from_the_collate_records():
    """Trainer.padding_hints_of: the batch maximum of the collate's Drug_Tokens records, rounded up to a multiple of 128 (few
    graph keys); no hint without the records or when the block would save nothing.  Hints travel as an explicit argument
    (protein_plan.BatchHints -> model(..., hints=...)): there is no module-global hint state."""
    from druglamp_amd import functional as Fn
    from druglamp_amd.protein_plan import BatchHints
    from druglamp_amd.trainer import Trainer
    llm_d = torch.zeros(3, 512, 8)
    batch = (None, None, None, llm_d, None)
    mk = lambda *n: [{"Drug_Tokens": k} for k in n]      # noqa: E731
    assert Trainer.padding_hints_of(mk(12, 64, 40), batch) == {"drug_tokens": 128}
    assert Trainer.padding_hints_of(mk(12, 129, 40), batch) == {"drug_tokens": 256}
    assert Trainer.padding_hints_of(mk(385, 3, 3), batch) == {}                  # block 512: nothing to save
    assert Trainer.padding_hints_of(mk(384, 3, 3), batch) == {"drug_tokens": 384}
    assert Trainer.padding_hints_of([{"Drug_Tokens": 5}, {"Y": 1.0}], batch) == {} and Trainer.padding_hints_of(None, batch) == {}
    assert not hasattr(Fn, "padding_hints") and not hasattr(Fn, "padding_hint")
    assert BatchHints().key() == (0, None) and BatchHints(128).drug_tokens == 128


def test_protein_plan_covers_every_position_with_the_right_multiplicities():
    """protein_plan: every position of the tiled sequence has exactly one representative row, the weights add up to the
    sequence length, representatives keep their receptive field (7 left / 8 right) inside their segment, and the symbolic
    input windows of a position and of its representative are identical (what makes the compact network equal to the full
    one for ANY parameter values).  The fp64 network-level check lives in tests/test_protein_plan_cpu.py."""
    import numpy as np
    from druglamp_amd.data import repeat_integer_label
    from druglamp_amd.protein_plan import HALO, RF_LEFT, RF_RIGHT, ProteinPlan, sample_template
    S = 2304
    for L in (13, 98, 254, 398, 766, 767, 1022, 1150, 1151, 2000):
        src, w, first, stride, count, row_of = sample_template(L, S)
        assert abs(float(np.clip(w, 0, None).sum()) - S) < 1e-6 and (row_of >= 0).all()
        # symbolic sequence: distinct residue symbols inside a period (so only the tiling creates equalities), -1 outside
        P = L + 2
        sym = np.zeros(S, np.int64)
        sym[:(S // P) * P] = np.tile(np.arange(1, P + 1), S // P)
        ext = np.concatenate([np.full(RF_LEFT, -1), sym, np.full(RF_RIGHT, -1)])
        win = np.lib.stride_tricks.sliding_window_view(ext, RF_LEFT + RF_RIGHT + 1)        # window of position t = win[t]
        rep_pos = src[row_of]
        assert (rep_pos >= 0).all() and (win[np.arange(S)] == win[rep_pos]).all(), L
        # a representative's receptive field lies inside its segment (no halo row within reach unless it is a real boundary)
        for r in np.unique(row_of):
            t = src[r]
            lo, hi = r - RF_LEFT, r + RF_RIGHT
            seg = src[max(lo, 0):hi + 1]
            inside = seg[seg >= 0]
            assert (np.diff(inside) == 1).all()
            assert t - inside.min() == min(RF_LEFT, t) and inside.max() - t == min(RF_RIGHT, S - 1 - t), (L, t)
        # multiplicities = how many positions map to the row
        cnt = np.bincount(row_of, minlength=len(w))
        assert (cnt == np.clip(w, 0, None)).all()
    plan = ProteinPlan([98, 398, 1022], S, bucket=64)
    assert plan.rows % 64 == 0 and plan.row_of.shape == (3 * S,) and plan.src.shape == (plan.rows,)
    assert plan.pays() and not ProteinPlan([1500], S).pays()
    # flat tables: the representative of position (b, t) reads sample b
    b_of = plan.src[plan.row_of] // S
    assert (b_of == np.repeat(np.arange(3), S)).all()


def test_plan_spec_row_counts_and_capacity_classes():
    """What the training thread computes per batch (round 5: the tables themselves are built on the device): the vectorised
    row count equals the per-sample tables' sizes in every case of the plan, the capacity classes of captured graphs form a
    monotone ladder that never undershoots, and the CPU form of PlanDev holds the host tables."""
    import numpy as np
    from druglamp_amd.protein_plan import PlanDev, PlanSpec, ProteinPlan, ROW_BUCKET, row_class, sample_rows, sample_template
    for S in (2304, 300, 64):
        Ls = list(range(1, S + 40))
        assert [sample_template(L, S)[0].shape[0] for L in Ls] == sample_rows(Ls, S).tolist()
    prev = 0
    for need in range(1, 400 * ROW_BUCKET, 997):
        c = row_class(need)
        assert c >= need and c % ROW_BUCKET == 0 and c >= prev and c <= max(1.26 * need + ROW_BUCKET, 2 * ROW_BUCKET)
        prev = c
    assert len({row_class(n) for n in range(150_000, 183_000, 500)}) <= 2         # batch 256: the lengths' spread is one or two classes
    rs = np.random.RandomState(0)
    lengths = rs.randint(100, 1023, 16)
    spec = PlanSpec(lengths, 2304)
    host = ProteinPlan(lengths, 2304, bucket=1)
    assert spec.need == host.rows and spec.pays() and not PlanSpec([1500, 1400], 2304).pays()
    pd = PlanDev(spec, "cpu", rows=row_class(spec.need))
    R = host.rows
    assert np.array_equal(pd.src.numpy()[:R], host.src) and (pd.src.numpy()[R:] == -1).all() and (pd.w.numpy()[R:] == -1).all()
    assert np.array_equal(pd.rep.numpy()[:R], host.rep) and np.array_equal(pd.row_of.numpy(), host.row_of)
    assert np.array_equal(pd.period.numpy(), host.period)
    # a protein whose period exceeds the sequence (reps = 0) keeps the plain layout and makes no periodic claim
    assert ProteinPlan([98, 400], 300, bucket=1).period.tolist() == [100, 0]
    with pytest.raises(ValueError):
        pd.fill(PlanSpec(rs.randint(900, 1023, 16) * 0 + 1022, 2304))              # needs more rows than the tables hold


def test_ops_reject_cpu_tensors():
    from druglamp_amd import ops
    x = torch.randn(8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(x, x, M=8, N=8, K=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_fwd(x, torch.ones(8), torch.zeros(8), 1e-6)


def test_load_reference_checkpoint_strips_the_lightning_prefix(tmp_path):
    """Reference checkpoints are Lightning files of ExpModule: {"state_dict": {"exp_model.<key>": tensor, ...}}
    (trainer.py:43,151-156); the helper strips the prefix, builds the lazy SimSiam projectors when present and loads
    strictly."""
    import torch
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface, load_reference_checkpoint
    cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
    torch.manual_seed(0)
    src = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640)
    src.ssl_model.build_projectors(128, 385)
    ck = {"epoch": 3, "state_dict": {"exp_model." + k: v.clone() for k, v in src.state_dict().items()}}
    ck["state_dict"]["valid_metrics.auroc.preds"] = torch.zeros(3)          # Lightning stores metric states next to the model
    path = tmp_path / "max_val_ausum= 1.50000.ckpt"
    torch.save(ck, path)
    torch.manual_seed(1)
    dst = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640)
    assert dst.ssl_model.net.projector is None
    res = load_reference_checkpoint(dst, str(path))
    assert not res.missing_keys and not res.unexpected_keys
    a, b = src.state_dict(), dst.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)


def test_batchnorm_over_one_row_raises_like_torch():
    """Training-mode BatchNorm1d over a single row: torch raises ValueError (the reference's MLPDecoder does for a batch of
    one pair); the HIP path raises the same error before any launch instead of producing NaN statistics."""
    import pytest
    import torch
    from druglamp_amd import functional as Fn
    bn = torch.nn.BatchNorm1d(4).train()
    with pytest.raises(ValueError, match="Expected more than 1 value per channel when training"):
        torch.nn.functional.batch_norm(torch.zeros(1, 4), bn.running_mean, bn.running_var, bn.weight, bn.bias, True)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel when training"):
        Fn.batch_norm_rows(bn, torch.zeros(1, 4))


def test_graph_auto_policy_is_per_step_kind_and_batch():
    """Trainer(graph_steps="auto"): replay at per-GPU batches <= 128 for every kind, and the CM kinds at any batch."""
    from types import SimpleNamespace
    from druglamp_amd.trainer import Trainer
    auto = SimpleNamespace(graph_steps="auto", GRAPH_AUTO_MAX_BATCH=Trainer.GRAPH_AUTO_MAX_BATCH)
    assert Trainer.wants_graph(auto, 32, False) and Trainer.wants_graph(auto, 128, False)
    assert not Trainer.wants_graph(auto, 256, False) and Trainer.wants_graph(auto, 256, True)
    on, off = SimpleNamespace(graph_steps=True), SimpleNamespace(graph_steps=False)
    assert Trainer.wants_graph(on, 256, False) and not Trainer.wants_graph(off, 32, True)


def test_capture_capacities_are_not_erased_by_a_batch_without_a_plan():
    """a batch whose compact ProteinCNN layout does not pay (or that has no Prot_Len / Drug_Tokens records) is
    served by the full layout FOR THAT BATCH; the capacities earlier batches of the shape asked for stay on record, and
    the next capture is never below them."""
    from types import SimpleNamespace
    from druglamp_amd.protein_plan import PlanSpec, row_class
    from druglamp_amd.trainer import Trainer
    tr = SimpleNamespace(fixed_caps=None, _graph_caps={})
    base = ("cls", 16)
    rs = np.random.RandomState(0)
    small, big = PlanSpec(rs.randint(100, 300, 16), 2304), PlanSpec(rs.randint(600, 900, 16), 2304)
    assert Trainer._capture_caps(tr, base, 128, big) == (128, row_class(big.need))
    assert Trainer._capture_caps(tr, base, 0, None) == (0, None)                 # no records: every row, this batch only
    assert tr._graph_caps[base] == (128, row_class(big.need))
    assert Trainer._capture_caps(tr, base, 128, small) == (128, row_class(big.need))   # never below an earlier request
    assert Trainer._capture_caps(tr, base, 256, small) == (256, row_class(big.need))
    nopay = PlanSpec([1500] * 16, 2304)
    assert not nopay.pays() and Trainer._capture_caps(tr, base, 256, nopay) == (256, None)
    assert tr._graph_caps[base] == (256, row_class(big.need))
    tr.fixed_caps = (384, 40960)
    assert Trainer._capture_caps(tr, base, 128, small) == (384, 40960)


def test_id_code_of_missing_and_mixed_ids():
    """NaN / inf ids (a pandas missing id) hash like any other non-integral id instead of raising from int()."""
    from druglamp_amd.dist_ops import id_code
    assert id_code(float("nan")) == id_code(np.float64("nan")) >= (1 << 62)
    assert id_code(float("inf")) >= (1 << 62) and id_code(float("inf")) != id_code(float("-inf"))
    assert id_code(5) == id_code(np.int64(5)) == id_code(5.0) == 5 and id_code(True) != 1
    assert id_code("P12345") == id_code("P12345") >= (1 << 62)
