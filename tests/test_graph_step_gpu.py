"""hipGraph-replayed training steps (trainer.GraphedStep) against the eager step.

The graph holds forward + BCE + backward + gradient packing of a cls step (reference step: trainer.py:196-200,225);
all-reduce and AdamW stay eager.  Checked here:
  * with dropout off, N graphed steps leave the parameter arena, the AdamW moments, the BatchNorm running statistics
    and the losses BIT-IDENTICAL to N eager steps (same kernels, same order, same arguments);
  * a different batch passed to a captured graph is copied into its static inputs (result == eager on that batch);
  * with dropout on, every replay draws fresh masks (device-side seed offset) although the launch arguments are frozen,
    and a replay's forward and backward agree on the masks (finite-difference-free check: with lr = 0 two replays give
    different losses; with the offset frozen they give identical ones).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _make(p_drop, graph, seed=4321, kind="DrugLAMP"):
    from druglamp_amd import ops
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface
    from druglamp_amd.trainer import Trainer
    torch.manual_seed(seed)
    ops.manual_seed(77)
    ops.seed_offset_tensor(DEV).zero_()
    cfg = load_yaml_into(get_cfg_defaults(), kind)
    model = MInterface(kind, cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(DEV)
    model.pmma.p_drop = p_drop
    model.pmma.embeddings.p_drop = p_drop
    tr = Trainer(model, cfg, device=DEV, compute_dtype=torch.bfloat16, graph_steps=graph)
    tr.set_lrs(1e-3, 1e-3, 1e-3)
    return tr


def _state(tr):
    bufs = torch.cat([b.detach().float().flatten() for b in tr.model.buffers()])
    return tr.flat.arena.clone(), tr.opt.exp_avg.clone(), tr.opt.exp_avg_sq.clone(), bufs


def test_graphed_steps_are_bit_identical_to_eager_steps_without_dropout():
    from druglamp_amd import ops
    from druglamp_amd.synthetic import make_batch
    batch, meta = make_batch(8, DEV, seed=7, with_graph=True, llm_dtype=torch.bfloat16)
    other, meta2 = make_batch(8, DEV, seed=8, with_graph=True, llm_dtype=torch.bfloat16)
    # steps 3.. are replays; step 5 brings new data (its own meta: the batch's protein lengths are what the ProteinCNN row
    # tables are built from — round 4 — and the device-side guard rejects records that do not belong to the batch)
    seq = [(batch, meta), (batch, meta), (batch, meta), (batch, meta), (other, meta2), (batch, meta)]
    res = {}
    try:
        for graph in (False, True):
            tr = _make(0.0, graph)
            tr.fixed_caps = tr.fixed_caps_for(seq)      # (reductions over rows associate by the table capacity: same on both sides)
            losses = [float(tr.training_step(b, meta=mt, cur_epoch=1)["cls"]) for b, mt in seq]
            tr.check_device_flags()
            if graph:
                # (one graph per table shape: the two batches' row counts fall into the same 2048-row bucket or not)
                assert 1 <= len(tr._graphs) <= 2 and sum(g.replays for g in tr._graphs.values()) >= len(seq) - 2 * tr.graph_warmup
            res[graph] = (losses, _state(tr))
    finally:
        ops.use_seed_offset(False)
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.equal(a, b)


def test_graph_replays_draw_fresh_dropout_masks():
    from druglamp_amd import ops
    from druglamp_amd.synthetic import make_batch
    batch, meta = make_batch(8, DEV, seed=7, with_graph=True, llm_dtype=torch.bfloat16)
    try:
        tr = _make(0.1, True)
        tr.set_lrs(0.0, 0.0, 0.0)                  # parameters frozen: only the masks can change the loss
        tr.opt.wd = 0.0
        for _ in range(tr.graph_warmup):
            tr.training_step(batch, meta=meta, cur_epoch=1)
        a = float(tr.training_step(batch, meta=meta, cur_epoch=1)["cls"])       # capture + first replay
        b = float(tr.training_step(batch, meta=meta, cur_epoch=1)["cls"])
        assert a != b, "two replays produced the same loss: the dropout masks did not change"
        g = next(iter(tr._graphs.values()))
        off = ops.seed_offset_tensor(DEV)
        # same offset -> same masks -> same loss and the same gradients (forward / backward mask agreement is what
        # makes the packed gradient reproducible)
        off.fill_(41)
        g.graph.replay()
        l1, g1 = float(g.out["cls"]), tr.flat.grads.clone()
        off.fill_(41)
        g.graph.replay()
        l2, g2 = float(g.out["cls"]), tr.flat.grads.clone()
        assert l1 == l2 and torch.equal(g1, g2)
        assert int(off) == 42
    finally:
        ops.use_seed_offset(False)


def test_cm_steps_replay_a_graph_bit_identical_to_eager_steps():
    """Round 3: steps with the cross-modality head (DrugLAMP2C2P from RS.INIT_EPOCH + 1 on; reference trainer.py:213-221) are
    captured (kind "cm"): the head is shape-static (padded unique-row blocks, masked BatchNorm statistics, (B, B) label
    matrix) and the batch's label matrix is refreshed from the host before each replay.  Dropout off: graphed and eager
    steps leave bit-identical parameters, moments of all optimisers that stepped, BatchNorm buffers and losses — also when
    a replay gets another batch with other ids (other n_p / n_d)."""
    from druglamp_amd import ops
    from druglamp_amd.synthetic import make_batch
    batch, meta = make_batch(8, DEV, seed=7, with_graph=True, llm_dtype=torch.bfloat16)
    other, meta2 = make_batch(8, DEV, seed=8, with_graph=True, llm_dtype=torch.bfloat16)
    meta2 = [dict(m, Prot_ID=m["Prot_ID"] if i % 3 else meta2[0]["Prot_ID"]) for i, m in enumerate(meta2)]   # fewer unique proteins
    seq = [(batch, meta), (batch, meta), (batch, meta), (other, meta2), (batch, meta)]
    res = {}
    try:
        for graph in (False, True):
            tr = _make(0.0, graph, kind="DrugLAMP2C2P")
            tr.cm_weight = 10.0
            tr.fixed_caps = tr.fixed_caps_for(seq)
            outs = []
            for b, mt in seq:
                o = tr.training_step(b, meta=mt, cur_epoch=6)
                outs.append((float(o["cls"]), float(o["cm"])))
            tr.check_device_flags()
            if graph:
                # (graphs are captured at capacity classes: the other batch replays the first graph when it fits its tables)
                assert 1 <= len(tr._graphs) <= 2 and sum(g.replays for g in tr._graphs.values()) >= 1
                for g in tr._graphs.values():
                    assert g.kind == "cm" and g.byval == (tr.model.cm_model.m_sch_loss_fn.margin, 10.0)
            res[graph] = (outs, _state(tr) + (tr.opt_cm.exp_avg.clone(), tr.opt_cm.exp_avg_sq.clone()))
    finally:
        ops.use_seed_offset(False)
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    assert len({o[1] for o in res[True][0]}) > 2          # the head's loss moves (it trains, and step 4 has other labels)
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.equal(a, b)


def test_cm_graphs_follow_the_margin_schedule_and_ssl_epochs():
    """The triplet margin is a by-value argument of the capture: when the schedule moves it (once per epoch) the next CM
    step is eager again, then re-captured, and the graph with the old margin is released.  On an SSL epoch the step kind is
    "sslcm" (SSL forward for the logged loss, its backward dead like in the eager step)."""
    from druglamp_amd import ops
    from druglamp_amd.synthetic import make_batch
    batch, meta = make_batch(8, DEV, seed=7, with_graph=True, llm_dtype=torch.bfloat16)
    try:
        tr = _make(0.0, True, kind="DrugLAMP2C2P")
        for _ in range(4):
            tr.training_step(batch, meta=meta, cur_epoch=6)
        (g0,) = tr._graphs.values()
        assert g0.replays == 2
        tr.on_train_epoch_end(6)                               # margin schedule steps
        m1 = tr.model.cm_model.m_sch_loss_fn.margin
        assert m1 != g0.byval[0]
        for _ in range(4):
            out = tr.training_step(batch, meta=meta, cur_epoch=7)
        (g1,) = tr._graphs.values()                            # the old graph is gone
        assert g1.byval[0] == m1 and g1.replays == 2 and g1 is not g0
        for _ in range(4):
            out = tr.training_step(batch, meta=meta, cur_epoch=10)
            assert set(out) == {"cls", "ssl", "cm"} and all(torch.isfinite(v).all() for v in out.values())
        kinds = sorted(g.kind for g in tr._graphs.values())
        assert kinds == ["cm", "sslcm"], kinds
        assert torch.isfinite(tr.flat.arena).all()
    finally:
        ops.use_seed_offset(False)


def test_cm_steps_stay_eager_in_the_epoch_the_head_starts():
    """cur_epoch == RS.INIT_EPOCH: the cm_weight auto-scale reads losses on the host (trainer.py:216-220 of the reference)."""
    from druglamp_amd import ops
    from druglamp_amd.synthetic import make_batch
    batch, meta = make_batch(8, DEV, seed=7, with_graph=True, llm_dtype=torch.bfloat16)
    try:
        tr = _make(0.1, True, kind="DrugLAMP2C2P")
        for ep in (1, 1, 1, 5, 1):
            torch.manual_seed(50 + ep)
            out = tr.training_step(batch, meta=meta, cur_epoch=ep)
            assert all(torch.isfinite(v).all() for v in out.values())
            assert ("cm" in out) == (ep >= tr.cm_init_epoch)
        assert len(tr._graphs) == 1 and next(iter(tr._graphs.values())).replays == 2
    finally:
        ops.use_seed_offset(False)


@pytest.mark.gpu
def test_captured_scratch_survives_growth_of_the_shared_workspace():
    """A hipGraph-captured split-K GEMM must not point into the shared scratch buffer of ops.py: that buffer is replaced
    when a later EAGER call needs more room, and a replay would then write its slabs through the freed address (found by
    tools/soak.py --graph).  Detection: a victim tensor takes the freed block; a replay must leave it untouched."""
    import torch
    from druglamp_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    K, M, N = 8192, 256, 256
    dy = (torch.randn(K, M, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    x = (torch.randn(K, N, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    out = torch.zeros(M, N, device=dev)
    ops._ws.bufs.clear()
    ops._ws2.bufs.clear()
    run = lambda: ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0, out=out)   # noqa: E731
    run()
    ref = out.clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        run()
    (old,) = ops._ws.bufs.values()
    n0, ptr0 = old.numel(), old.data_ptr()
    del old
    # an eager call with a far larger scratch need replaces the shared buffer ...
    big_dy = torch.zeros(65536, 1024, dtype=torch.bfloat16, device=dev)
    big_x = torch.zeros(65536, 1024, dtype=torch.bfloat16, device=dev)
    ops.gemm(big_dy, big_x, M=1024, N=1024, K=65536, x_kslow=True, w_kslow=True, ldx=1024, ldw=1024, out_dtype=torch.float32, split_k=0)
    torch.cuda.synchronize()
    # ... and the freed block goes to the next tensor of that size
    victim = torch.full((n0,), 0x5A, dtype=torch.uint8, device=dev)
    took_it = victim.data_ptr() == ptr0
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        assert bool((victim == 0x5A).all()), "a replay wrote through the stale scratch address"
    if not took_it:
        pytest.skip("the allocator did not hand the freed block to the victim: inconclusive on this run")


def test_replays_survive_new_weight_images_registered_after_capture():
    """A captured step's dl_weight_prep node reads its item table on every replay.  When images are registered AFTER the
    capture (the SSL / CM heads at their first epoch) the table is rebuilt; the old one must stay alive, or the replay reads
    whatever took its place (tools/soak.py --graph: memory access fault).  Detection: zero-filled victim tensors of the old
    tables' sizes are allocated right after the rebuild — a replay that reads them refreshes no image, and the losses leave
    the eager sequence."""
    from druglamp_amd import functional as Fn
    from druglamp_amd import ops
    from druglamp_amd.synthetic import make_batch
    batch, meta = make_batch(8, DEV, seed=7, with_graph=True, llm_dtype=torch.bfloat16)
    res = {}
    try:
        for graph in (False, True):
            tr = _make(0.0, graph)
            losses = [float(tr.training_step(batch, meta=meta, cur_epoch=1)["cls"]) for _ in range(4)]
            sizes = [(t[1].numel(), t[2].numel()) for t in Fn._lowp_tables.values()]
            extra = torch.nn.Parameter(torch.randn(64, 64, device=DEV))
            Fn.lowp((extra,), torch.bfloat16)              # a new planned image: the next refresh rebuilds the table
            Fn.bump_param_epoch()
            Fn.lowp((extra,), torch.bfloat16)
            torch.cuda.synchronize()
            victims = [torch.zeros(n, dtype=torch.uint8, device=DEV) for a, b in sizes for n in (a, a, a)] + \
                      [torch.zeros(b, dtype=torch.int32, device=DEV) for a, b in sizes for _ in range(3)]
            losses += [float(tr.training_step(batch, meta=meta, cur_epoch=1)["cls"]) for _ in range(3)]
            del victims
            res[graph] = (losses, _state(tr))
    finally:
        ops.use_seed_offset(False)
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.equal(a, b)


def test_capture_right_after_an_evaluation_still_refreshes_the_weight_images():
    """ADVICE (round 2): if an eager forward with no optimiser step behind it (evaluate()) runs between the warm-up steps
    and the capturing step, the weight images are current at capture time; the capture must still record the
    dl_weight_prep refresh, or every replay computes with the images frozen at capture while AdamW moves the masters."""
    from druglamp_amd import ops
    from druglamp_amd.synthetic import make_batch
    batch, meta = make_batch(8, DEV, seed=7, with_graph=True, llm_dtype=torch.bfloat16)
    res = {}
    try:
        for graph in (False, True):
            tr = _make(0.0, graph)
            losses = []
            for step in range(6):
                if step == tr.graph_warmup:          # right before the step that captures
                    ev = tr.evaluate([batch])
                    assert ev["loss"] == ev["loss"]
                losses.append(float(tr.training_step(batch, meta=meta, cur_epoch=1)["cls"]))
            if graph:
                assert len(tr._graphs) == 1
            res[graph] = (losses, _state(tr))
    finally:
        ops.use_seed_offset(False)
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for a, b in zip(res[False][1], res[True][1]):
        assert torch.equal(a, b)


def test_pointer_audit_of_a_captured_step_finds_only_pool_or_pinned_memory():
    """DL_GRAPH_PTR_AUDIT=1 (a debug mode: every library call goes through a Python shim): each device pointer that a
    captured launch received must lie in the graph's private pool or in a buffer pinned for the trainer's life — the
    systematic form of the two use-after-free fixes of round 2.  Runs in a subprocess (the flag is read at import)."""
    import os
    import subprocess
    import sys
    code = r'''
import torch
from druglamp_amd import ops
from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
from druglamp_amd.model import MInterface
from druglamp_amd.synthetic import make_batch
from druglamp_amd.trainer import Trainer
dev = torch.device("cuda", 0)
torch.manual_seed(1); ops.manual_seed(2)
cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16, graph_steps=True)
tr.set_lrs(1e-3, 1e-3, 1e-3)
batch, meta = make_batch(8, dev, seed=7, with_graph=True, llm_dtype=torch.bfloat16)
for _ in range(5):
    out = tr.training_step(batch, meta=meta, cur_epoch=1)
g = next(iter(tr._graphs.values()))
n, n_pool, n_pin = g.audit
assert n > 500 and n_pool > 0 and n_pin > 0, g.audit
assert float(out["cls"]) == float(out["cls"])
victim = torch.zeros(1024, device=dev)          # an eagerly allocated buffer nobody pinned: the audit must reject it
try:
    g._audit_pointers([("dl_fake", "arg0", victim.data_ptr())])
    raise SystemExit("audit accepted an unpinned default-pool pointer")
except RuntimeError as e:
    assert "neither the graph's pool nor a pinned buffer" in str(e)
assert g._audit_pointers([("dl_fake", "arg0", tr.flat.arena.data_ptr() + 64)]) == (1, 0, 1)
print("AUDIT", n, n_pool, n_pin)
'''
    env = dict(os.environ, DL_GRAPH_PTR_AUDIT="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "AUDIT" in r.stdout


def test_ssl_epoch_steps_replay_a_graph_with_fresh_masks():
    """Round 3: steps of an SSL epoch (DrugLAMP: cls forward + SimSiam + masked-LM heads, reference trainer.py:192-212) are
    captured too (kind "ssl").  The MLM mask is drawn on the device inside the graph: with lr = 0 two replays must give
    DIFFERENT ssl losses (fresh masks per replay) but the same cls loss; with training on, graphed and eager runs must
    both stay finite and reach comparable ssl losses (the mask streams differ, so no bit comparison)."""
    from druglamp_amd import ops
    from druglamp_amd.synthetic import make_batch
    batch, meta = make_batch(8, DEV, seed=7, with_graph=True, llm_dtype=torch.bfloat16)
    try:
        tr = _make(0.0, True)
        tr.set_lrs(0.0, 0.0, 0.0)
        outs = []
        for _ in range(5):                                   # (a replay returns the graph's own output tensors: read them now)
            o = tr.training_step(batch, meta=meta, cur_epoch=5)
            outs.append({k: float(v) for k, v in o.items()})
        (g,) = [v for v in tr._graphs.values() if v.kind == "ssl"]
        assert g.kind == "ssl" and g.replays == 3
        ssl = [o["ssl"] for o in outs[2:]]
        cls = [o["cls"] for o in outs[2:]]
        assert len(set(ssl)) == 3, ssl                      # three replays, three different mask draws
        assert len(set(cls)) == 1, cls                      # ... over the same weights and inputs
        # cls steps of the same trainer get their own graph
        tr.training_step(batch, meta=meta, cur_epoch=1); tr.training_step(batch, meta=meta, cur_epoch=1)
        tr.training_step(batch, meta=meta, cur_epoch=1)
        assert len(tr._graphs) == 2
        res = {}
        for graph in (False, True):
            t2 = _make(0.0, graph)
            res[graph] = [float(t2.training_step(batch, meta=meta, cur_epoch=5)["ssl"]) for _ in range(8)]
            assert torch.isfinite(t2.flat.arena).all()
        assert all(v == v for v in res[True]) and abs(res[True][-1] - res[False][-1]) <= 0.25 * abs(res[False][-1]), res
    finally:
        ops.use_seed_offset(False)
