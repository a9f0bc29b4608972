"""Two data-parallel ranks (gloo rendezvous, both on cuda:0): hipGraph-replayed cls steps (graph = forward + backward + packing;
gradient all-reduce and AdamW outside, trainer.GraphedStep) against eager steps — with dropout off the parameter arenas must
be bit-identical between the two modes and across the ranks, including a step that follows a different (eager) step kind.
Replica synchronisation at construction is exercised too: rank 1 starts from different random weights."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
EPOCHS = (1, 1, 1, 1, 5, 1)


def _worker(rank, world, port, graph, q, grad_bf16=False):
    os.environ["DL_GRAD_BF16"] = "1" if grad_bf16 else "0"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from druglamp_amd import ops
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface
    from druglamp_amd.synthetic import make_batch
    from druglamp_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    torch.manual_seed(1234 + 17 * rank)                   # DIFFERENT initial weights per rank: Trainer must broadcast rank 0's
    ops.manual_seed(1000 + rank)
    cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
    model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
    model.pmma.p_drop = 0.0
    model.pmma.embeddings.p_drop = 0.0
    tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16, graph_steps=graph)
    assert tr.replicas_in_sync()
    tr.set_lrs(1e-3, 1e-3, 1e-3)
    batch, meta = make_batch(8, dev, seed=100 + rank, with_graph=True, llm_dtype=torch.bfloat16)
    for ep in EPOCHS:
        torch.manual_seed(77 + ep)
        tr.training_step(batch, meta=meta, cur_epoch=ep)
    torch.cuda.synchronize()
    replays = sum(g.replays for g in tr._graphs.values())
    q.put((rank, tr.flat.arena.detach().cpu().numpy(), replays, tr.replicas_in_sync()))
    ops.use_seed_offset(False)
    dist.destroy_process_group()


def _run(graph, grad_bf16=False):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, graph, q, grad_bf16)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    return res


def test_graphed_data_parallel_steps_equal_eager_ones_on_two_ranks():
    eager, graphed = _run(False), _run(True)
    assert np.isfinite(eager[0][1]).all()
    assert np.array_equal(eager[0][1], eager[1][1]) and np.array_equal(graphed[0][1], graphed[1][1])   # replicas identical
    assert eager[0][3] and graphed[0][3]
    assert np.array_equal(eager[0][1], graphed[0][1])                                                  # graph == eager
    assert graphed[0][2] == 3 and eager[0][2] == 0          # cls steps 3, 4 and 6 were replays (2 eager warm-up steps first)


def test_bf16_gradient_buckets_keep_replicas_identical_and_drift_little():
    """DL_GRAD_BF16=1: the gradient all-reduce moves bf16 (half the bytes); the optimiser still reads fp32 and updates fp32
    masters.  Replicas must stay bit-identical to each other; against the fp32 reduction the parameters after six steps
    (lr 1e-3, AdamW: an update is lr-sized whatever the gradient's scale) differ by a small fraction of what the steps moved."""
    f32, b16 = _run(True), _run(True, grad_bf16=True)
    assert np.array_equal(b16[0][1], b16[1][1]) and b16[0][3]
    moved = np.abs(f32[0][1]).max()
    d = np.abs(b16[0][1] - f32[0][1])
    assert np.isfinite(b16[0][1]).all()
    assert d.max() <= 2e-2, d.max()                # a few lr-sized steps at worst (elements whose tiny gradient changes sign)
    assert d.mean() <= 0.15 * 1e-3 and moved > 0, d.mean()   # on average a fraction of ONE step


def _worker_nccl1(port, q, force):
    """One rank on the RCCL backend: DL_GRAPH_ALLREDUCE=force makes the captured step contain its gradient all-reduce."""
    os.environ["DL_GRAPH_ALLREDUCE"] = "force" if force else "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    from druglamp_amd import ops
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface
    from druglamp_amd.synthetic import make_batch
    from druglamp_amd.trainer import Trainer
    torch.manual_seed(1234)
    ops.manual_seed(1000)
    cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
    model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
    model.pmma.p_drop = 0.0
    model.pmma.embeddings.p_drop = 0.0
    tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16, graph_steps=True)
    tr.set_lrs(1e-3, 1e-3, 1e-3)
    batch, meta = make_batch(8, dev, seed=100, with_graph=True, llm_dtype=torch.bfloat16)
    for _ in range(6):
        tr.training_step(batch, meta=meta, cur_epoch=1)
    torch.cuda.synchronize()
    g = next(iter(tr._graphs.values()))
    q.put((tr.flat.arena.detach().cpu().numpy(), bool(g.reduced), g.replays))
    ops.use_seed_offset(False)
    dist.destroy_process_group()


def test_a_captured_step_may_contain_its_rccl_all_reduce():
    """DL_GRAPH_ALLREDUCE: the gradient all-reduce as nodes of the replayed graph (RCCL supports stream capture).  One rank on
    the nccl backend — the capture / replay path of the collective, not the multi-GPU transport: parameters after six steps
    must equal those of the default mode (all-reduce issued after the replay; a no-op sum at world size 1)."""
    res = []
    for force in (False, True):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        p = ctx.Process(target=_worker_nccl1, args=(port, q, force))
        p.start()
        res.append(q.get(timeout=600))
        p.join(60)
    assert res[0][1] is False and res[1][1] is True and res[0][2] == res[1][2] == 4
    assert np.array_equal(res[0][0], res[1][0])


def _worker_c3(rank, world, port, q):
    """BASELINE config C3 in miniature: DrugLAMP, SSL epoch, NT-Xent (simclr) over the ALL-GATHERED batch, two ranks."""
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from druglamp_amd import ops
    from druglamp_amd.configs import get_cfg_defaults, load_yaml_into
    from druglamp_amd.model import MInterface
    from druglamp_amd.synthetic import make_batch
    from druglamp_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    ops.manual_seed(1000 + rank)
    cfg = load_yaml_into(get_cfg_defaults(), "DrugLAMP")
    cfg["RS"]["GLOBAL_BATCH"] = True
    cfg["RS"]["DRUG_SSL_TYPE"] = "simclr"
    model = MInterface("DrugLAMP", cfg).load_model(n_drug_feature=384, n_prot_feature=640).to(dev)
    tr = Trainer(model, cfg, device=dev, compute_dtype=torch.bfloat16, graph_steps=False)
    tr.set_lrs(1e-3, 1e-3, 1e-3)
    batch, meta = make_batch(4, dev, seed=100 + rank, with_graph=True, llm_dtype=torch.bfloat16)
    losses = []
    for ep in (1, 5, 5, 1):
        out = tr.training_step(batch, meta=meta, cur_epoch=ep)
        losses.append({k: float(v) for k, v in out.items()})
    torch.cuda.synchronize()
    q.put((rank, losses, tr.replicas_in_sync(), bool(torch.isfinite(tr.flat.arena).all())))
    dist.destroy_process_group()


def test_config_c3_step_global_batch_ntxent_on_two_ranks():
    """DrugLAMP with RS.GLOBAL_BATCH + RS.DRUG_SSL_TYPE = simclr on two ranks: the SSL-epoch steps run the NT-Xent head over
    the all-gathered node rows (2 x 4 x 512 rows per half), gradients are reduced, replicas stay bit-identical."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_c3, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for rank, losses, in_sync, finite in res:
        assert in_sync and finite
        assert "ssl" in losses[1] and "ssl" in losses[2] and "ssl" not in losses[0]
        assert all(v == v for l in losses for v in l.values())
    # the ranks see different pairs: their (rank-mean) NT-Xent values differ, but both are losses over the same 2 x 4096 columns
    assert res[0][1][1]["ssl"] != res[1][1][1]["ssl"]
