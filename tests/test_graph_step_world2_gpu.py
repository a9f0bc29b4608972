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


def _worker(rank, world, port, graph, q):
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


def _run(graph):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, graph, q)) for r in range(2)]
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
