"""Global-batch cross-modal head on two ranks (gloo rendezvous, both ranks on cuda:0): with global_batch=True every
rank evaluates the triplet loss of the WHOLE batch from all-gathered token means (dist_ops.all_gather_rows), so the
loss and the head's parameter gradients equal the single-process full-batch ones, and the input gradients are
world x the full-batch gradient of the local rows (the DP step's 1/world averaging then restores them)."""
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N, L, H = 8, 6, 128


def _inputs():
    g = torch.Generator().manual_seed(7)
    ts = [torch.randn(N, L, H, generator=g) for _ in range(4)]
    meta = [{"Prot_ID": "p%d" % (i % 5), "Drug_ID": "d%d" % (i % 6), "Y": float((i * 7) % 3 == 0)} for i in range(N)]
    return ts, meta


def _head(global_batch):
    from druglamp_amd.model.cross_modality import CrossModality
    torch.manual_seed(11)
    m = CrossModality(use_cm=True, hidden_size=H, max_margin=0.5, n_re=100, global_batch=global_batch).cuda()
    m.eval()                      # BatchNorm with running statistics: the comparison is about the gather, not BN
    return m


def _worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ts, meta = _inputs()
    n = N // world
    loc = [t[rank * n:(rank + 1) * n].cuda().requires_grad_(True) for t in ts]
    m = _head(True)
    loss = m(*loc, meta=meta[rank * n:(rank + 1) * n])
    loss.backward()
    q.put((rank, float(loss.detach()), [p.grad.detach().cpu().numpy() for p in m.parameters() if p.grad is not None],
           [t.grad.detach().cpu().numpy() for t in loc]))          # numpy: plain pickles, no shared-memory handles
    dist.destroy_process_group()


def test_global_batch_cm_matches_full_batch():
    ts, meta = _inputs()
    full = [t.cuda().requires_grad_(True) for t in ts]
    m = _head(False)
    loss = m(*full, meta=meta)
    loss.backward()
    ref_pg = [p.grad.detach().cpu() for p in m.parameters() if p.grad is not None]
    ref_ig = [t.grad.detach().cpu() for t in full]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
    n = N // 2
    for rank, l, pg, ig in res:
        assert abs(l - float(loss.detach())) <= 1e-5 * max(1.0, abs(float(loss.detach())))
        for a, b in zip(pg, ref_pg):
            a = torch.from_numpy(a)
            assert (a - b).abs().max() <= 1e-5 * max(1e-6, float(b.abs().max())) + 1e-7
        for a, b in zip(ig, ref_ig):
            a = torch.from_numpy(a)
            assert (a - 2.0 * b[rank * n:(rank + 1) * n]).abs().max() <= 1e-5 * max(1e-6, float(b.abs().max())) + 1e-7
