"""world_size-2 gloo test of the data-parallel gradient path (trainer.FlatParams + per-run all-reduce)
on CPU tensors: after the reduction every rank holds the SUM of both ranks' gradients for exactly the
parameters that had one, laid out in the flat buffer the fused AdamW kernel consumes."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from druglamp_amd.trainer import FlatParams, Trainer
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s)) for s in [(8, 4), (5,), (3, 3), (16,)]]
    t = Trainer.__new__(Trainer)
    t.flat = FlatParams(ps)
    t.world, t.rank = world, rank
    for i in (0, 1, 3):                                   # parameter 2 has no gradient on any rank
        ps[i].grad = torch.full_like(ps[i], float(rank + 1) * (i + 1))
    idx = t._reduce_and_pack()
    ok = idx == [0, 1, 3]
    for i in idx:
        ok &= bool(torch.allclose(t.flat.grad_views[i], torch.full_like(ps[i], 3.0 * (i + 1))))
    ok &= float(t.flat.grad_views[2].abs().sum()) == 0.0
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from druglamp_amd import dist_ops
    torch.manual_seed(rank)
    x = torch.randn(3, 4, requires_grad=True)
    full = dist_ops.all_gather_rows(x)
    ok = tuple(full.shape) == (6, 4) and bool(torch.equal(full[rank * 3:(rank + 1) * 3], x.detach()))
    # every rank evaluates the same "global" loss with rank-dependent upstream weights
    w = torch.arange(24.0).view(6, 4) * (rank + 1)
    (full * w).sum().backward()
    expect = (torch.arange(24.0).view(6, 4) * 3.0)[rank * 3:(rank + 1) * 3]      # sum over ranks of w, local rows
    ok &= bool(torch.allclose(x.grad, expect))
    meta = dist_ops.all_gather_meta([{"Prot_ID": "p%d" % rank, "Drug_ID": rank, "Y": 1.0}])
    ok &= [m["Prot_ID"] for m in meta] == ["p0", "p1"]
    # round 4: the global-batch CM head gathers INTEGER id codes as one tensor collective and builds the label matrix of the
    # gathered batch on the device — equal to the host label matrix of the concatenated records on every rank
    from druglamp_amd.model.cross_modality import DeviceLabels, label_matrix
    # (mixed id types, ADVICE r4: rank 0 holds drug ids as Python ints, rank 1 the same ids as numpy.int64 — one entity each)
    import numpy as np
    allmeta = [{"Prot_ID": "p%d" % (i % 3), "Drug_ID": (i * 5) % 4 if i < 4 else np.int64((i * 5) % 4), "Y": float(i % 2)} for i in range(8)]
    mine = allmeta[rank * 4:(rank + 1) * 4]
    codes = dist_ops.all_gather_codes(torch.from_numpy(dist_ops.id_codes(mine)))
    ok &= tuple(codes.shape) == (8, 3)
    lab = DeviceLabels(codes, use_cm=True)
    pidx, didx, gt = label_matrix(allmeta, True)
    n_p, n_d = len(pidx), len(didx)
    ok &= int(lab.n[0]) == n_p and int(lab.n[1]) == n_d and lab.idx[0][:n_p].tolist() == pidx and lab.idx[1][:n_d].tolist() == didx
    ok &= bool((lab.gt[:n_p, :n_d].numpy() == gt).all())
    q.put((rank, ok))
    dist.destroy_process_group()


def test_all_gather_rows_autograd_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from druglamp_amd.trainer import FlatParams, GradOverlap
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                              torch.nn.Linear(32, 3))
    unused = torch.nn.Parameter(torch.randn(7))                       # never reaches the loss
    params = list(net.parameters())[:3] + [unused] + list(net.parameters())[3:]
    flat = FlatParams(params)
    ov = GradOverlap(flat, bucket_bytes=1024)                           # several buckets on this toy
    ok, launched_early = True, []
    head = lambda t: net[2](net[1](net[0](t)))                         # rank 1 leaves the last layer out in step 2
    for step in range(3):
        g = torch.Generator().manual_seed(10 * step + rank)
        x = torch.randn(5, 6, generator=g)
        for p in params:
            p.grad = None
        partial = step == 2 and rank == 1
        loss = (head(x) if partial else net(x)).square().sum()
        ov.arm("cls")
        loss.backward()
        launched_early.append(len(ov.reduced))
        idx = ov.finish()
        ok &= idx == [0, 1, 2, 4, 5, 6]                                 # the agreed set, whatever this rank produced
        # what both ranks' gradients sum to, recomputed locally from both ranks' inputs
        want = [torch.zeros_like(p) for p in params]
        for r in range(world):
            xr = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 * step + r))
            part = step == 2 and r == 1
            on = [i for i in idx if not (part and i >= 5)]              # parameters 5, 6 = the last Linear
            gs = torch.autograd.grad((head(xr) if part else net(xr)).square().sum(), [params[i] for i in on])
            for i, gi in zip(on, gs):
                want[i] += gi
        for i in idx:
            ok &= bool(torch.allclose(flat.grad_views[i], want[i], rtol=1e-5, atol=1e-6))
        ok &= float(flat.grad_views[3].abs().sum()) == 0.0
    # pass 1 agrees on the set, later ones overlap; rank 1's first bucket never completes in step 2 (strict order: finish() sends it)
    ok &= launched_early[0] == 0 and launched_early[1] > 0 and (launched_early[2] > 0 or rank == 1)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_overlapped_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _loss_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from druglamp_amd.trainer import _epoch_loss_means
    # rank 0 ran 2 steps, rank 1 ran 3 (uneven shards): the epoch mean is over all 5 steps of both ranks
    sums = {"cls": torch.tensor(4.0 if rank == 0 else 9.0), "ssl": torch.tensor(1.0 if rank == 0 else 1.5)}
    out = _epoch_loss_means(sums, 2 if rank == 0 else 3, world, "cpu")
    ok = abs(out["train_loss"] - 13.0 / 5) < 1e-12 and abs(out["ssl_loss"] - 0.5) < 1e-12 and out["cm_loss"] == 0.0
    ok &= abs(out["all_loss"] - (13.0 / 5 + 0.5)) < 1e-12
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_epoch_loss_means_world2():
    """The epoch-level loss means of Trainer.fit (the reference's sync_dist=True logging) over two ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loss_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
