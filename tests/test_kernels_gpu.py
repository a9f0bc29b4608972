"""Kernel-level parity through the C ABI against fp64 torch references (same battery as
tools/gpu_probe.py): GEMM layouts/epilogues, LayerNorm, attention fwd/bwd incl. paired segments and
masked tails, MHLA gate, elementwise, AdamW, loss kernels."""
import importlib.util
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe():
    spec = importlib.util.spec_from_file_location("gpu_probe", os.path.join(ROOT, "tools", "gpu_probe.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("group", ["gemm", "ln", "attn", "misc", "loss"])
def test_kernel_group(group):
    p = _probe()
    p.RESULTS.clear()
    {"gemm": p.gemm_cases, "ln": p.ln_cases, "attn": p.attn_cases, "misc": p.misc_cases, "loss": p.loss_cases}[group]()
    bad = [r for r in p.RESULTS if not r[3]]
    assert len(p.RESULTS) > 5 and not bad, bad[:5]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,odt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                       (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("site_len,S,F", [(9, 2304, 640), (1, 512, 384), (3, 12, 8)])
def test_fill_pool_matches_reference_formulation(dtype, odt, site_len, S, F):
    """fill bit + site pooling in one pass == reference DrugLAMP.py:11-19,39-40 (cat fill bit, view, mean)."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(5)
    B = 3
    x = torch.randn(B, S, F, generator=g)
    x[0, S // 2:] = 0                      # padded tail -> fill bit 1
    x[2, 1] = 0
    x = x.to(dtype).cuda()
    fill, pooled = ops.fill_pool(x, site_len, odt)
    xf = x.float()
    ref_fill = (xf.sum(-1) == 0).float()
    cat = torch.cat((xf, ref_fill.unsqueeze(-1)), -1)
    ref = cat.view(B, site_len, S // site_len, F + 1).mean(1)
    assert torch.equal(fill.float(), ref_fill)
    Fp = (F + 1 + 7) // 8 * 8
    assert pooled.shape == (B, S // site_len, Fp) and pooled.dtype == odt
    tol = 1e-6 if odt == torch.float32 else 8e-3
    torch.testing.assert_close(pooled[..., :F + 1].float(), ref, rtol=tol, atol=tol)
    assert torch.count_nonzero(pooled[..., F + 1:]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(49152, 512, 256), (49000, 520, 192), (8192, 512, 2048), (8100, 264, 1024), (16384, 256, 512),
                                   (65536, 1024, 256), (49152, 1536, 512), (33024, 640, 512), (57344, 512, 1024)])
def test_large_tile_gemm_is_bitwise_equal_to_the_128_tile_path(M, N, K):
    """The 256x256 persistent kernel (gemm_big.cuh), its few-tile 128x128 deep-ring form (shapes 3-5: at most
    256 tiles, K >= 512 — the strong-scaling batches) and gemm_kernel accumulate every output element in the same
    k order and share epilogue math and dropout counters: outputs must be identical bit for bit, for every
    specialised epilogue, including ragged M / N edges.  (Shapes 6-9 were added with round 6's trickle kernel — study library
    only, tools/trickle_bench.py checks it bit for bit there — and stay as further large-tile cases: N = 1024 / 1536 / 640, a
    partial last round of tiles.)"""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(11)
    dt = torch.bfloat16
    x = (torch.randn(M, K, generator=g) * 0.5).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.1).to(dt).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(dt).cuda()
    pre_in = torch.randn(M, N, generator=g).to(dt).cuda()
    cases = {
        "plain": dict(), "bias": dict(bias=b), "relu": dict(bias=b, act=2),
        "gelu+pre+drop": dict(bias=b, act=1, pre_out=True, dropout_p=0.1, seed=5),
        "dgelu+drop": dict(dact_pre=pre_in, dropout_p=0.1, seed=5),
        "res+drop": dict(bias=b, residual=res, dropout_p=0.1, seed=7),
    }
    for name, kw in cases.items():
        got = {}
        for mode in ("0", "1"):
            k2 = dict(kw, algo=1 if mode == "0" else 0)       # DL_GEMM_ALGO_TILE128 / AUTO (an explicit argument)

            pre = None
            if k2.get("pre_out"):
                pre = torch.zeros(M, N, device="cuda", dtype=dt)
                k2["pre_out"] = pre
            out = torch.full((M, N), 7.0, device="cuda", dtype=dt)
            ops.gemm(x, w, M=M, N=N, K=K, out=out, **k2)
            torch.cuda.synchronize()
            got[mode] = (out, pre)
        assert torch.equal(got["0"][0], got["1"][0]), name
        if got["0"][1] is not None:
            assert torch.equal(got["0"][1], got["1"][1]), name
    # and against an fp64 reference for the plain case
    ref = x[:512].double() @ w.double().t()
    out = ops.gemm(x, w, M=M, N=N, K=K)
    assert (out[:512].double() - ref).abs().max() <= 2e-2 * ref.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(1536, 512, 16384), (128, 768, 262147), (1280, 520, 9000)])
def test_large_tile_weight_gradient_matches_fp64_and_the_128_tile_path(M, N, K):
    """gemm_big_tt_kernel (both operands K-slow, split-K slabs, zero-page K tail) against an fp64 reference on a
    row sample and against the 128-tile split-K path (different slab count -> fp32 summation order only)."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(3)
    dt = torch.bfloat16
    dy = (torch.randn(K, M, generator=g) * 0.5).to(dt).cuda()
    x = (torch.randn(K, N, generator=g) * 0.5).to(dt).cuda()
    outs = {}
    for mode in ("0", "1"):
        outs[mode] = ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N,
                              out_dtype=torch.float32, split_k=0, algo=1 if mode == "0" else 0).clone()
        torch.cuda.synchronize()
    ref = dy[:, :96].double().t() @ x.double()
    scale = ref.abs().max()
    for mode in ("0", "1"):
        assert (outs[mode][:96].double() - ref).abs().max() <= 5e-6 * scale, mode
    assert (outs["0"] - outs["1"]).abs().max() <= 1e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,dt", [(1536, 512, 16384, torch.bfloat16), (256, 128, 8192, torch.bfloat16),
                                      (768, 256, 4100, torch.bfloat16), (128, 80, 3000, torch.float32),
                                      (128, 768, 262147, torch.bfloat16)])
def test_weight_gradient_gemm_emits_bias_gradient(M, N, K, dt):
    """x_colsum: the column sums of the K-slow X operand (bias gradient) come out of the same launch, for the
    128-tile split-K kernel (DMA and register-staged, f32 and bf16) and the large-tile kernel."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(9)
    dy = (torch.randn(K, M, generator=g) * 0.5).to(dt).cuda()
    x = (torch.randn(K, N, generator=g) * 0.5).to(dt).cuda()
    db = torch.full((M,), 123.0, device="cuda")
    dw = ops.gemm(dy, x, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0,
                  x_colsum=db)
    ref_db = dy.double().sum(0)
    assert (db.double() - ref_db).abs().max() <= 2e-6 * max(1.0, float(ref_db.abs().max())) + 1e-3 * (dt == torch.bfloat16) * 0
    ref = dy[:, :64].double().t() @ x.double()
    assert (dw[:64].double() - ref).abs().max() <= 5e-6 * ref.abs().max()


@pytest.mark.gpu
def test_weight_images_are_refreshed_in_place_by_one_launch():
    """lowp(): compute-dtype / transposed / concatenated weight images keep their addresses and follow the fp32
    masters after bump_param_epoch() (ragged shapes included)."""
    from druglamp_amd import functional as Fn
    g = torch.Generator().manual_seed(2)
    ps = [torch.nn.Parameter(torch.randn(r, c, generator=g).cuda()) for r, c in [(128, 256), (75, 128), (256, 641), (64, 256), (192, 256)]]
    def expect(params, dt, tr):
        w = torch.cat([p.detach() for p in params], 0)
        return (w.t() if tr else w).to(dt)
    reqs = [((ps[0],), torch.bfloat16, False), ((ps[0],), torch.bfloat16, True), ((ps[1],), torch.bfloat16, True),
            ((ps[2],), torch.bfloat16, False), ((ps[2],), torch.bfloat16, True), ((ps[0], ps[3], ps[4]), torch.bfloat16, False),
            ((ps[0], ps[3], ps[4]), torch.bfloat16, True), ((ps[1],), torch.float32, True)]
    imgs = [Fn.lowp(p, dt, tr) for p, dt, tr in reqs]
    for (p, dt, tr), im in zip(reqs, imgs):
        assert torch.equal(im, expect(p, dt, tr))
    ptrs = [im.data_ptr() for im in imgs]
    with torch.no_grad():
        for p in ps:
            p.mul_(1.5).add_(0.25)          # version bump
    again = [Fn.lowp(p, dt, tr) for p, dt, tr in reqs]
    assert [im.data_ptr() for im in again] == ptrs
    for (p, dt, tr), im in zip(reqs, again):
        assert torch.equal(im, expect(p, dt, tr))
    for p in ps:                             # raw-pointer style update + epoch bump
        p.data.copy_(torch.randn(p.shape, generator=g))
    Fn.bump_param_epoch()
    for (p, dt, tr) in reqs:
        assert torch.equal(Fn.lowp(p, dt, tr), expect(p, dt, tr))


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,C,S", [(3, 2304, 128, 9), (2, 96, 64, 3), (2, 160, 128, 5),
                                     (3, 2304, 128, 1), (2, 100, 64, 1)])       # S = 1: the view reinterpretation alone (the masked-LM pass)
def test_cnn_view_and_site_pool_kernel_matches_reference_formulation(B, L, C, S):
    """dl_cnn_sitepool_fwd/bwd == (channel-first buffer).view(B, L, C).view(B, S, L/S, C).mean(1) and its
    autograd backward (basic_model.py:176-179 + DrugLAMP.py:39-40), incl. the zeroed halo rows of dz."""
    from druglamp_amd import ops
    halo = 4
    g = torch.Generator().manual_seed(4)
    z = torch.randn(B, L + 2 * halo, C, generator=g).to(torch.bfloat16).cuda()
    z[:, :halo] = 0
    z[:, halo + L:] = 0
    zin = z[:, halo:halo + L].float().requires_grad_(True)                  # (B, L, C) channel-last
    ref = zin.transpose(1, 2).contiguous().view(B, L, C).view(B, S, L // S, C).mean(1)
    got = ops.cnn_sitepool_fwd(z, L, halo, S)
    torch.testing.assert_close(got.float(), ref.detach(), rtol=1e-2, atol=1e-2)
    dout = torch.randn(B, L // S, C, generator=g).to(torch.bfloat16).cuda()
    ref.backward(dout.float())
    dz = ops.cnn_sitepool_bwd(dout, L, halo, S)
    assert torch.count_nonzero(dz[:, :halo]) == 0 and torch.count_nonzero(dz[:, halo + L:]) == 0
    torch.testing.assert_close(dz[:, halo:halo + L].float(), zin.grad, rtol=1e-2, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,C,halo", [(2, 128, 64, 0), (3, 2304, 128, 0), (2, 192, 128, 7), (1, 64, 192, 1)])
def test_view_reinterpretation_without_pooling_is_an_exact_transpose(B, L, C, halo):
    """site_len 1 with L and C multiples of 64 takes transpose_view_kernel: pooled[b].view(C, L) == z[b, halo:halo+L].T bit for
    bit (basic_model.py:176-179: the (B, C, L) buffer read as (B, L, C)), the gradient is the transpose back, halo rows zero."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(11)
    z = torch.randn(B, L + 2 * halo, C, generator=g).to(torch.bfloat16).cuda()
    got = ops.cnn_sitepool_fwd(z, L, halo, 1)
    want = z[:, halo:halo + L].transpose(1, 2).contiguous().view(B, L, C)
    assert torch.equal(got, want)
    dout = torch.randn(B, L, C, generator=g).to(torch.bfloat16).cuda()
    dz = ops.cnn_sitepool_bwd(dout, L, halo, 1)
    assert dz.shape == (B, L + 2 * halo, C)
    assert torch.equal(dz[:, halo:halo + L], dout.view(B, C, L).transpose(1, 2))
    if halo:
        assert torch.count_nonzero(dz[:, :halo]) == 0 and torch.count_nonzero(dz[:, halo + L:]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_embed_pad_matches_embedding_cat_pad_and_its_gradient(dt):
    """EmbedPadFn == pad(cat(embedding(ids), fill)) with the embedding gradient of nn.Embedding(padding_idx=0); the
    row count (8 * 2312) makes the one-hot weight-gradient GEMM's k-steps not divide its split count (empty k-ranges
    once started past the operands' last row)."""
    from druglamp_amd import functional as Fn
    g = torch.Generator().manual_seed(6)
    B, L, V, D, halo = 8, 2304, 27, 127, Fn._CNN_HALO
    ids = torch.randint(0, V, (B, L), generator=g).cuda()
    w = torch.randn(V, D, generator=g).to(dt).cuda().requires_grad_(True)
    fill = (torch.rand(B, L, generator=g) > 0.5).to(dt).cuda()
    out = Fn.EmbedPadFn.apply(ids, w, fill, 0)
    ref = torch.nn.functional.pad(torch.cat((w.detach()[ids], fill.unsqueeze(-1)), -1), (0, 0, halo, halo))
    assert torch.equal(out, ref)
    dy = torch.randn(B, L + 2 * halo, D + 1, generator=g).to(dt).cuda()
    out.backward(dy)
    wr = w.detach().float().clone().requires_grad_(True)
    torch.nn.functional.embedding(ids, wr, padding_idx=0).backward(dy[:, halo:halo + L, :D].float())
    tol = 1e-4 if dt == torch.float32 else 2e-2
    assert (w.grad.float() - wr.grad).abs().max() <= tol * wr.grad.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_embedding_store_batches_equal_reference_collate_padding(dt):
    """dl_gather_pad through EmbeddingStore vs oracle/collate.py (pinned to the reference's tail_pad / repeat_pad): pure
    copies, so bit-exact; ragged lengths incl. 1, maxsize, > maxsize, repeated keys; at the real shapes as well."""
    import numpy as np
    from druglamp_amd.embedding_store import EmbeddingStore
    from oracle import collate
    lens = [1, 5, 7, 16, 24, 25, 47, 48, 49, 60]
    xs = collate.ragged_inputs("store.small", lens, 8)
    st = EmbeddingStore(8, dtype=dt)
    for i, a in enumerate(xs):
        st.add("k%d" % i, a)
    keys = ["k3", "k0", "k9", "k3", "k7", "k8", "k1", "k2", "k4", "k5", "k6"]
    sel = [xs[int(k[1:])].astype(np.float32) for k in keys]
    if dt == torch.bfloat16:
        sel = [torch.from_numpy(a).to(dt).float().numpy() for a in sel]
    rep = st.batch(keys, 48, repeat=True).float().cpu().numpy()
    assert np.array_equal(rep, collate.repeat_pad(sel, 48))
    ok = [i for i, a in enumerate(sel) if a.shape[0] <= 48]
    tail = st.batch([keys[i] for i in ok], 48, repeat=False).float().cpu().numpy()
    assert np.array_equal(tail, collate.tail_pad([sel[i] for i in ok], 48))
    # real shapes: 2304 x 640 protein rows (repeat), 512 x 384 drug rows (tail)
    g = torch.Generator().manual_seed(0)
    ps = EmbeddingStore(640, dtype=dt)
    plen = [300, 1022, 2304, 77]
    pe = [torch.randn(n, 640, generator=g) for n in plen]
    for i, e in enumerate(pe):
        ps.add(i, e)
    out = ps.batch([2, 0, 3, 1, 0], 2304, repeat=True)
    for row, k in zip(out, [2, 0, 3, 1, 0]):
        e = pe[k].to(dt).cuda()
        n = plen[k]
        reps = 2304 // n
        assert torch.equal(row[:reps * n], e.repeat(reps, 1)) and not row[reps * n:].any()
    # add() after the store went to the device (finalize is implied by batch) keeps working ...
    extra = torch.randn(11, 640, generator=g)
    ps.add("late", extra)
    out = ps.batch(["late", 1], 2304, repeat=True)
    assert torch.equal(out[0, :11], extra.to(dt).cuda()) and torch.equal(out[1, :1022], pe[1].to(dt).cuda())
    # ... and tail padding refuses a sequence longer than the window, like the reference's tail_pad (utils.py:304-312)
    with pytest.raises(ValueError):
        st.batch(["k9"], 48, repeat=False)


def test_layernorm_bwd_shared_dy_rows_equal_expanded_dy():
    """dy_share = L (gradient of a token mean read in place) against the same gradient expanded to every row."""
    from druglamp_amd import ops
    B, L, D = 5, 12, 512
    g = torch.Generator().manual_seed(3)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(B * L, D, generator=g).cuda().to(dt)
        w = torch.randn(D, generator=g).cuda()
        b = torch.randn(D, generator=g).cuda()
        dyb = torch.randn(B, D, generator=g).cuda().to(dt)
        _, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-6)
        full = dyb.unsqueeze(1).expand(B, L, D).reshape(B * L, D).contiguous()
        dx0, dg0, db0 = ops.layernorm_bwd(full, x, mean, rstd, w)
        dx1, dg1, db1 = ops.layernorm_bwd(dyb, x, mean, rstd, w, dy_share=L)
        assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    with pytest.raises(ValueError):
        ops.layernorm_bwd(dyb, x, mean, rstd, w, dy_share=L + 1)


def test_interleave_streams_equals_cat_and_inverts():
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(5)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(2, 3, 7, 64, generator=g).cuda().to(dt)
        y = ops.interleave_streams(x)
        assert torch.equal(y, torch.cat((x[0], x[1]), dim=-1))
        assert torch.equal(ops.interleave_streams(y, inverse=True), x)


def test_concat2_equals_cat_and_splits_back():
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(6)
    for dt in (torch.float32, torch.bfloat16):
        a = torch.randn(3, 5, 128, generator=g).cuda().to(dt)
        b = torch.randn(3, 5, 64, generator=g).cuda().to(dt)
        y = ops.concat2(a, b)
        assert torch.equal(y, torch.cat((a, b), dim=-1))
        a2, b2 = ops.split2(y, 128)
        assert torch.equal(a2, a) and torch.equal(b2, b)


def test_norm_adjacency_matches_the_torch_formula():
    """dl_norm_adjacency against D^-1/2 A^T D^-1/2 with clamped degrees (the dense restatement of dgl's GraphConv norm)."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(8)
    for n in (128, 37):
        adj = (torch.rand(4, n, n, generator=g) < 0.1).float()
        adj[3].zero_()                                             # isolated nodes: degrees clamp to 1
        adj = adj.cuda()
        dout = adj.sum(-1).clamp(min=1).pow(-0.5)
        din = adj.sum(-2).clamp(min=1).pow(-0.5)
        want = adj.transpose(1, 2) * din.unsqueeze(-1) * dout.unsqueeze(-2)
        got = ops.norm_adjacency(adj, torch.float32)
        assert (got - want).abs().max() <= 1e-6
        got16 = ops.norm_adjacency(adj, torch.bfloat16).float()
        assert (got16 - want).abs().max() <= 4e-3


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("n", [128, 77, 5, 129, 190, 333, 512])
def test_graph_aggregate_matches_batched_product(dt, tol, n):
    """dl_graph_aggregate (MolecularGCN neighbourhood sum on dense batched graphs, reference basic_model.py:591-617) against
    torch.bmm in fp64, forward (ahat) and gradient (ahat^T) forms; virtual nodes n..N-1 pass through unchanged."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(n)
    B, N, C = 5, 512, 128
    ahat = (torch.rand(B, n, n, generator=g) * (torch.rand(B, n, n, generator=g) < 0.2)).to(dt).cuda()
    feat = torch.randn(B, N, C, generator=g).to(dt).cuda()
    for transpose in (False, True):
        out = ops.graph_aggregate(ahat, feat, transpose=transpose)
        a = ahat.double().transpose(1, 2) if transpose else ahat.double()
        ref = torch.bmm(a, feat[:, :n].double())
        assert float((out[:, :n].double() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
        assert torch.equal(out[:, n:], feat[:, n:])
    if n > 190:       # adjacency normalisation beyond the LDS tile (round 3: no torch fallback for any graph size)
        adj = (torch.rand(2, n, n, generator=g) < 0.02).float().cuda()
        dout = adj.sum(-1).clamp(min=1).pow(-0.5)
        din = adj.sum(-2).clamp(min=1).pow(-0.5)
        want = adj.transpose(1, 2) * din.unsqueeze(-1) * dout.unsqueeze(-2)
        assert (ops.norm_adjacency(adj, torch.float32) - want).abs().max() <= 1e-6


@pytest.mark.parametrize("M,N,K", [(49152, 512, 256), (49000, 520, 192), (65536, 1024, 256)])
def test_dynamic_tile_tickets_leave_the_large_tile_gemm_bitwise_unchanged(M, N, K):
    """dl_gemm_args.tile_tickets: the persistent one-workgroup-per-CU kernel draws its later tiles from an atomic counter
    (used while an overlapped all-reduce shares the CUs).  Every output tile is still computed by exactly one workgroup in
    the same k order: outputs are bit-identical, for each specialised epilogue, over repeated launches (the two ticket
    words must be back at zero after every launch)."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(13)
    dt = torch.bfloat16
    x = (torch.randn(M, K, generator=g) * 0.5).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.1).to(dt).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(dt).cuda()
    pre_in = torch.randn(M, N, generator=g).to(dt).cuda()
    cases = {"plain": dict(), "gelu+pre+drop": dict(bias=b, act=1, pre_out=True, dropout_p=0.1, seed=5),
             "dgelu+drop": dict(dact_pre=pre_in, dropout_p=0.1, seed=5), "res": dict(bias=b, residual=res)}
    try:
        for name, kw in cases.items():
            got = {}
            for dyn in (False, True, True, True):
                ops.dynamic_tiles(dyn)
                k2 = dict(kw)
                pre = None
                if k2.get("pre_out"):
                    pre = torch.zeros(M, N, device="cuda", dtype=dt)
                    k2["pre_out"] = pre
                out = torch.full((M, N), 7.0, device="cuda", dtype=dt)
                ops.gemm(x, w, M=M, N=N, K=K, out=out, **k2)
                torch.cuda.synchronize()
                if dyn:
                    assert ops._tickets[("cuda", 0)].tolist() == [0, 0], name
                    assert torch.equal(out, got[False][0]), name
                    if pre is not None:
                        assert torch.equal(pre, got[False][1]), name
                else:
                    got[False] = (out, pre)
    finally:
        ops.dynamic_tiles(False)


@pytest.mark.gpu
def test_deferred_reductions_are_bitwise_equal_to_immediate_ones():
    """dl_reduce_batch (split-K slabs of several weight-gradient GEMMs incl. their bias column sums, and LayerNorm
    dgamma / dbeta partials, all in ONE launch) against the launches dl_gemm / dl_layernorm_bwd make on their own."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(21)
    dt = torch.bfloat16
    K = 8192
    shapes = [(1024, 256), (256, 1024), (1536, 512), (128, 128), (768, 256), (8, 1024)]
    ops_in = [((torch.randn(K, M, generator=g) * 0.5).to(dt).cuda(), (torch.randn(K, N, generator=g) * 0.5).to(dt).cuda()) for M, N in shapes]
    x = torch.randn(K, 512, generator=g).to(dt).cuda()
    dy = torch.randn(K, 512, generator=g).to(dt).cuda()
    gamma = torch.randn(512, generator=g).cuda()
    _, mean, rstd = ops.layernorm_fwd(x, gamma, torch.zeros(512, device="cuda"), 1e-6)

    def run():
        outs = []
        for (M, N), (a, b) in zip(shapes, ops_in):
            db = torch.empty(M, dtype=torch.float32, device="cuda")
            dw = ops.gemm(a, b, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0, x_colsum=db)
            outs += [dw, db]
        dx, dg, dbt = ops.layernorm_bwd(dy, x, mean, rstd, gamma)
        return outs + [dx, dg, dbt]

    ref = [t.clone() for t in run()]
    keep = ops.group_wgrad_max_k, ops.group_wgrad_small_mn    # (grouped weight gradients change the slab counts: own test below)
    try:
        ops.group_wgrad_max_k, ops.group_wgrad_small_mn = 0, 0
        with ops.deferred_reductions():
            got = run()
            assert len(ops._pending) == len(shapes) + 1          # every reduction is queued, none has run
    finally:
        ops.group_wgrad_max_k, ops.group_wgrad_small_mn = keep
    torch.cuda.synchronize()
    for r, o in zip(ref, got):
        assert torch.equal(r, o)
    # accumulate=True is never deferred (nor grouped), and flushes what is queued before it runs
    with ops.deferred_reductions():
        a, b = ops_in[0]
        dw = ops.gemm(a, b, M=1024, N=256, K=K, x_kslow=True, w_kslow=True, ldx=1024, ldw=256, out_dtype=torch.float32, split_k=0)
        ops.gemm(a, b, M=1024, N=256, K=K, x_kslow=True, w_kslow=True, ldx=1024, ldw=256, out_dtype=torch.float32, split_k=0,
                 out=dw, accumulate=True)
        assert not ops._pending
    torch.cuda.synchronize()
    assert torch.equal(dw, ref[0] * 2)


@pytest.mark.gpu
def test_grouped_weight_gradients_match_single_launches():
    """dl_gemm_group (round 3): the weight-gradient products queued inside a deferred_reductions() block leave as one launch
    per 16 with 1-4 slabs each; against the same products as single dl_gemm calls (16-32 slabs): fp32 sums of the same
    bf16 products in another association, so equal to ~1e-6 of the largest entry, column sums (bias gradients) included.
    Covers ragged tiles (M, N not multiples of the 128 x 256 tile), different K inside one group, more than 16 products
    (two launches), outputs that are row slices of a larger tensor, and members the group does not take (K over the
    threshold: launched one by one)."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(22)
    dt = torch.bfloat16
    probs = [(1024, 256, 8192), (256, 1024, 8192), (256, 256, 8192), (256, 512, 8192), (768, 256, 8192), (512, 2048, 8192),
             (1000, 648, 4100), (8, 1024, 8192), (128, 128, 16384), (136, 264, 8191), (2048, 512, 4096), (1536, 512, 8192)]    # (K need not be a multiple of the 64-row step)
    probs = probs + probs[:6]                                   # 18 products: two group launches (128-row tiles)
    probs += [(2048, 512, 16384), (512, 2048, 16384), (512, 512, 16384), (1536, 512, 16384), (520, 1032, 16448)]   # third launch: 256-row tiles
    data = [((torch.randn(K, M, generator=g) * 0.5).to(dt).cuda(), (torch.randn(K, N, generator=g) * 0.5).to(dt).cuda()) for M, N, K in probs]
    big = ((torch.randn(32768, 256, generator=g) * 0.5).to(dt).cuda(), (torch.randn(32768, 512, generator=g) * 0.5).to(dt).cuda())

    def run():
        outs = []
        for i, ((M, N, K), (a, b)) in enumerate(zip(probs, data)):
            db = torch.empty(M, dtype=torch.float32, device="cuda") if i % 3 else None
            whole = torch.empty(M + 16, N, dtype=torch.float32, device="cuda") if i % 4 == 1 else None
            dw = ops.gemm(a, b, M=M, N=N, K=K, x_kslow=True, w_kslow=True, ldx=M, ldw=N, out_dtype=torch.float32, split_k=0, x_colsum=db,
                          out=None if whole is None else whole[8:8 + M])
            outs += [dw] + ([db] if db is not None else [])
            if i == 17:
                ops.flush_wgrads()                               # (no-op outside a deferring block) the 18 products above leave now
        outs.append(ops.gemm(big[0], big[1], M=256, N=512, K=32768, x_kslow=True, w_kslow=True, ldx=256, ldw=512, out_dtype=torch.float32, split_k=0))
        return outs

    keep = ops.group_wgrad_max_k, ops.group_wgrad_small_mn
    try:
        ops.group_wgrad_max_k, ops.group_wgrad_small_mn = 0, 0
        ref = [t.clone() for t in run()]
        ops.group_wgrad_max_k = 20000
        with ops.deferred_reductions():
            got = run()
            assert len(ops._wgroup) == 5 and len(ops._pending) == 18 + 1       # the last five are queued; the K = 32768 product went out alone (its reduction waits)
        torch.cuda.synchronize()
        # run-to-run: the grouped launch is deterministic
        with ops.deferred_reductions():
            again = run()
        torch.cuda.synchronize()
    finally:
        ops.group_wgrad_max_k, ops.group_wgrad_small_mn = keep
    for r, o in zip(ref, got):
        assert o.shape == r.shape
        assert float((o - r).abs().max()) <= 2e-6 * float(r.abs().max()) + 1e-30, (r.shape, float((o - r).abs().max()), float(r.abs().max()))
    for a_, b_ in zip(got, again):
        assert torch.equal(a_, b_)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(8192, 256, 256), (8100, 264, 192), (4096, 1024, 256), (49152, 512, 256)])
def test_gemm_pair_is_bitwise_equal_to_two_calls(M, N, K):
    """dl_gemm_pair: the two streams' products in one launch (128-tile path; the last shape takes the large tile and
    therefore two launches) against two dl_gemm calls, for every epilogue incl. per-problem dropout seeds."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(13)
    dt = torch.bfloat16
    mk = lambda *sh, sc=1.0: [(torch.randn(*sh, generator=g) * sc).to(dt).cuda() for _ in range(2)]      # noqa: E731
    xs, ws = mk(M, K, sc=0.5), mk(N, K, sc=0.1)
    bs = [torch.randn(N, generator=g).cuda() for _ in range(2)]
    res, pre_in = mk(M, N), mk(M, N)
    cases = {
        "plain": dict(), "bias": dict(bias=bs), "relu": dict(bias=bs, act=2),
        "gelu+pre+drop": dict(bias=bs, act=1, pre_out=True, dropout_p=0.1, seed=[5, 9]),
        "dgelu+drop": dict(dact_pre=pre_in, dropout_p=0.1, seed=[5, 9]),
        "res+drop": dict(bias=bs, residual=res, dropout_p=0.1, seed=[7, 11]),
    }
    for name, kw in cases.items():
        outs = {}
        for mode in ("pair", "two"):
            k2 = dict(kw)
            pres = None
            if k2.get("pre_out"):
                pres = [torch.zeros(M, N, device="cuda", dtype=dt) for _ in range(2)]
                k2["pre_out"] = pres
            out = [torch.full((M, N), 7.0, device="cuda", dtype=dt) for _ in range(2)]
            if mode == "pair":
                ops.gemm_pair(xs, ws, M=M, N=N, K=K, out=out, **k2)
            else:
                for i in range(2):
                    ops.gemm(xs[i], ws[i], M=M, N=N, K=K, out=out[i], **{k: (v[i] if isinstance(v, list) else v) for k, v in k2.items()})
            torch.cuda.synchronize()
            outs[mode] = (out, pres)
        for i in range(2):
            assert torch.equal(outs["pair"][0][i], outs["two"][0][i]), (name, i)
            if outs["pair"][1] is not None:
                assert torch.equal(outs["pair"][1][i], outs["two"][1][i]), (name, i)
    assert not torch.equal(outs["pair"][0][0], outs["pair"][0][1])


@pytest.mark.gpu
def test_conv_weight_images_follow_the_masters_through_weight_prep():
    """Round 3: the two GEMM layouts of a Conv1d weight (forward Wg[co][j*ci + c] = w[co][c][j]; data gradient
    Wd[ci][j'*co + o] = w[o][ci][k-1-j']) are weight images refreshed by dl_weight_prep's strided items.  After the master
    changes through a raw pointer (what the fused AdamW does) + bump_param_epoch, the SAME image tensors hold the new
    layouts, bit-equal to torch's permute / flip of the bf16-rounded master."""
    from druglamp_amd import functional as Fn
    g = torch.Generator().manual_seed(3)
    for (co, ci, k) in [(128, 128, 3), (128, 128, 6), (128, 128, 9), (72, 40, 5)]:
        w = torch.nn.Parameter(torch.randn(co, ci, k, generator=g).cuda())
        for dt in (torch.bfloat16, torch.float32):
            f0, b0 = Fn._conv_weight(w, dt, False), Fn._conv_weight(w, dt, True)
            for rep in range(2):
                ref_f = w.detach().permute(0, 2, 1).reshape(co, k * ci).to(dt)
                ref_b = w.detach().flip(2).permute(1, 2, 0).reshape(ci, k * co).to(dt)
                f1, b1 = Fn._conv_weight(w, dt, False), Fn._conv_weight(w, dt, True)
                assert f1.data_ptr() == f0.data_ptr() and b1.data_ptr() == b0.data_ptr()      # fixed addresses (graph replays)
                assert torch.equal(f1, ref_f) and torch.equal(b1, ref_b), (co, ci, k, dt, rep)
                with torch.no_grad():
                    w.data.view(-1).mul_(1.5).add_(0.25)          # a raw update of the master
                Fn.bump_param_epoch()


@pytest.mark.parametrize("P,H,S,Lq,Lk", [(320, 4, 2, 256, 256),      # paired, 640 workgroups: several per CU, AUTO takes the one-pass kernel
                                         (288, 4, 1, 160, 200),      # one segment, ragged lengths
                                         (160, 2, 2, 72, 250)])      # forced (too few workgroups for AUTO), ragged, short q
def test_one_pass_attention_backward_equals_the_kernel_pair(P, H, S, Lq, Lk):
    """dl_attn_bwd at head_dim 64 in bf16 with Lk <= 256: the one-pass kernel (one evaluation of P and dS for dQ, dK, dV; the two
    shares of a paired dQ row summed inside one workgroup) against the dQ + dK/dV kernel pair — same fp32 accumulation, so
    equal to bf16 rounding; bitwise repeatable."""
    from druglamp_amd import ops
    dev, dt, hd = "cuda:0", torch.bfloat16, 64
    d, shift = H * hd, (P // 2 if S == 2 else 0)
    g = torch.Generator().manual_seed(3)
    q = (torch.randn(P * Lq, d, generator=g) * 0.5).to(dev, dt)
    kv = (torch.randn(P * Lk, 2 * d, generator=g) * 0.5).to(dev, dt)
    k, v = kv[:, :d], kv[:, d:]
    qs, ks = (Lq * d, hd, d), (Lk * 2 * d, hd, 2 * d)
    do = (torch.randn(S, P * Lq, d, generator=g) * 0.1).to(dev, dt)
    o = torch.zeros(S, P * Lq, d, device=dev, dtype=dt)
    common = dict(n_problems=P, n_heads=H, n_segments=S, partner_shift=shift, Lq=Lq, Lk=Lk, head_dim=hd, scale=hd ** -0.5,
                  q_strides=qs, k_strides=ks, v_strides=ks)
    lse = ops.attn_fwd(q, k, v, out=o, o_strides=(Lq * d, hd, d), o_ss=P * Lq * d, **common)
    got = {}
    for name, algo in (("pair", 2), ("one", 3), ("one2", 3), ("auto", 0)):
        dq = torch.full((P * Lq, d), float("nan"), device=dev, dtype=dt)
        dk = torch.full((P * Lk, d), float("nan"), device=dev, dtype=dt)
        dv = torch.full((P * Lk, d), float("nan"), device=dev, dtype=dt)
        ops.attn_bwd(q, k, v, o, do, lse, o_strides=(Lq * d, hd, d), o_ss=P * Lq * d, do_strides=(Lq * d, hd, d), do_ss=P * Lq * d,
                     dq=dq, dq_strides=(Lq * d, hd, d), dk=dk, dk_strides=(Lk * d, hd, d), dv=dv, dv_strides=(Lk * d, hd, d),
                     algo=algo, **common)
        got[name] = (dq, dk, dv)
    for i, what in enumerate(("dq", "dk", "dv")):
        ref = got["pair"][i].float()
        assert torch.isfinite(got["one"][i].float()).all(), what
        err = float((got["one"][i].float() - ref).abs().max() / ref.abs().max())
        assert err <= 1.2e-2, (what, err)                                   # a bf16 ulp at the largest magnitude
        assert torch.equal(got["one"][i], got["one2"][i]), what           # repeatable
    nwg = H * (shift if S == 2 else P)
    same = all(torch.equal(got["auto"][i], got["one" if nwg >= 256 else "pair"][i]) for i in range(3))
    assert same, "AUTO did not take the form its rule names"


@pytest.mark.parametrize("dt,hd,H,P,Lq,lead,tail,w", [(torch.bfloat16, 128, 1, 24, 256, 128, 8, 48),     # PGCA's shape: 512 keys -> 136
                                                       (torch.bfloat16, 128, 2, 6, 100, 70, 8, 5),       # ragged lengths, two heads
                                                       (torch.bfloat16, 64, 4, 6, 256, 128, 8, 48),      # head_dim 64 takes the streaming forms
                                                       (torch.float32, 128, 1, 4, 96, 61, 3, 7),         # fp32 pipeline (delta launch, generic kernels)
                                                       (torch.float32, 64, 2, 3, 80, 128, 8, 16)])
def test_attention_with_key_multiplicities_equals_the_attention_over_the_expanded_keys(dt, hd, H, P, Lq, lead, tail, w):
    """dl_attn_fwd / dl_attn_bwd with key_tail_rows (round 5): the last `tail` keys each stand for `w` identical keys.  Against the
    attention over the EXPANDED key set (lead + w * tail rows, tail row j repeated as rows lead + j + m * tail) in fp64: same
    output, LSE and dQ; dK / dV of the lead keys equal, of a tail key the SUM over its copies.  And against the library's own
    plain attention on the expanded keys (the path the compact form replaces)."""
    from druglamp_amd import ops
    dev = "cuda:0"
    d, Lc, Lf = H * hd, lead + tail, lead + w * tail
    g = torch.Generator().manual_seed(7)
    q = (torch.randn(P * Lq, d, generator=g) * 0.7).to(dev, dt)
    kvc = (torch.randn(P, Lc, 2 * d, generator=g) * 0.7).to(dev, dt)
    kvf = torch.cat([kvc[:, :lead], kvc[:, lead:].unsqueeze(1).expand(P, w, tail, 2 * d).reshape(P, w * tail, 2 * d)], 1).contiguous()
    do = (torch.randn(P * Lq, d, generator=g) * 0.2).to(dev, dt)
    scale = hd ** -0.5

    def run(kv, Lk, key_tail):
        kv2 = kv.reshape(P * Lk, 2 * d)
        k, v = kv2[:, :d], kv2[:, d:]
        common = dict(n_problems=P, n_heads=H, n_segments=1, partner_shift=0, Lq=Lq, Lk=Lk, head_dim=hd, scale=scale,
                      q_strides=(Lq * d, hd, d), k_strides=(Lk * 2 * d, hd, 2 * d), v_strides=(Lk * 2 * d, hd, 2 * d), key_tail=key_tail)
        o = torch.full((P * Lq, d), float("nan"), device=dev, dtype=dt)
        lse = ops.attn_fwd(q, k, v, out=o, o_strides=(Lq * d, hd, d), o_ss=0, **common)
        dq = torch.full((P * Lq, d), float("nan"), device=dev, dtype=dt)
        dkv = torch.full((P * Lk, 2 * d), float("nan"), device=dev, dtype=dt)
        ops.attn_bwd(q, k, v, o, do, lse, o_strides=(Lq * d, hd, d), o_ss=0, do_strides=(Lq * d, hd, d), do_ss=0, dq=dq,
                     dq_strides=(Lq * d, hd, d), dk=dkv, dk_strides=(Lk * 2 * d, hd, 2 * d), dv=dkv[:, d:], dv_strides=(Lk * 2 * d, hd, 2 * d),
                     **common)
        return o.double(), lse.double(), dq.double(), dkv.double().reshape(P, Lk, 2 * d)

    oc, lc, dqc, dkvc = run(kvc, Lc, (tail, float(w)))
    of, lf, dqf, dkvf = run(kvf, Lf, None)
    # fp64 truth on the expanded keys
    qd = q.double().reshape(P, Lq, H, hd).permute(0, 2, 1, 3).requires_grad_(True)
    kd = kvf.double()[..., :d].reshape(P, Lf, H, hd).permute(0, 2, 1, 3).requires_grad_(True)
    vd = kvf.double()[..., d:].reshape(P, Lf, H, hd).permute(0, 2, 1, 3).requires_grad_(True)
    sc = (qd @ kd.transpose(-1, -2)) * scale
    ot = (torch.softmax(sc, -1) @ vd).permute(0, 2, 1, 3).reshape(P * Lq, d)
    ot.backward(do.double())
    lt = torch.logsumexp(sc, -1).detach()                                        # (P, H, Lq)

    def fold(t):                                                                  # (P, H, Lf, hd) gradient -> compact rows (sum over the copies)
        t = t.permute(0, 2, 1, 3).reshape(P, Lf, d)
        return torch.cat([t[:, :lead], t[:, lead:].reshape(P, w, tail, d).sum(1)], 1)
    dk_t, dv_t = fold(kd.grad), fold(vd.grad)
    dq_t = qd.grad.permute(0, 2, 1, 3).reshape(P * Lq, d)
    tol = 2e-2 if dt == torch.bfloat16 else 2e-5
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))       # noqa: E731
    assert rel(oc, ot.detach()) <= tol and rel(lc.reshape(P, H, Lq), lt) <= (2e-3 if dt == torch.bfloat16 else 1e-5)
    assert rel(dqc, dq_t) <= tol
    assert rel(dkvc[..., :d], dk_t) <= tol and rel(dkvc[..., d:], dv_t) <= tol
    # ... and the plain attention over the expanded keys agrees with the compact one to the pipeline's rounding
    assert rel(oc, of) <= tol and rel(dqc, dqf) <= tol
    assert rel(dkvc[:, :lead], dkvf[:, :lead]) <= tol


@pytest.mark.parametrize("dt,N,C,ld", [(torch.float32, 5000, 27, 32), (torch.bfloat16, 70001, 27, 32), (torch.float32, 300, 5, 5), (torch.bfloat16, 257, 40, 48)])
def test_cross_entropy_rows_against_torch(dt, N, C, ld):
    """dl_ce_rows_fwd / bwd (round 5: the masked-LM heads' F.cross_entropy(logits, labels, ignore_index=0)) against torch in
    fp64 on the same (rounded) logits: loss, gradient incl. the zero padding columns and ignored rows; all-ignored labels."""
    from druglamp_amd import functional as Fn
    g = torch.Generator().manual_seed(12)
    full = (torch.randn(N, ld, generator=g) * 3).to("cuda:0", dt)
    labels = torch.randint(0, C, (N,), generator=g).cuda()
    labels[torch.rand(N, generator=g).cuda() < 0.6] = 0                               # ignore_index = 0, as in the MLM head
    x = full.clone().requires_grad_(True)
    loss = Fn.CrossEntropyRowsFn.apply(x, labels, C, 0)
    (loss * 0.37).backward()
    xr = full[:, :C].double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, labels, ignore_index=0)
    (ref * 0.37).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    gtol = 1e-2 if dt == torch.bfloat16 else 2e-6
    assert float((x.grad[:, :C].double() - xr.grad).abs().max()) <= gtol * float(xr.grad.abs().max())
    assert torch.count_nonzero(x.grad[:, C:]) == 0 and torch.count_nonzero(x.grad[labels == 0]) == 0
    assert float(Fn.CrossEntropyRowsFn.apply(full, labels, C, 0)) == float(loss)      # fixed summation order
    # as torch: every label ignored -> NaN (0 / 0); a label outside [0, C) that is not ignore_index (torch: device
    # assert) poisons the loss instead of being skipped
    none = Fn.CrossEntropyRowsFn.apply(full.clone().requires_grad_(True), torch.zeros_like(labels), C, 0)
    assert math.isnan(float(none)) and math.isnan(float(torch.nn.functional.cross_entropy(xr.detach(), torch.zeros_like(labels), ignore_index=0)))
    bad = labels.clone()
    bad[N // 2] = C
    assert math.isnan(float(Fn.CrossEntropyRowsFn.apply(full, bad, C, 0)))


@pytest.mark.parametrize("dt,R,C", [(torch.float32, 3000, 512), (torch.bfloat16, 20000, 512), (torch.bfloat16, 777, 128), (torch.float32, 65, 36)])
def test_batch_norm_relu_rows_against_torch(dt, R, C):
    """BatchNormRowsFn(relu=True) (round 5: BatchNorm1d -> ReLU in one kernel each way) against torch's batch_norm + relu in fp64:
    output, input / gamma / beta gradients, running statistics."""
    from druglamp_amd import functional as Fn
    g = torch.Generator().manual_seed(13)
    x0 = (torch.randn(R, C, generator=g) * 1.3 + 0.2).to("cuda:0", dt)
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    dz = torch.randn(R, C, generator=g).to("cuda:0", dt)
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(gam); bn.bias.copy_(bet)
    x = x0.clone().requires_grad_(True)
    z = Fn.batch_norm_rows(bn, x, relu=True)
    z.backward(dz)
    xr = x0.double().requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64, device="cuda:0"), torch.ones(C, dtype=torch.float64, device="cuda:0")
    zr = torch.relu(torch.nn.functional.batch_norm(xr, rm, rv, gr, br, True, 0.1, 1e-5))
    zr.backward(dz.double())
    tol = 2e-2 if dt == torch.bfloat16 else 2e-5
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))     # noqa: E731
    assert rel(z, zr.detach()) <= tol
    # (an element whose normalised value sits within rounding of the ReLU kink may open on one side only: compare in aggregate)
    assert float((x.grad.double() - xr.grad).norm() / xr.grad.norm()) <= tol
    assert rel(bn.weight.grad, gr.grad) <= tol and rel(bn.bias.grad, br.grad) <= tol
    assert rel(bn.running_mean, rm) <= max(tol, 1e-5) and rel(bn.running_var, rv) <= max(tol, 1e-5)
    assert bool((z >= 0).all())


@pytest.mark.parametrize("dt,M,dd", [(torch.bfloat16, 65536, 1024), (torch.float32, 1000, 1024), (torch.bfloat16, 777, 64), (torch.float32, 33, 2048)])
def test_gate_dpre_against_fp64(dt, M, dd):
    """dl_gate_dpre (round 5: MHLA's lin2 data gradient + the GELU derivative as one elementwise pass) against
    gelu'(pre) * (dl @ w2) in fp64; bf16 uses the pipeline's rational gelu' (1e-4 absolute)."""
    from druglamp_amd import ops
    g = torch.Generator().manual_seed(5)
    dl = torch.randn(M, 8, generator=g).to("cuda:0", dt)
    w2 = (torch.randn(8, dd, generator=g) * 0.2).to("cuda:0", dt)
    pre = (torch.randn(M, dd, generator=g) * 1.5).to("cuda:0", dt)
    out = ops.gate_dpre(dl, w2, pre)
    x = pre.double()
    gp = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    ref = gp * (dl.double() @ w2.double())
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    assert err <= (1.2e-2 if dt == torch.bfloat16 else 2e-6), err
    assert torch.equal(out, ops.gate_dpre(dl, w2, pre))


@pytest.mark.parametrize("dt,R,C,win", [(torch.bfloat16, 5000, 128, (0, 0, 0)), (torch.bfloat16, 4 * 2312, 128, (2312, 4, 2304)),
                                       (torch.float32, 777, 72, (0, 0, 0)), (torch.bfloat16, 3000, 96, (0, 0, 0)),
                                       (torch.bfloat16, 9000, 256, (0, 0, 0))])
@pytest.mark.parametrize("weighted", [False, True])
def test_bn_stats_finalize_is_bit_identical_to_the_three_launch_form(dt, R, C, win, weighted):
    """dl_bn_stats_finalize (partial sums, then ONE kernel for the second reduction stage + mean / var / rstd + the running
    statistics) against dl_bn_stats (+ _rw) followed by dl_bn_finalize: every output bitwise equal."""
    from druglamp_amd import ops
    if weighted and win != (0, 0, 0):
        pytest.skip("row weights replace the window rule")
    dev = "cuda:0"
    g = torch.Generator().manual_seed(11)
    y = torch.randn(R, C, generator=g).to(dev, dt)
    rw = None
    n = R if win == (0, 0, 0) else (R // win[0]) * win[2]
    if weighted:
        rw = torch.randint(-1, 4, (R,), generator=g).float().to(dev)
        n = int(rw.clamp(min=0).sum().item())
    rm0, rv0 = torch.randn(C, generator=g).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)
    rm_a, rv_a, rm_b, rv_b = rm0.clone(), rv0.clone(), rm0.clone(), rv0.clone()
    sums = ops.bn_stats(y, *win, rw)
    ma, va, ra = ops.bn_finalize(sums, n, 1e-5, 0.1, rm_a, rv_a)
    mb, vb, rb = ops.bn_stats_finalize(y, *win, n, 1e-5, 0.1, rm_b, rv_b, rw)
    for a, b, what in ((ma, mb, "mean"), (va, vb, "var"), (ra, rb, "rstd"), (rm_a, rm_b, "running mean"), (rv_a, rv_b, "running var")):
        assert torch.equal(a, b), what
    assert not torch.equal(rm_a, rm0)


@pytest.mark.gpu
def test_gelu_epilogue_of_the_bf16_pipeline_is_within_half_a_bf16_ulp_of_erf_gelu():
    """Round 6: the bf16 GEMM epilogues evaluate GELU as x (1/2 + x R(x^2)) with a degree-8 minimax R on |x| <= 4.3 (13 VALU slots per
    pair instead of the rational erf's 28; common.cuh::gelu_fast2).  Here the pre-activations are EXACT: x rows are one-hot, so
    acc = one weight value, bias 0 — a grid over [-12, 12] plus the region around 0 and the minimum of GELU.  Output against
    bf16(erf-GELU(pre)) in fp64: the error may not exceed half a bf16 ulp of the exact value + 8e-5 (the stated bound of the
    approximation); the pre-activation copy is the exact value."""
    from druglamp_amd import ops
    dt = torch.bfloat16
    M, N, K = 4096, 256, 64
    grid = torch.cat((torch.linspace(-12, 12, 8192), torch.linspace(-1.5, 1.5, 4096), torch.linspace(-4.5, -3.5, 2048),
                      torch.linspace(3.5, 4.5, 2048)))[:N * K].to(dt)
    w = grid.reshape(N, K).contiguous().cuda()                      # w[n][k]
    x = torch.zeros(M, K, dtype=dt)
    x[torch.arange(M), torch.arange(M) % K] = 1.0                   # row m selects column k = m % 64
    x = x.cuda()
    for algo in (1, 0):
        pre = torch.empty(M, N, device="cuda", dtype=dt)
        h = ops.gemm(x, w, M=M, N=N, K=K, act=1, pre_out=pre, bias=torch.zeros(N, device="cuda"), algo=algo)
        want_pre = w.t()[torch.arange(M, device="cuda") % K]          # (M, N): pre[m][n] = w[n][m % K]
        assert torch.equal(pre, want_pre)
        p64 = pre.double()
        ref = 0.5 * p64 * (1.0 + torch.erf(p64 / 2 ** 0.5))
        err = (h.double() - ref).abs()
        bound = 0.5 * ref.abs() * 2.0 ** -7 + 8e-5                    # half a bf16 ulp (<= 2^-8 relative to the binade's top: 2^-7 x value / 2) + the bound
        assert bool((err <= bound).all()), (algo, float((err - bound).max()), float(p64.flatten()[(err - bound).argmax()]))
