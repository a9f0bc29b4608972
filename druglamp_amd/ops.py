"""Tensor-level wrappers over the C ABI (druglamp_amd/_lib.py) and the autograd Functions built on
them.  torch is used for device memory, streams and autograd plumbing only: every arithmetic op on
the hot path below is a libdruglamp_hip entry point.  There is no CPU path here — tensors must live
on a HIP device and the shared library must be present.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
from typing import Optional

import torch

from . import _lib
from ._lib import DL_BF16, DL_F32, AttnBwdArgs, AttnFwdArgs, GemmArgs, check

_DT = {torch.float32: DL_F32, torch.bfloat16: DL_BF16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("druglamp_amd: unsupported dtype %s (float32 / bfloat16 only)" % t.dtype)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current torch stream's hipStream_t.  torch.cuda.current_stream() builds a Stream object through three
    Python layers (9 us, ~260 times per step); the raw getter is the same handle in well under a microsecond."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("druglamp_amd: the HIP hot path needs device tensors (got a %s tensor); "
                               "there is no CPU fallback" % t.device)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class _Workspace:
    """One growing scratch buffer per device; all launches that use it are stream-ordered."""

    def __init__(self):
        self.bufs = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # a captured launch must not point into the shared buffer: that buffer is REPLACED (and the old one freed) the
            # first time an eager call needs more room, and a later replay would write through the stale address.  Scratch
            # taken during capture comes from the graph's own pool and stays reserved for as long as the graph lives.
            return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
        # one buffer per (device, stream): launches that share it are stream-ordered; branches of the forward that run on
        # side streams (DL_BRANCH_STREAMS) get their own
        key = (device.type, device.index, _stream() if device.type == "cuda" else 0)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


_ws = _Workspace()
_ws2 = _Workspace()  # second buffer so that two scratch users can be live inside one op


# ------------------------------------------------------------------------------------------------
# raw ops
# ------------------------------------------------------------------------------------------------
# Sub-family tag of the dl_gemm launches issued inside a `with prof_tag(T):` block (dl_gemm_args.prof_tag; timing hooks
# only).  Weight-gradient products outside the ProteinCNN are tagged TAG_WGRAD whatever the block says.
_prof_tag = 0


class prof_tag:
    __slots__ = ("tag", "prev")

    def __init__(self, tag: int):
        self.tag = tag

    def __enter__(self):
        global _prof_tag
        self.prev = _prof_tag
        _prof_tag = self.tag
        return self

    def __exit__(self, *exc):
        global _prof_tag
        _prof_tag = self.prev
        return False


def _gemm_args(x: torch.Tensor, w: torch.Tensor, *, M: int, N: int, K: int, x_kslow=False, w_kslow=False,
               ldx: Optional[int] = None, ldw: Optional[int] = None, bias=None, residual=None, res_row_mod=0,
               res_before_dropout=False, act=0, pre_out=None, dact_pre=None, dropout_p=0.0, seed=0, out=None,
               out_dtype=None, accumulate=False, split_k=-1, x_colsum=None, algo=0):
    """The dl_gemm_args block of one product (and its output tensor)."""
    _need_gpu(x, w)
    out_dtype = out_dtype or x.dtype
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=x.device)
    a = GemmArgs()
    a.X, a.W, a.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    a.ldx = ldx if ldx is not None else (M if x_kslow else K)
    a.ldw = ldw if ldw is not None else (N if w_kslow else K)
    a.ldc = out.stride(0) if out.dim() == 2 else N
    a.x_kslow, a.w_kslow = int(x_kslow), int(w_kslow)
    a.M, a.N, a.K = M, N, K
    a.in_dtype, a.out_dtype = _dt(x), _dt(out)
    a.bias = _ptr(bias)
    a.residual = _ptr(residual)
    a.ldr = N if residual is None else residual.stride(-2)
    a.res_row_mod = res_row_mod
    a.res_before_dropout = int(res_before_dropout)
    a.act = act
    a.pre_out = _ptr(pre_out)
    a.ldp = N
    a.dact_pre = _ptr(dact_pre)
    a.lddp = N
    a.dropout_p = float(dropout_p)
    a.dropout_seed = int(seed)
    a.accumulate = int(accumulate)
    a.split_k = split_k
    a.x_colsum = _ptr(x_colsum)
    a.dropout_seed_offset = _seed_offset_ptr(x.device) if dropout_p > 0 else None
    a.algo = algo
    a.tile_tickets = _tickets_ptr(x.device)
    a.prof_tag = _lib.TAG_WGRAD if (x_kslow and w_kslow and _prof_tag != _lib.TAG_CONV) else _prof_tag
    return a, out


def gemm(x: torch.Tensor, w: torch.Tensor, **kw) -> torch.Tensor:
    """C[M,N] = epilogue(sum_k X[m,k] W[n,k]); see include/druglamp_hip.h (dl_gemm).  Keywords: _gemm_args."""
    a, out = _gemm_args(x, w, **kw)
    accumulate, x_colsum = kw.get("accumulate", False), kw.get("x_colsum")
    if (_wgroup is not None and a.x_kslow and a.w_kslow and a.split_k == 0 and not accumulate
            and (a.K <= group_wgrad_max_k or a.M * a.N < group_wgrad_small_mn)
            and x.dtype == torch.bfloat16 and out.dtype == torch.float32 and a.bias is None and a.residual is None and not a.act):
        _wgroup.append((a, out, x_colsum, x, w))         # leaves with the block's other weight gradients (flush_wgrads)
        return out
    _gemm_launch(a, out, x, accumulate, x_colsum)
    return out


def _gemm_launch(a, out, x, accumulate, x_colsum) -> None:
    L = _lib.lib()
    if accumulate and (_pending or _wgroup):
        flush_reductions()                  # an accumulating product must see every earlier (queued) write to its output
    nbytes = L.dl_gemm_workspace_bytes(C.byref(a))
    item = None
    if nbytes:
        if _pending is not None and not accumulate:
            # inside deferred_reductions(): the slabs get a buffer of their own (it has to outlive the batched launch) and
            # the reduction is queued instead of launched
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            item = _lib.ReduceItem()
            a.deferred = C.pointer(item)
        else:
            ws = _ws.get(nbytes, x.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    check(L.dl_gemm(C.byref(a), _stream()), "dl_gemm")
    if item is not None and item.kind != 0:
        _pending.append((item, ws, out, x_colsum))


# Grouped weight gradients (dl_gemm_group, round 3).  Inside deferred_reductions() — a block's backward — the weight-gradient
# products (bf16, both operands K-slow, f32 output, at most group_wgrad_max_k rows) are not launched one by one: they queue
# up and leave as ONE launch when the block ends (16 per launch), followed by the block's one batched reduction.  At 32-64
# pairs per GPU each of them alone is a few tiles over 8192-16384 rows behind a 16- to 32-way split; together they fill the
# chip with 1-4 slabs each.  The rule of deferred_reductions() covers it: nothing inside the block reads these outputs,
# and their operands (saved activations, fresh gradient tensors) are not written again before the block ends.
# Measured (bench.py, same box, off -> on): batch 32 4.71 -> 4.23 ms, 64 6.22 -> 5.75, 128 9.32 -> 8.77, 256 15.53 -> 15.0.
group_wgrad_max_k = int(os.environ.get("DL_GROUP_WGRAD_MAX_K", "65536"))         # 0 (and small_mn 0): off
group_wgrad_small_mn = int(os.environ.get("DL_GROUP_WGRAD_SMALL_MN", "655360"))  # products with fewer outputs are grouped at any K
group_wgrad_min_n = 2          # a single queued product takes its usual dl_gemm path
_wgroup = None


def flush_wgrads() -> None:
    if not _wgroup:
        return
    L = _lib.lib()
    todo = list(_wgroup)
    del _wgroup[:]
    for i in range(0, len(todo), _lib.GEMM_GROUP_MAX):
        part = todo[i:i + _lib.GEMM_GROUP_MAX]
        n = len(part)
        arr = (GemmArgs * n)(*[t[0] for t in part])
        splits = (C.c_int32 * n)()
        if n < group_wgrad_min_n or L.dl_gemm_group_plan(arr, n, splits) != 0:
            for a, out, x_colsum, x, _w in part:             # not a group the library takes: one by one, as before
                _gemm_launch(a, out, x, False, x_colsum)
            continue
        keep = []
        for k, (a, out, x_colsum, x, _w) in enumerate(part):
            nbytes = int(splits[k]) * a.M * (a.N + (1 if x_colsum is not None else 0)) * 4
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            item = _lib.ReduceItem()
            arr[k].workspace, arr[k].workspace_bytes = ws.data_ptr(), nbytes
            arr[k].deferred = C.pointer(item)
            keep.append((item, ws, out, x_colsum))
        check(L.dl_gemm_group(arr, n, _stream()), "dl_gemm_group")
        _pending.extend(keep)


def gemm_pair(xs, ws, **kw):
    """Two products of one shape / layout / epilogue — the two streams of a paired block — through dl_gemm_pair: ONE launch
    when both are on the 128-tile path without split-K, two otherwise; bit-identical to two gemm() calls.  xs, ws: the two
    operand pairs; every other keyword of gemm() is either one value for both or a 2-tuple / 2-list (one per problem)."""
    L = _lib.lib()
    per = [{k: (v[i] if isinstance(v, (tuple, list)) else v) for k, v in kw.items()} for i in range(2)]
    (a0, o0), (a1, o1) = _gemm_args(xs[0], ws[0], **per[0]), _gemm_args(xs[1], ws[1], **per[1])
    if L.dl_gemm_workspace_bytes(C.byref(a0)) or L.dl_gemm_workspace_bytes(C.byref(a1)):
        return gemm(xs[0], ws[0], **per[0]), gemm(xs[1], ws[1], **per[1])      # split-K shapes: the plain path (own workspaces)
    check(L.dl_gemm_pair(C.byref(a0), C.byref(a1), _stream()), "dl_gemm_pair")
    return o0, o1


# Deferred second-stage reductions (dl_reduce_item / dl_reduce_batch): inside `with deferred_reductions():` every split-K
# GEMM (and LayerNorm backward) only writes its partial results; the reductions leave together, DL_REDUCE_BATCH_MAX per
# launch, when the block ends.  NOTHING inside the block may read an output of those calls — the backward passes that use
# it only hand the gradients back to autograd.
_pending = None


@contextlib.contextmanager
def deferred_reductions():
    global _pending, _wgroup
    if _pending is not None:            # nested: the outermost block flushes
        yield
        return
    _pending = []
    _wgroup = [] if (group_wgrad_max_k > 0 or group_wgrad_small_mn > 0) else None
    try:
        yield
    finally:
        try:
            flush_reductions()
        finally:
            _pending = None
            _wgroup = None


def flush_reductions() -> None:
    """Launch what is queued (keeps the deferred mode, if any, active): the grouped weight gradients first — their slabs
    are part of the batched reduction."""
    flush_wgrads()
    if not _pending:
        return
    L = _lib.lib()
    todo = list(_pending)
    del _pending[:]
    for i in range(0, len(todo), _lib.REDUCE_BATCH_MAX):
        part = todo[i:i + _lib.REDUCE_BATCH_MAX]
        arr = (_lib.ReduceItem * len(part))(*[t[0] for t in part])
        check(L.dl_reduce_batch(arr, len(part), _stream()), "dl_reduce_batch")


# Dynamic tile hand-out of the persistent large-tile GEMM (dl_gemm_args.tile_tickets): on while gradient buckets are being
# all-reduced DURING backward (trainer.GradOverlap), when RCCL channel workgroups occupy some CUs for a while — a CU that
# starts late then simply takes fewer tiles.  Two int32 per device, zero between launches (the kernel restores them).
_tickets = {}
_tickets_on = False


def dynamic_tiles(on: bool) -> None:
    global _tickets_on
    _tickets_on = bool(on)


def reset_tickets() -> None:
    """Zero the ticket words (Trainer construction; after a failed launch they could hold a stale count and later
    ticketed GEMMs would skip tiles — ADVICE round 2)."""
    for t in _tickets.values():
        t.zero_()


def _tickets_ptr(device):
    if not _tickets_on:
        return None
    key = (device.type, device.index)
    t = _tickets.get(key)
    if t is None:
        t = _tickets[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return t.data_ptr()


def colsum(x2d: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate=False) -> torch.Tensor:
    _need_gpu(x2d)
    L = _lib.lib()
    M, N = x2d.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x2d.device)
    ws = _ws2.get(L.dl_colsum_workspace_bytes(M, N), x2d.device)
    check(L.dl_colsum(x2d.data_ptr(), x2d.stride(0), M, N, _dt(x2d), out.data_ptr(), int(accumulate),
                      ws.data_ptr(), ws.numel(), _stream()), "dl_colsum")
    return out


def layernorm_fwd(x2d, gamma, beta, eps):
    _need_gpu(x2d, gamma, beta)
    L = _lib.lib()
    M, D = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(M, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x2d.device)
    check(L.dl_layernorm_fwd(x2d.data_ptr(), x2d.stride(0), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                             y.stride(0), mean.data_ptr(), rstd.data_ptr(), M, D, float(eps), _dt(x2d), _stream()),
          "dl_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy2d, x2d, mean, rstd, gamma, dres=None, need_param_grads=True, out=None, dy_share=1):
    """dy_share > 1: dy2d has M / dy_share rows, each shared by dy_share consecutive rows of x2d."""
    L = _lib.lib()
    M, D = x2d.shape
    if dy2d.shape[0] * dy_share != M:
        raise ValueError("layernorm_bwd: dy has %d rows, x %d, dy_share %d" % (dy2d.shape[0], M, dy_share))
    dx = torch.empty_like(x2d) if out is None else out
    dgamma = dbeta = None
    if need_param_grads:
        gb = torch.empty((2, D), dtype=torch.float32, device=x2d.device)   # adjacent: one final-reduction launch
        dgamma, dbeta = gb[0], gb[1]
    nbytes = L.dl_layernorm_bwd_workspace_bytes(M, D)
    item = None
    if _pending is not None and need_param_grads:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x2d.device)     # the partials outlive this call (deferred_reductions)
        item = _lib.ReduceItem()
    else:
        ws = _ws.get(nbytes, x2d.device)
    check(L.dl_layernorm_bwd(dy2d.data_ptr(), dy2d.stride(0), int(dy_share), x2d.data_ptr(), x2d.stride(0), mean.data_ptr(),
                             rstd.data_ptr(), gamma.data_ptr(), _ptr(dres), 0 if dres is None else dres.stride(0),
                             dx.data_ptr(), dx.stride(0), _ptr(dgamma), _ptr(dbeta), 0, M, D, _dt(x2d),
                             ws.data_ptr(), ws.numel(), None if item is None else C.pointer(item), _stream()), "dl_layernorm_bwd")
    if item is not None and item.kind != 0:
        _pending.append((item, ws, gb, None))
    return dx, dgamma, dbeta


def cast(src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if src.dtype == dtype:
        return src
    _need_gpu(src)
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(_lib.lib().dl_cast(src.data_ptr(), _dt(src), dst.data_ptr(), _DT[dtype], src.numel(), _stream()), "dl_cast")
    return dst


def attn_fwd(q, k, v, *, n_problems, n_heads, n_segments, partner_shift, Lq, Lk, head_dim, scale,
             q_strides, k_strides, v_strides, out, o_strides, o_ss, need_lse=True, raw_logits=None, algo=0, key_tail=None):
    """Strides are (problem, head, row) in elements.  Returns the LSE tensor (or None).
    key_tail = (rows, weight): the last `rows` keys each stand for `weight` identical keys (dl_attn_fwd_args.key_tail_rows)."""
    _need_gpu(q, k, v, out)
    a = AttnFwdArgs()
    a.Q, a.K, a.V, a.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    lse = None
    if need_lse:
        lse = torch.empty((n_segments, n_problems, n_heads, Lq), dtype=torch.float32, device=q.device)
    a.LSE = _ptr(lse)
    a.raw_logits = _ptr(raw_logits)
    a.q_ps, a.q_hs, a.q_rs = q_strides
    a.k_ps, a.k_hs, a.k_rs = k_strides
    a.v_ps, a.v_hs, a.v_rs = v_strides
    a.o_ps, a.o_hs, a.o_rs = o_strides
    a.o_ss = o_ss
    a.n_problems, a.n_heads, a.n_segments, a.partner_shift = n_problems, n_heads, n_segments, partner_shift
    a.Lq, a.Lk, a.head_dim, a.dtype = Lq, Lk, head_dim, _dt(q)
    a.scale = float(scale)
    a.algo = algo
    if key_tail is not None:
        a.key_tail_rows, a.key_tail_weight = int(key_tail[0]), float(key_tail[1])
    check(_lib.lib().dl_attn_fwd(C.byref(a), _stream()), "dl_attn_fwd")
    return lse


_ATTN_BWD_ALGO = int(os.environ.get("DL_ATTN_BWD_ALGO", "0"))    # A/B switch: 2 = always the dQ + dK/dV kernel pair, 3 = one-pass wherever eligible


def attn_bwd(q, k, v, o, do, lse, *, n_problems, n_heads, n_segments, partner_shift, Lq, Lk, head_dim, scale,
             q_strides, k_strides, v_strides, o_strides, o_ss, do_strides, do_ss, dq, dq_strides, dk, dk_strides,
             dv, dv_strides, algo=0, key_tail=None):
    _need_gpu(q, k, v, o, do)
    a = AttnBwdArgs()
    delta = torch.empty_like(lse)
    a.Q, a.K, a.V, a.O, a.dO = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), do.data_ptr()
    a.LSE, a.Delta = lse.data_ptr(), delta.data_ptr()
    a.dQ, a.dK, a.dV = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    a.q_ps, a.q_hs, a.q_rs = q_strides
    a.k_ps, a.k_hs, a.k_rs = k_strides
    a.v_ps, a.v_hs, a.v_rs = v_strides
    a.o_ps, a.o_hs, a.o_rs = o_strides
    a.o_ss = o_ss
    a.do_ps, a.do_hs, a.do_rs = do_strides
    a.do_ss = do_ss
    a.dq_ps, a.dq_hs, a.dq_rs = dq_strides
    a.dk_ps, a.dk_hs, a.dk_rs = dk_strides
    a.dv_ps, a.dv_hs, a.dv_rs = dv_strides
    a.n_problems, a.n_heads, a.n_segments, a.partner_shift = n_problems, n_heads, n_segments, partner_shift
    a.Lq, a.Lk, a.head_dim, a.dtype = Lq, Lk, head_dim, _dt(q)
    a.scale = float(scale)
    a.algo = algo if algo else _ATTN_BWD_ALGO
    if key_tail is not None:
        a.key_tail_rows, a.key_tail_weight = int(key_tail[0]), float(key_tail[1])
    check(_lib.lib().dl_attn_bwd(C.byref(a), _stream()), "dl_attn_bwd")


def dropout_apply(x2d: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    y = torch.empty_like(x2d)
    rows, D = x2d.shape
    check(_lib.lib().dl_dropout_apply(x2d.data_ptr(), y.data_ptr(), rows, D, x2d.stride(0), y.stride(0), float(p),
                                      int(seed), _seed_offset_ptr(x2d.device), _dt(x2d), _stream()), "dl_dropout_apply")
    return y


# ------------------------------------------------------------------------------------------------
# dropout seeds: one generator per process; each dropout site draws a fresh 63-bit seed per forward
# ------------------------------------------------------------------------------------------------
_seed_gen = torch.Generator(device="cpu")
_seed_gen.manual_seed(0x5EED)


def manual_seed(seed: int) -> None:
    _seed_gen.manual_seed(int(seed))


# Device-resident seed offset (one uint64 per device): every dropout site keys its mask by (site seed + offset), the
# kernels read the offset when they RUN.  Eager steps leave it at 0; a hipGraph-captured step (trainer.GraphedStep)
# bumps it with an in-graph add so that each replay of the frozen launch arguments still draws fresh masks.
_seed_offsets = {}
_seed_offset_on = False


def seed_offset_tensor(device) -> torch.Tensor:
    key = (device.type, device.index)
    t = _seed_offsets.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=device)
        _seed_offsets[key] = t
    return t


def use_seed_offset(on: bool) -> None:
    """Route dropout sites through the device-resident offset (needed while capturing / replaying a graphed step)."""
    global _seed_offset_on
    _seed_offset_on = bool(on)


def _seed_offset_ptr(device):
    return seed_offset_tensor(device).data_ptr() if _seed_offset_on else None


def next_seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,), generator=_seed_gen).item())


def rowmod_sum(x2d: torch.Tensor, L: int) -> torch.Tensor:
    M, D = x2d.shape
    out = torch.empty((L, D), dtype=torch.float32, device=x2d.device)
    check(_lib.lib().dl_rowmod_sum(x2d.data_ptr(), out.data_ptr(), M, D, L, 0, _dt(x2d), _stream()), "dl_rowmod_sum")
    return out


def add_rowmod_dropout(x2d: torch.Tensor, pe2d: Optional[torch.Tensor], p: float, seed: int) -> torch.Tensor:
    M, D = x2d.shape
    y = torch.empty_like(x2d)
    Lr = pe2d.shape[0] if pe2d is not None else 1
    check(_lib.lib().dl_add_rowmod_dropout(x2d.data_ptr(), _ptr(pe2d), y.data_ptr(), M, D, Lr, float(p), int(seed),
                                           _seed_offset_ptr(x2d.device) if p > 0 else None, _dt(x2d), _stream()),
          "dl_add_rowmod_dropout")
    return y


def token_gate_fwd(v: torch.Tensor, logits: torch.Tensor, H: int, add_residual: bool):
    B, L, D = v.shape
    out = torch.empty_like(v)
    gate = torch.empty((B, H, L), dtype=torch.float32, device=v.device)
    check(_lib.lib().dl_token_gate_fwd(v.data_ptr(), logits.data_ptr(), out.data_ptr(), gate.data_ptr(), B, L, D, H,
                                       int(add_residual), _dt(v), _stream()), "dl_token_gate_fwd")
    return out, gate


def token_gate_bwd(dout, v, gate, H: int, add_residual: bool):
    B, L, D = v.shape
    dv = torch.empty_like(v)
    dlogits = torch.empty((B, L, H), dtype=v.dtype, device=v.device)
    check(_lib.lib().dl_token_gate_bwd(dout.data_ptr(), v.data_ptr(), gate.data_ptr(), dv.data_ptr(),
                                       dlogits.data_ptr(), B, L, D, H, int(add_residual), _dt(v), _stream()),
          "dl_token_gate_bwd")
    return dv, dlogits


def adamw_step(param, grad, exp_avg, exp_avg_sq, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, step,
               grad_scale=1.0, lowp=None):
    n = param.numel()
    check(_lib.lib().dl_adamw_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n,
                                   float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                   float(grad_scale), _ptr(lowp), _DT[lowp.dtype] if lowp is not None else 0,
                                   _stream()), "dl_adamw_step")


def cos_rowloss_fwd(x, y):
    n, D = x.shape
    row = torch.empty(n, dtype=torch.float32, device=x.device)
    check(_lib.lib().dl_cos_rowloss_fwd(x.data_ptr(), y.data_ptr(), row.data_ptr(), None, n, D, _stream()),
          "dl_cos_rowloss_fwd")
    return row


def cos_rowloss_bwd(x, y, grad_scale: float):
    n, D = x.shape
    dx = torch.empty_like(x)
    check(_lib.lib().dl_cos_rowloss_bwd(x.data_ptr(), y.data_ptr(), float(grad_scale), dx.data_ptr(), n, D, _stream()),
          "dl_cos_rowloss_bwd")
    return dx


def ntxent_fwd(q, k, temperature: float):
    n, d = q.shape
    L = _lib.lib()
    loss = torch.empty(1, dtype=torch.float32, device=q.device)
    lse = torch.empty(2 * n, dtype=torch.float32, device=q.device)
    ws = _ws.get(L.dl_ntxent_workspace_bytes(n, d), q.device)
    check(L.dl_ntxent_fwd(q.data_ptr(), k.data_ptr(), n, d, float(temperature), loss.data_ptr(), lse.data_ptr(),
                          ws.data_ptr(), ws.numel(), _stream()), "dl_ntxent_fwd")
    return loss, lse


def ntxent_bwd(q, k, temperature: float, lse, grad_out: float):
    n, d = q.shape
    dq, dk = torch.empty_like(q), torch.empty_like(k)
    check(_lib.lib().dl_ntxent_bwd(q.data_ptr(), k.data_ptr(), n, d, float(temperature), lse.data_ptr(),
                                   float(grad_out), dq.data_ptr(), dk.data_ptr(), _stream()), "dl_ntxent_bwd")
    return dq, dk


def _ntx_args(aq, ak, bq, bk, off_a, off_b, n_global, temperature, lse_a=None, lse_b=None):
    _need_gpu(aq, ak, bq, bk)
    assert aq.dtype == ak.dtype == bq.dtype == bk.dtype and aq.dtype in (torch.float32, torch.bfloat16)
    assert aq.is_contiguous() and ak.is_contiguous() and bq.is_contiguous() and bk.is_contiguous()
    assert aq.shape == ak.shape and bq.shape == bk.shape and aq.shape[1] == bq.shape[1]
    a = _lib.NtxentArgs()
    a.a.q, a.a.k, a.a.n, a.a.gid_offset = aq.data_ptr(), ak.data_ptr(), aq.shape[0], int(off_a)
    a.b.q, a.b.k, a.b.n, a.b.gid_offset = bq.data_ptr(), bk.data_ptr(), bq.shape[0], int(off_b)
    a.a.lse = lse_a.data_ptr() if lse_a is not None else None
    a.b.lse = lse_b.data_ptr() if lse_b is not None else None
    a.n_global, a.d, a.dtype, a.temperature = int(n_global), aq.shape[1], _dt(aq), float(temperature)
    return a


def ntxent_fwd_ex(aq, ak, bq, bk, off_a: int, off_b: int, n_global: int, temperature: float):
    """Rows [aq; ak] scored against rows [bq; bk] (dl_ntxent_fwd_ex): returns (mean row loss (1,), row_lse (2 n_a,),
    row_loss (2 n_a,)), all fp32."""
    a = _ntx_args(aq, ak, bq, bk, off_a, off_b, n_global, temperature)
    n2 = 2 * aq.shape[0]
    loss = torch.empty(1, dtype=torch.float32, device=aq.device)
    lse = torch.empty(n2, dtype=torch.float32, device=aq.device)
    row_loss = torch.empty(n2, dtype=torch.float32, device=aq.device)
    check(_lib.lib().dl_ntxent_fwd_ex(a, lse.data_ptr(), row_loss.data_ptr(), loss.data_ptr(), _stream()), "dl_ntxent_fwd_ex")
    return loss, lse, row_loss


def ntxent_bwd_ex(aq, ak, bq, bk, off_a: int, off_b: int, n_global: int, temperature: float, lse_a, lse_b, grad_scale: float):
    """Gradient with respect to the rows [aq; ak] (dl_ntxent_bwd_ex), fp32."""
    a = _ntx_args(aq, ak, bq, bk, off_a, off_b, n_global, temperature, lse_a, lse_b)
    da_q = torch.empty(aq.shape, dtype=torch.float32, device=aq.device)
    da_k = torch.empty(ak.shape, dtype=torch.float32, device=aq.device)
    check(_lib.lib().dl_ntxent_bwd_ex(a, float(grad_scale), da_q.data_ptr(), da_k.data_ptr(), _stream()), "dl_ntxent_bwd_ex")
    return da_q, da_k


def triplet_sigcos_fwd(p_lats, d_lats, gt_i8, margin: float):
    n_p, dim = p_lats.shape
    n_d = d_lats.shape[0]
    L = _lib.lib()
    buf = torch.empty(L.dl_triplet_sigcos_buffer_floats(n_p, n_d), dtype=torch.float32, device=p_lats.device)
    loss = torch.empty(1, dtype=torch.float32, device=p_lats.device)
    ntri = torch.empty(1, dtype=torch.float32, device=p_lats.device)
    check(L.dl_triplet_sigcos_fwd(p_lats.data_ptr(), d_lats.data_ptr(), gt_i8.data_ptr(), n_p, n_d, dim, float(margin),
                                  buf.data_ptr(), loss.data_ptr(), ntri.data_ptr(), _stream()), "dl_triplet_sigcos_fwd")
    return loss, ntri, buf


def triplet_sigcos_bwd(p_lats, d_lats, gt_i8, margin: float, buf, ntri, grad_out: float):
    n_p, dim = p_lats.shape
    n_d = d_lats.shape[0]
    dp, dd = torch.empty_like(p_lats), torch.empty_like(d_lats)
    check(_lib.lib().dl_triplet_sigcos_bwd(p_lats.data_ptr(), d_lats.data_ptr(), gt_i8.data_ptr(), buf.data_ptr(), n_p,
                                           n_d, dim, float(margin), ntri.data_ptr(), float(grad_out), dp.data_ptr(),
                                           dd.data_ptr(), _stream()), "dl_triplet_sigcos_bwd")
    return dp, dd


def bn_stats(y2d, win, halo, valid, rw=None):
    """rw (R,) fp32 row weights (< 0 halo, 0 context, m >= 1 multiplicity) replace the window rule (dl_bn_stats_rw)."""
    R, Cc = y2d.shape
    L = _lib.lib()
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=y2d.device)
    ws = _ws2.get(L.dl_bn_workspace_bytes(R, Cc), y2d.device)
    if rw is not None:
        check(L.dl_bn_stats_rw(y2d.data_ptr(), R, Cc, rw.data_ptr(), _dt(y2d), sums.data_ptr(), ws.data_ptr(), ws.numel(),
                               _stream()), "dl_bn_stats_rw")
        return sums
    check(L.dl_bn_stats(y2d.data_ptr(), R, Cc, win, halo, valid, _dt(y2d), sums.data_ptr(), ws.data_ptr(), ws.numel(),
                        _stream()), "dl_bn_stats")
    return sums


def bn_stats_finalize(y2d, win, halo, valid, n, eps, momentum=0.0, running_mean=None, running_var=None, rw=None):
    """bn_stats + bn_finalize in two launches instead of three (dl_bn_stats_finalize): (mean, biased var, rstd), bit-identical."""
    R, Cc = y2d.shape
    L = _lib.lib()
    out = torch.empty((3, Cc), dtype=torch.float32, device=y2d.device)
    ws = _ws2.get(L.dl_bn_workspace_bytes(R, Cc), y2d.device)
    check(L.dl_bn_stats_finalize(y2d.data_ptr(), R, Cc, win, halo, valid, _ptr(rw), _dt(y2d), int(n), float(eps), float(momentum), None,
                                 out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), _ptr(running_mean), _ptr(running_var),
                                 ws.data_ptr(), ws.numel(), _stream()), "dl_bn_stats_finalize")
    return out[0], out[1], out[2]


def bn_finalize(sums, n, eps, momentum=0.0, running_mean=None, running_var=None):
    """(mean, biased var, rstd) from dl_bn_stats sums; updates the running statistics in place when given."""
    Cc = sums.numel() // 2
    out = torch.empty((3, Cc), dtype=torch.float32, device=sums.device)
    check(_lib.lib().dl_bn_finalize(sums.data_ptr(), int(n), float(eps), float(momentum), out[0].data_ptr(), out[1].data_ptr(),
                                    out[2].data_ptr(), _ptr(running_mean), _ptr(running_var), Cc, _stream()), "dl_bn_finalize")
    return out[0], out[1], out[2]


def bn_apply_fwd(y2d, mean, rstd, gamma, beta, win, halo, valid, rw=None):
    R, Cc = y2d.shape
    z = torch.empty_like(y2d)
    if rw is not None:
        check(_lib.lib().dl_bn_apply_fwd_rw(y2d.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                            beta.data_ptr(), R, Cc, rw.data_ptr(), _dt(y2d), _stream()), "dl_bn_apply_fwd_rw")
        return z
    check(_lib.lib().dl_bn_apply_fwd(y2d.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                     beta.data_ptr(), R, Cc, win, halo, valid, _dt(y2d), _stream()), "dl_bn_apply_fwd")
    return z


def bn_bwd_reduce(dz2d, y2d, mean, rstd, win, halo, valid, rw=None):
    R, Cc = y2d.shape
    L = _lib.lib()
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=y2d.device)
    ws = _ws2.get(L.dl_bn_workspace_bytes(R, Cc), y2d.device)
    if rw is not None:
        check(L.dl_bn_bwd_reduce_rw(dz2d.data_ptr(), y2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), R, Cc, rw.data_ptr(),
                                    _dt(y2d), sums.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "dl_bn_bwd_reduce_rw")
        return sums
    check(L.dl_bn_bwd_reduce(dz2d.data_ptr(), y2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), R, Cc, win, halo, valid,
                             _dt(y2d), sums.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "dl_bn_bwd_reduce")
    return sums


def bn_bwd_apply(dz2d, y2d, mean, rstd, gamma, sums, inv_n, relu_mask, win, halo, valid, rw=None):
    R, Cc = y2d.shape
    dy = torch.empty_like(y2d)
    if rw is not None:
        check(_lib.lib().dl_bn_bwd_apply_rw(dz2d.data_ptr(), y2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                            sums.data_ptr(), float(inv_n), int(relu_mask), dy.data_ptr(), R, Cc, rw.data_ptr(),
                                            _dt(y2d), _stream()), "dl_bn_bwd_apply_rw")
        return dy
    check(_lib.lib().dl_bn_bwd_apply(dz2d.data_ptr(), y2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                     sums.data_ptr(), float(inv_n), int(relu_mask), dy.data_ptr(), R, Cc, win, halo, valid,
                                     _dt(y2d), _stream()), "dl_bn_bwd_apply")
    return dy


def bn_apply_relu_fwd(y2d, mean, rstd, gamma, beta):
    """z = max(0, BN(y)) (dl_bn_apply_relu_fwd)."""
    R, Cc = y2d.shape
    z = torch.empty_like(y2d)
    check(_lib.lib().dl_bn_apply_relu_fwd(y2d.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          R, Cc, _dt(y2d), _stream()), "dl_bn_apply_relu_fwd")
    return z


def bn_relu_bwd(dz2d, y2d, mean, rstd, gamma, beta, inv_n):
    """Backward of z = max(0, BN(y)): (dy, sums [2C] = (d beta | d gamma)) (dl_bn_relu_bwd)."""
    R, Cc = y2d.shape
    L = _lib.lib()
    dy = torch.empty_like(y2d)
    sums = torch.empty(2 * Cc, dtype=torch.float32, device=y2d.device)
    ws = _ws2.get(L.dl_bn_workspace_bytes(R, Cc), y2d.device)
    check(L.dl_bn_relu_bwd(dz2d.data_ptr(), y2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                           float(inv_n), dy.data_ptr(), sums.data_ptr(), R, Cc, _dt(y2d), ws.data_ptr(), ws.numel(), _stream()),
          "dl_bn_relu_bwd")
    return dy, sums


def bn_tail_fix(dy2d, y2d, mean, rstd, gamma, sums, inv_n, w, win, lead):
    """In place: the tail rows of every `win`-row window take the mean terms of the BatchNorm backward w times (dl_bn_tail_fix)."""
    R, Cc = y2d.shape
    check(_lib.lib().dl_bn_tail_fix(dy2d.data_ptr(), y2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                    sums.data_ptr(), float(inv_n), int(w), R, Cc, int(win), int(lead), _dt(y2d), _stream()),
          "dl_bn_tail_fix")
    return dy2d


def ce_rows_fwd(logits2d, labels, C: int, ignore_index: int):
    """Cross entropy over the first C columns of logits2d (N, >= C) (dl_ce_rows_fwd): (out2 = [mean loss, counted rows], lse (N,))."""
    _need_gpu(logits2d, labels)
    N, ld = logits2d.shape[0], logits2d.stride(0)
    L = _lib.lib()
    lse = torch.empty(N, dtype=torch.float32, device=logits2d.device)
    out2 = torch.empty(2, dtype=torch.float32, device=logits2d.device)
    ws = _ws2.get(L.dl_ce_rows_workspace_bytes(N), logits2d.device)
    check(L.dl_ce_rows_fwd(logits2d.data_ptr(), ld, labels.data_ptr(), N, C, int(ignore_index), _dt(logits2d), lse.data_ptr(),
                           out2.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "dl_ce_rows_fwd")
    return out2, lse


def ce_rows_bwd(logits2d, labels, C: int, ignore_index: int, lse, out2, grad_out):
    """d(mean loss) / d logits, (N, width of logits2d): zeros in the columns >= C and in ignored rows (dl_ce_rows_bwd)."""
    N, Cp = logits2d.shape
    d = torch.empty((N, Cp), dtype=logits2d.dtype, device=logits2d.device)
    check(_lib.lib().dl_ce_rows_bwd(logits2d.data_ptr(), logits2d.stride(0), labels.data_ptr(), N, C, int(ignore_index), _dt(logits2d),
                                    lse.data_ptr(), out2.data_ptr(), grad_out.data_ptr(), d.data_ptr(), Cp, Cp, _stream()), "dl_ce_rows_bwd")
    return d


def gate_dpre(dl2d, w2, pre2d):
    """dpre = gelu'(pre) o (dl @ w2): dl (M, 8), w2 (8, d_diff), pre (M, d_diff), all in one compute dtype (dl_gate_dpre)."""
    _need_gpu(dl2d, w2, pre2d)
    M, H = dl2d.shape
    dd = pre2d.shape[1]
    out = torch.empty_like(pre2d)
    check(_lib.lib().dl_gate_dpre(dl2d.data_ptr(), w2.data_ptr(), pre2d.data_ptr(), out.data_ptr(), M, dd, H, _dt(pre2d), _stream()),
          "dl_gate_dpre")
    return out


def gelu_bwd(dy2d, pre2d):
    dx = torch.empty_like(dy2d)
    check(_lib.lib().dl_gelu_bwd(dy2d.data_ptr(), pre2d.data_ptr(), dx.data_ptr(), dy2d.numel(), _dt(dy2d), _stream()),
          "dl_gelu_bwd")
    return dx


def fill_pool(x: torch.Tensor, site_len: int, out_dtype: torch.dtype):
    """x (B, S, F) -> (fill (B, S) in x.dtype, pooled (B, S // site_len, ceil8(F + 1)) in out_dtype)."""
    _need_gpu(x)
    x = x.contiguous()
    B, S, F = x.shape
    Fp = (F + 1 + 7) // 8 * 8
    fill = torch.empty((B, S), dtype=x.dtype, device=x.device)
    pooled = torch.empty((B, S // site_len, Fp), dtype=out_dtype, device=x.device)
    check(_lib.lib().dl_fill_pool(x.data_ptr(), fill.data_ptr(), pooled.data_ptr(), B, S, F, site_len, _dt(x),
                                  _DT[out_dtype], _stream()), "dl_fill_pool")
    return fill, pooled


_weight_prep_launches = 0


def weight_prep_launches() -> int:
    """dl_weight_prep launches issued so far (GraphedStep checks that its capture recorded one)."""
    return _weight_prep_launches


def weight_prep(items_dev: torch.Tensor, block_map_dev: torch.Tensor, n_blocks: int, out_dtype: torch.dtype) -> None:
    """One launch over a device-side list of (fp32 master -> image) copies; see dl_weight_prep."""
    global _weight_prep_launches
    _need_gpu(items_dev, block_map_dev)
    _weight_prep_launches += 1
    check(_lib.lib().dl_weight_prep(items_dev.data_ptr(), block_map_dev.data_ptr(), int(n_blocks), _DT[out_dtype],
                                    _stream()), "dl_weight_prep")


def cnn_sitepool_fwd(z: torch.Tensor, L: int, halo: int, site_len: int) -> torch.Tensor:
    """z (B, L + 2*halo, C) channel-last -> pooled (B, L // site_len, C); see dl_cnn_sitepool_fwd."""
    _need_gpu(z)
    B, LP, C = z.shape
    out = torch.empty((B, L // site_len, C), dtype=z.dtype, device=z.device)
    check(_lib.lib().dl_cnn_sitepool_fwd(z.data_ptr(), out.data_ptr(), B, L, C, halo, site_len, _dt(z), _stream()),
          "dl_cnn_sitepool_fwd")
    return out


def cnn_sitepool_bwd(dpooled: torch.Tensor, L: int, halo: int, site_len: int) -> torch.Tensor:
    _need_gpu(dpooled)
    dpooled = dpooled.contiguous()
    B, _, C = dpooled.shape
    dz = torch.empty((B, L + 2 * halo, C), dtype=dpooled.dtype, device=dpooled.device)
    check(_lib.lib().dl_cnn_sitepool_bwd(dpooled.data_ptr(), dz.data_ptr(), B, L, C, halo, site_len, _dt(dpooled),
                                         _stream()), "dl_cnn_sitepool_bwd")
    return dz


def cnn_sitepool_rows_supported(L: int, site_len: int, C: int, dtype: torch.dtype) -> bool:
    """The shape rules of dl_cnn_sitepool_rows_fwd / _bwd (bf16; the forward's strip of site_len x 32 rows and the backward's
    image of one sample's pooled gradient must fit the 160 KB of LDS: 256 sites at 128 channels do, the 1024 sites of
    PROTEIN.SEQ_LEN 9216 do not — such shapes take the expansion + dense pooling kernels)."""
    if dtype != torch.bfloat16 or site_len <= 0 or L % site_len or C % 8 or site_len > C:
        return False
    n_site = L // site_len
    return n_site % 8 == 0 and site_len * C * 34 * 2 <= 160 * 1024 and (n_site + 2) * C * 2 + C * 16 + 16 * 1024 <= 160 * 1024


def cnn_sitepool_rows_fwd(z2d: torch.Tensor, row_of: torch.Tensor, B: int, L: int, site_len: int) -> torch.Tensor:
    """compact rows z2d (R, C) + position map -> pooled (B, L // site_len, C); see dl_cnn_sitepool_rows_fwd."""
    _need_gpu(z2d, row_of)
    z2d = z2d.contiguous()
    Cc = z2d.shape[1]
    out = torch.empty((B, L // site_len, Cc), dtype=z2d.dtype, device=z2d.device)
    check(_lib.lib().dl_cnn_sitepool_rows_fwd(z2d.data_ptr(), row_of.data_ptr(), out.data_ptr(), B, L, Cc, site_len, _dt(z2d), _stream()),
          "dl_cnn_sitepool_rows_fwd")
    return out


def cnn_sitepool_rows_bwd(dpooled: torch.Tensor, rep: torch.Tensor, row_of: torch.Tensor, L: int, site_len: int) -> torch.Tensor:
    _need_gpu(dpooled, rep, row_of)
    dpooled = dpooled.contiguous()
    B, _, Cc = dpooled.shape
    R = rep.shape[0]
    dz = torch.empty((R, Cc), dtype=dpooled.dtype, device=dpooled.device)
    check(_lib.lib().dl_cnn_sitepool_rows_bwd(dpooled.data_ptr(), rep.data_ptr(), row_of.data_ptr(), dz.data_ptr(), B, L, Cc, R, site_len,
                                              _dt(dpooled), _stream()), "dl_cnn_sitepool_rows_bwd")
    return dz


def embed_pad(ids: torch.Tensor, weight: torch.Tensor, fill: torch.Tensor, halo: int) -> torch.Tensor:
    """ids (B, L) int64, weight (V, D), fill (B, L) -> (B, L + 2*halo, D + 1); see dl_embed_pad."""
    _need_gpu(ids, weight, fill)
    B, L = ids.shape
    V, D = weight.shape
    ids = ids.contiguous()
    weight = torch.nn.functional.pad(weight, (0, 1)).contiguous()      # [V][D + 1]: 16-byte aligned table rows
    fill = fill.to(weight.dtype).contiguous()
    out = torch.empty((B, L + 2 * halo, D + 1), dtype=weight.dtype, device=weight.device)
    check(_lib.lib().dl_embed_pad(ids.data_ptr(), weight.data_ptr(), fill.data_ptr(), out.data_ptr(), B, L, V, D, halo,
                                  _dt(weight), _stream()), "dl_embed_pad")
    return out


# Device-side guard flags (dl_embed_rows / dl_rows_equal_check OR sticky bits into one int32 word per device): the compact
# padding forms are only valid for inputs with the padding structure of the reference's collate; Trainer polls the word.
_guard_flags = {}
FLAG_TEXT = {_lib.FLAG_PROT_PERIOD: "a protein's residue codes / fill bits are not tiled with the period its length gives "
                                    "(utils.py:392-412 repeat_integer_label_protein); disable with DL_CNN_COMPACT=0",
             _lib.FLAG_DRUG_TOKEN_PAD: "drug LLM token rows beyond the hinted block (meta 'Drug_Tokens') are not identical padding rows; "
                                       "disable with DL_PAD_COMPACT=0",
             _lib.FLAG_GCN_NODE_PAD: "drug graph nodes beyond the adjacency block are not identical virtual padding nodes "
                                     "(handler/dataset.py:216-221); disable with DL_GCN_COMPACT=0",
             _lib.FLAG_PLAN_ROWS: "the ProteinCNN row tables were built with fewer rows than the batch's residue counts need "
                                  "(dl_protein_plan_build capacity)"}


def guard_flags(device) -> torch.Tensor:
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))
    t = _guard_flags.get(key)
    if t is None:
        t = _guard_flags[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


def guard_text(bits: int) -> str:
    return "; ".join(v for k, v in FLAG_TEXT.items() if bits & k) or "unknown flag bits %#x" % bits


def check_guard_flags(device) -> None:
    """Synchronising check of the device guard word: raises (and clears the word) when a compact padding form ran on a batch
    that does not have the structure it assumes."""
    t = guard_flags(device)
    bits = int(t.item())
    if bits:
        t.zero_()
        raise RuntimeError("druglamp_amd: padding guard tripped — " + guard_text(bits))


def embed_rows(ids: torch.Tensor, weight: torch.Tensor, fill: torch.Tensor, src: torch.Tensor, period: Optional[torch.Tensor]) -> torch.Tensor:
    """Compact ProteinCNN input (dl_embed_rows): ids (B, L) int64, weight (V, D), fill (B, L), src (R,) int32 flat indices or -1
    -> (R, D + 1).  period (B,) int32: also run the periodicity guard (sets the device guard word on a violation)."""
    _need_gpu(ids, weight, fill, src)
    B, L = ids.shape
    V, D = weight.shape
    ids = ids.contiguous()
    weight = torch.nn.functional.pad(weight, (0, 1)).contiguous()
    fill = fill.to(weight.dtype).contiguous()
    R = src.numel()
    out = torch.empty((R, D + 1), dtype=weight.dtype, device=weight.device)
    flags = guard_flags(weight.device) if period is not None else None
    check(_lib.lib().dl_embed_rows(ids.data_ptr(), weight.data_ptr(), fill.data_ptr(), src.data_ptr(), out.data_ptr(), R, V, D,
                                   _ptr(period), B, L, _ptr(flags), _dt(weight), _stream()), "dl_embed_rows")
    return out


def protein_plan_build(pd) -> None:
    """The ProteinCNN distinct-row tables of protein_plan.PlanDev `pd` from its residue counts `pd.len_dev`, on the device
    (dl_protein_plan_build, current stream); a capacity overflow sets the device guard word."""
    _need_gpu(pd.buf)
    check(_lib.lib().dl_protein_plan_build(pd.len_dev.data_ptr(), pd.B, pd.S, pd.rows, pd.src.data_ptr(), pd.w.data_ptr(),
                                           pd.rep.data_ptr(), pd.row_of.data_ptr(), pd.period.data_ptr(),
                                           guard_flags(pd.buf.device).data_ptr(), _stream()), "dl_protein_plan_build")


def rows_gather(src2d: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """out[i] = src2d[index[i]] (zeros where index[i] < 0); index int32 (dl_rows_gather)."""
    _need_gpu(src2d, index)
    src2d = src2d.contiguous()
    N, Cc = index.numel(), src2d.shape[1]
    out = torch.empty((N, Cc), dtype=src2d.dtype, device=src2d.device)
    check(_lib.lib().dl_rows_gather(src2d.data_ptr(), index.data_ptr(), out.data_ptr(), N, Cc * src2d.element_size(), _stream()),
          "dl_rows_gather")
    return out


def rows_sum_strided(x2d: torch.Tensor, rep: torch.Tensor) -> torch.Tensor:
    """out[r] = sum_{k < rep[r, 2]} x2d[rep[r, 0] + k * rep[r, 1]]; rep (R, 3) int32 (dl_rows_sum_strided)."""
    _need_gpu(x2d, rep)
    x2d = x2d.contiguous()
    R, Cc = rep.shape[0], x2d.shape[1]
    out = torch.empty((R, Cc), dtype=x2d.dtype, device=x2d.device)
    check(_lib.lib().dl_rows_sum_strided(x2d.data_ptr(), rep.data_ptr(), out.data_ptr(), R, Cc, _dt(x2d), _stream()),
          "dl_rows_sum_strided")
    return out


def rows_equal_check(x3d: torch.Tensor, row0: int, code: int) -> None:
    """Guard: rows row0.. of every sample of x3d (B, N, F) must equal row row0 of sample 0 (dl_rows_equal_check)."""
    _need_gpu(x3d)
    x3d = x3d.contiguous()
    B, N, F_ = x3d.shape
    check(_lib.lib().dl_rows_equal_check(x3d.data_ptr(), B, N, F_ * x3d.element_size(), int(row0), int(code),
                                         guard_flags(x3d.device).data_ptr(), _stream()), "dl_rows_equal_check")


def interleave_streams(x: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """x [S, ..., d] -> [..., S * d] (cat of the streams along the last dim); inverse: x [..., S * d] with S = 2 -> [2, ..., d]."""
    _need_gpu(x)
    x = x.contiguous()
    if not inverse:
        S, d = x.shape[0], x.shape[-1]
        R = x[0].numel() // d
        out = torch.empty(x.shape[1:-1] + (S * d,), dtype=x.dtype, device=x.device)
    else:
        S, d = 2, x.shape[-1] // 2
        R = x.numel() // (S * d)
        out = torch.empty((S,) + x.shape[:-1] + (d,), dtype=x.dtype, device=x.device)
    check(_lib.lib().dl_interleave_streams(x.data_ptr(), out.data_ptr(), R, d * x.element_size(), S, int(inverse), _stream()),
          "dl_interleave_streams")
    return out


def norm_adjacency(adj: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(B, n, n) fp32 adjacency -> normalised, transposed adjacency in `dtype` (dl_norm_adjacency)."""
    _need_gpu(adj)
    adj = adj.contiguous()
    B, n, _ = adj.shape
    out = torch.empty((B, n, n), dtype=dtype, device=adj.device)
    check(_lib.lib().dl_norm_adjacency(adj.data_ptr(), out.data_ptr(), B, n, _DT[dtype], _stream()), "dl_norm_adjacency")
    return out


def graph_aggregate(ahat: torch.Tensor, feat: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """ahat (B, n, n), feat (B, N, C) -> (B, N, C): [A' @ feat[:, :n] ; feat[:, n:]] with A' = ahat or ahat^T (dl_graph_aggregate)."""
    _need_gpu(ahat, feat)
    ahat, feat = ahat.contiguous(), feat.contiguous()
    B, N, Cc = feat.shape
    out = torch.empty_like(feat)
    check(_lib.lib().dl_graph_aggregate(ahat.data_ptr(), feat.data_ptr(), out.data_ptr(), B, ahat.shape[-1], N, Cc, int(bool(transpose)),
                                        _dt(feat), _stream()), "dl_graph_aggregate")
    return out


def concat2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """cat((a, b), -1) for two tensors with equal leading shape."""
    _need_gpu(a, b)
    a, b = a.contiguous(), b.contiguous()
    R = a.numel() // a.shape[-1]
    out = torch.empty(a.shape[:-1] + (a.shape[-1] + b.shape[-1],), dtype=a.dtype, device=a.device)
    check(_lib.lib().dl_concat2(a.data_ptr(), b.data_ptr(), out.data_ptr(), R, a.shape[-1] * a.element_size(),
                                b.shape[-1] * b.element_size(), 0, _stream()), "dl_concat2")
    return out


def split2(cat: torch.Tensor, wa: int):
    """inverse of concat2: (cat[..., :wa], cat[..., wa:]) as two contiguous tensors."""
    _need_gpu(cat)
    cat = cat.contiguous()
    wb = cat.shape[-1] - wa
    R = cat.numel() // cat.shape[-1]
    a = torch.empty(cat.shape[:-1] + (wa,), dtype=cat.dtype, device=cat.device)
    b = torch.empty(cat.shape[:-1] + (wb,), dtype=cat.dtype, device=cat.device)
    check(_lib.lib().dl_concat2(a.data_ptr(), b.data_ptr(), cat.data_ptr(), R, wa * cat.element_size(), wb * cat.element_size(),
                                1, _stream()), "dl_concat2")
    return a, b


def gather_pad(store: torch.Tensor, offsets: torch.Tensor, lengths: torch.Tensor, S: int, repeat: bool) -> torch.Tensor:
    """store (rows, F), offsets (B,) int64, lengths (B,) int32 -> (B, S, F); see dl_gather_pad."""
    _need_gpu(store, offsets, lengths)
    B, F = offsets.numel(), store.shape[1]
    out = torch.empty((B, S, F), dtype=store.dtype, device=store.device)
    check(_lib.lib().dl_gather_pad(store.data_ptr(), offsets.data_ptr(), lengths.data_ptr(), out.data_ptr(), B, S, F,
                                   int(bool(repeat)), _dt(store), _stream()), "dl_gather_pad")
    return out
