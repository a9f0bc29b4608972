"""ctypes binding of libdruglamp_hip.so (C ABI: include/druglamp_hip.h).

The product path has no CPU fallback: `lib()` raises if the shared object is missing, and every
wrapper raises RuntimeError with dl_last_error() on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported BEFORE the dlopen below: the library then binds to the HIP runtime
#                            torch already loaded instead of bringing up a second, device-less copy)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DL_USE_STUDY_LIB=1 (tools/ only) loads the -DDL_STUDY build, the only one that reads study switches from the environment
_study = os.environ.get("DL_USE_STUDY_LIB", "")
LIB_PATH = os.path.join(_HERE, "lib", "libdruglamp_hip_study.so" if _study == "1" else _study if _study.endswith(".so")   # (a named variant build: same-box A/B of compile-time choices, tools only)
                        else "libdruglamp_hip.so")

DL_F32, DL_BF16 = 0, 1

c_i64, c_i32, c_f32, c_u64, c_vp, c_sz = C.c_int64, C.c_int32, C.c_float, C.c_uint64, C.c_void_p, C.c_size_t


class ReduceItem(C.Structure):
    """dl_reduce_item: a pending second-stage reduction (include/druglamp_hip.h)."""
    _fields_ = [
        ("kind", c_i32), ("out_dtype", c_i32),
        ("src", c_vp), ("out", c_vp),
        ("mn", c_i64), ("ldc", c_i64),
        ("N", c_i32), ("splits", c_i32), ("accumulate", c_i32), ("M", c_i32),
        ("cs_slabs", c_vp), ("cs_out", c_vp),
    ]


REDUCE_BATCH_MAX = 24
GEMM_GROUP_MAX = 16


class GemmArgs(C.Structure):
    _fields_ = [
        ("X", c_vp), ("ldx", c_i64), ("x_kslow", c_i32),
        ("W", c_vp), ("ldw", c_i64), ("w_kslow", c_i32),
        ("C", c_vp), ("ldc", c_i64),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("in_dtype", c_i32), ("out_dtype", c_i32),
        ("bias", c_vp),
        ("residual", c_vp), ("ldr", c_i64),
        ("res_row_mod", c_i64),
        ("res_before_dropout", c_i32),
        ("act", c_i32),
        ("pre_out", c_vp), ("ldp", c_i64),
        ("dact_pre", c_vp), ("lddp", c_i64),
        ("dropout_p", c_f32), ("dropout_seed", c_u64),
        ("accumulate", c_i32),
        ("split_k", c_i32), ("workspace", c_vp), ("workspace_bytes", c_sz),
        ("x_colsum", c_vp),
        ("dropout_seed_offset", c_vp),
        ("algo", c_i32),
        ("tile_tickets", c_vp),
        ("deferred", C.POINTER(ReduceItem)),
        ("prof_tag", c_i32),
    ]


FLAG_PROT_PERIOD, FLAG_DRUG_TOKEN_PAD, FLAG_GCN_NODE_PAD, FLAG_PLAN_ROWS = 1, 2, 4, 8
TAG_OTHER, TAG_QKV_OUT, TAG_FFN, TAG_CONV, TAG_WGRAD, TAG_ADAPTOR = 0, 1, 2, 3, 4, 5
TAG_NAMES = {0: "other", 1: "qkv_out", 2: "ffn", 3: "conv", 4: "wgrad", 5: "adaptor"}


class AttnFwdArgs(C.Structure):
    _fields_ = [
        ("Q", c_vp), ("K", c_vp), ("V", c_vp), ("O", c_vp), ("LSE", c_vp), ("raw_logits", c_vp),
        ("q_ps", c_i64), ("q_hs", c_i64), ("q_rs", c_i64),
        ("k_ps", c_i64), ("k_hs", c_i64), ("k_rs", c_i64),
        ("v_ps", c_i64), ("v_hs", c_i64), ("v_rs", c_i64),
        ("o_ps", c_i64), ("o_hs", c_i64), ("o_rs", c_i64), ("o_ss", c_i64),
        ("n_problems", c_i32), ("n_heads", c_i32), ("n_segments", c_i32), ("partner_shift", c_i32),
        ("Lq", c_i32), ("Lk", c_i32), ("head_dim", c_i32), ("dtype", c_i32),
        ("scale", c_f32),
        ("algo", c_i32),
        ("key_tail_rows", c_i32), ("key_tail_weight", c_f32),
    ]


class AttnBwdArgs(C.Structure):
    _fields_ = [
        ("Q", c_vp), ("K", c_vp), ("V", c_vp), ("O", c_vp), ("dO", c_vp),
        ("LSE", c_vp), ("Delta", c_vp),
        ("dQ", c_vp), ("dK", c_vp), ("dV", c_vp),
        ("q_ps", c_i64), ("q_hs", c_i64), ("q_rs", c_i64),
        ("k_ps", c_i64), ("k_hs", c_i64), ("k_rs", c_i64),
        ("v_ps", c_i64), ("v_hs", c_i64), ("v_rs", c_i64),
        ("o_ps", c_i64), ("o_hs", c_i64), ("o_rs", c_i64), ("o_ss", c_i64),
        ("do_ps", c_i64), ("do_hs", c_i64), ("do_rs", c_i64), ("do_ss", c_i64),
        ("dq_ps", c_i64), ("dq_hs", c_i64), ("dq_rs", c_i64),
        ("dk_ps", c_i64), ("dk_hs", c_i64), ("dk_rs", c_i64),
        ("dv_ps", c_i64), ("dv_hs", c_i64), ("dv_rs", c_i64),
        ("n_problems", c_i32), ("n_heads", c_i32), ("n_segments", c_i32), ("partner_shift", c_i32),
        ("Lq", c_i32), ("Lk", c_i32), ("head_dim", c_i32), ("dtype", c_i32),
        ("scale", c_f32),
        ("algo", c_i32),
        ("key_tail_rows", c_i32), ("key_tail_weight", c_f32),
    ]


class NtxentSide(C.Structure):
    _fields_ = [("q", c_vp), ("k", c_vp), ("n", c_i64), ("gid_offset", c_i64), ("lse", c_vp)]


class NtxentArgs(C.Structure):
    _fields_ = [("a", NtxentSide), ("b", NtxentSide), ("n_global", c_i64), ("d", c_i64), ("dtype", c_i32),
                ("temperature", c_f32)]


# name -> (restype, argtypes); every symbol include/druglamp_hip.h declares
SIGNATURES = {
    "dl_last_error": (C.c_char_p, []),
    "dl_version": (c_i32, []),
    "dl_gemm_workspace_bytes": (c_sz, [C.POINTER(GemmArgs)]),
    "dl_gemm": (c_i32, [C.POINTER(GemmArgs), c_vp]),
    "dl_gemm_pair": (c_i32, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), c_vp]),
    "dl_gemm_group_plan": (c_i32, [C.POINTER(GemmArgs), c_i32, C.POINTER(c_i32)]),
    "dl_gemm_group": (c_i32, [C.POINTER(GemmArgs), c_i32, c_vp]),
    "dl_reduce_batch": (c_i32, [C.POINTER(ReduceItem), c_i32, c_vp]),
    "dl_colsum": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_i32, c_vp, c_i32, c_vp, c_sz, c_vp]),
    "dl_colsum_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "dl_bn_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "dl_bn_stats": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "dl_bn_apply_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "dl_bn_bwd_reduce": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "dl_bn_tail_fix": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "dl_bn_apply_relu_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp]),
    "dl_bn_relu_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_sz, c_vp]),
    "dl_bn_bwd_apply": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64,
                                c_i32, c_vp]),
    "dl_layernorm_fwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "dl_layernorm_bwd_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "dl_layernorm_bwd": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp,
                                 c_i32, c_i64, c_i64, c_i32, c_vp, c_sz, C.POINTER(ReduceItem), c_vp]),
    "dl_attn_fwd": (c_i32, [C.POINTER(AttnFwdArgs), c_vp]),
    "dl_attn_bwd": (c_i32, [C.POINTER(AttnBwdArgs), c_vp]),
    "dl_token_gate_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "dl_token_gate_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "dl_gate_dpre": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "dl_add_rowmod_dropout": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_u64, c_vp, c_i32, c_vp]),
    "dl_dropout_apply": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_f32, c_u64, c_vp, c_i32, c_vp]),
    "dl_fill_pool": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "dl_cnn_sitepool_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "dl_cnn_sitepool_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "dl_bn_stats_finalize": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_i32, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dl_bn_finalize": (c_i32, [c_vp, c_i64, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dl_interleave_streams": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "dl_norm_adjacency": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "dl_graph_aggregate": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "dl_concat2": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "dl_gather_pad": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "dl_embed_pad": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "dl_weight_prep": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "dl_gelu_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "dl_cast": (c_i32, [c_vp, c_i32, c_vp, c_i32, c_i64, c_vp]),
    "dl_rowmod_sum": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "dl_cos_rowloss_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "dl_cos_rowloss_bwd": (c_i32, [c_vp, c_vp, c_f32, c_vp, c_i64, c_i64, c_vp]),
    "dl_ntxent_fwd_ex": (c_i32, [C.POINTER(NtxentArgs), c_vp, c_vp, c_vp, c_vp]),
    "dl_ntxent_bwd_ex": (c_i32, [C.POINTER(NtxentArgs), c_f32, c_vp, c_vp, c_vp]),
    "dl_ntxent_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "dl_ntxent_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dl_ntxent_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_f32, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "dl_triplet_sigcos_buffer_floats": (c_sz, [c_i64, c_i64]),
    "dl_triplet_sigcos_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "dl_triplet_sigcos_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "dl_adamw_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, c_f32, c_vp,
                              c_i32, c_vp]),
    "dl_bn_stats_rw": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "dl_bn_apply_fwd_rw": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i32, c_vp]),
    "dl_bn_bwd_reduce_rw": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "dl_bn_bwd_apply_rw": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i32, c_vp]),
    "dl_cnn_sitepool_rows_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "dl_cnn_sitepool_rows_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_vp]),
    "dl_embed_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i32, c_vp]),
    "dl_rows_gather": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "dl_rows_sum_strided": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp]),
    "dl_rows_equal_check": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_i64, C.c_uint32, c_vp, c_vp]),
    "dl_ce_rows_workspace_bytes": (c_sz, [c_i64]),
    "dl_ce_rows_fwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "dl_ce_rows_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "dl_protein_plan_build": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dl_prof_enable": (c_i32, [c_i32, c_i32]),
    "dl_prof_collect": (c_i32, [c_i32, C.POINTER(c_i64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
    "dl_prof_totals": (c_i32, [c_i32, C.POINTER(c_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dl_prof_collect_tag": (c_i32, [c_i32, c_i32, C.POINTER(c_i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(c_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None

# ---- pointer audit of captured launches (debug: DL_GRAPH_PTR_AUDIT=1; run by tests/test_graph_step_gpu.py and tools/soak.py) --
# A hipGraph replay re-issues every captured launch with the SAME addresses, so each device pointer a captured launch
# received must stay alive and in place for the life of the graph.  Round 2 found two caches that broke this (shared
# scratch, weight-image tables).  With the audit on, every libdruglamp_hip call made while a stream is capturing records
# the device pointers among its arguments (plain void* arguments and the void* fields of argument blocks);
# trainer.GraphedStep then checks each one against the allocator's snapshot: it must lie in the graph's private pool or
# inside a buffer registered as pinned for the life of the trainer.
AUDIT = os.environ.get("DL_GRAPH_PTR_AUDIT") == "1"
_audit_log = None


def audit_begin() -> None:
    global _audit_log
    _audit_log = []


def audit_end():
    global _audit_log
    log, _audit_log = _audit_log, None
    return log or []


def _struct_ptrs(obj, out, name):
    for fname, ftype in getattr(obj, "_fields_", []):
        v = getattr(obj, fname)
        if ftype is c_vp:
            if v:
                out.append(("%s.%s" % (name, fname), int(v)))
        elif isinstance(v, C.Structure):
            _struct_ptrs(v, out, "%s.%s" % (name, fname))


def _audited(name, fn, argtypes):
    def call(*args):
        if _audit_log is not None:
            import torch
            if torch.cuda.is_current_stream_capturing():
                found = []
                n_items = next((a for a, t in zip(args, argtypes) if t is c_i32), 0) if name == "dl_reduce_batch" else 0
                for i, (a, t) in enumerate(zip(args, argtypes)):
                    if t is c_vp:
                        v = a.value if isinstance(a, C.c_void_p) else a
                        if v:
                            found.append(("arg%d" % i, int(v)))
                    elif isinstance(a, C.Array):
                        for k in range(int(n_items) or len(a)):
                            _struct_ptrs(a[k], found, "arg%d[%d]" % (i, k))
                    elif isinstance(a, C.Structure):
                        _struct_ptrs(a, found, "arg%d" % i)
                    elif hasattr(a, "_obj") and isinstance(getattr(a, "_obj"), C.Structure):     # byref(struct)
                        _struct_ptrs(a._obj, found, "arg%d" % i)
                    elif hasattr(a, "contents") and a:                                            # pointer(struct)
                        try:
                            _struct_ptrs(a.contents, found, "arg%d" % i)
                        except ValueError:
                            pass
                _audit_log.extend((name, w, v) for w, v in found)
        return fn(*args)
    return call


def lib():
    """Load the shared library (once).  Raises if it has not been built — there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libdruglamp_hip.so is missing (%s). Build it with `python -m druglamp_amd.build`; "
                "the HIP hot path has no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
            if AUDIT and name not in ("dl_last_error", "dl_version") and not name.endswith("_bytes") and not name.endswith("_floats") \
                    and not name.startswith("dl_prof_"):
                setattr(L, name, _audited(name, fn, args))
        _lib = L
    return _lib


_DEBUG_SYNC = bool(os.environ.get("DL_DEBUG_SYNC"))


def check(rc: int, what: str = "") -> None:
    if _DEBUG_SYNC:           # debugging aid: surface asynchronous faults at the launch that caused them
        import sys
        import torch
        print("[dl] %s" % what, file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        print("[dl-ok] %s" % what, file=sys.stderr, flush=True)
    if rc != 0:
        msg = lib().dl_last_error()
        raise RuntimeError("druglamp_hip %s failed (status %d): %s" % (what, rc, msg.decode() if msg else "?"))
