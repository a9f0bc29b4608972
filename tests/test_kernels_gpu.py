"""Kernel-level parity through the C ABI against fp64 torch references (same battery as
tools/gpu_probe.py): GEMM layouts/epilogues, LayerNorm, attention fwd/bwd incl. paired segments and
masked tails, MHLA gate, elementwise, AdamW, loss kernels."""
import importlib.util
import os

import pytest

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
