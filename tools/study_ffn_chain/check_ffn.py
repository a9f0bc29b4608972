"""Chained FFN forward (dl_ffn_fwd, round 4; reference model/PMMA/mlp.py:44-50 + block.py:52-60) against the unchained pair
of dl_gemm launches it replaces — bit for bit, with and without dropout — and against an fp64 restatement."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("D,Hd,M", [(512, 2048, 1024), (256, 1024, 896), (512, 2048, 128 * 300 + 128), (256, 1024, 65536)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_chained_ffn_equals_the_two_gemms_bitwise(D, Hd, M, p):
    from druglamp_amd import ops
    torch.manual_seed(D + M)
    dt = torch.bfloat16
    x = torch.randn(M, D, device=DEV).to(dt)
    res = torch.randn(M, D, device=DEV).to(dt)
    w1 = (torch.randn(Hd, D, device=DEV) * D ** -0.5).to(dt)
    w2 = (torch.randn(D, Hd, device=DEV) * Hd ** -0.5).to(dt)
    b1, b2 = torch.randn(Hd, device=DEV) * 0.1, torch.randn(D, device=DEV) * 0.1
    s1, s2 = 12345, 67890
    pre_ref = torch.empty(M, Hd, device=DEV, dtype=dt)
    act = ops.gemm(x, w1, M=M, N=Hd, K=D, bias=b1, act=1, pre_out=pre_ref, dropout_p=p, seed=s1)
    out_ref = ops.gemm(act, w2, M=M, N=D, K=Hd, bias=b2, dropout_p=p, seed=s2, residual=res)
    assert ops.ffn_fwd_ok(x, w1)
    out, pre = ops.ffn_fwd(x, w1, b1, w2, b2, residual=res, dropout_p=p, seed1=s1, seed2=s2)
    assert torch.equal(pre, pre_ref)
    assert torch.equal(out, out_ref), float((out.float() - out_ref.float()).abs().max())
    if M <= 1024 and p == 0.0:
        h = x.double() @ w1.double().t() + b1.double()
        ref = res.double() + torch.nn.functional.gelu(h) @ w2.double().t() + b2.double()
        assert float((out.double() - ref).abs().max() / ref.abs().max()) <= 2e-2
