"""GPU: the fused cross-attention sub-block (csrc/xattn.hip, m3ae_xattn_fwd) against
  (1) an fp32 torch reference of the REFERENCE'S formulation (bert_model.py:253-350 cross branch + :353-364:
      separate Q / K / V projections, softmax(QK^T / 8 + mask), PV, output dense + residual + LayerNorm) computed from
      the same bf16-rounded inputs and weights -- the fused kernels absorb the long side's projection into the short
      side (Q' = Q_h W_k,h etc.), so this is a test of the algebra as much as of the kernels.  The key bias b_k adds the
      same number to every key of a row and drops out of the softmax EXACTLY; the test uses a non-zero b_k;
  (2) the unfused composition (m3ae_gemm + m3ae_attn_fwd + m3ae_gemm + m3ae_layernorm_fwd) under dropout with the same
      seeds: both paths apply the same counter-hash masks (same index conventions), so they agree to bf16 noise.
Tolerance: the output is a LayerNorm output (rms 1); bf16 rounding of the intermediates gives rms errors of ~6e-3 and
max errors of ~4e-2 for BOTH paths; the fused path is held to 1.5x the unfused path's rms error and an absolute 0.08."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from m3ae_amd import ops  # noqa: E402
from m3ae_amd.modules.bert_model import BertAttention  # noqa: E402
from m3ae_amd.param_store import ParamStore  # noqa: E402

D, H = 768, 12


def make(seed=0, wscale=2.0):
    torch.manual_seed(seed)
    att = BertAttention(D, H, 1e-12, cross=True)
    with torch.no_grad():
        for n, p in att.named_parameters():
            if p.dim() == 2:
                p.copy_(torch.randn_like(p) * (wscale / math.sqrt(p.shape[1])))
            elif "LayerNorm.weight" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(0.1 * torch.randn_like(p))   # every bias non-zero, b_k included
    cfg = dict(learning_rate=1e-3, weight_decay=0.01, lr_multiplier_head=1, lr_multiplier_multi_modal=1)
    store = ParamStore(att, cfg, "cuda", torch.bfloat16, weight_units=att.weight_units)
    att.eval()
    return att, store


def reference(att, x, y, mask):
    f = lambda p: p.m3ae_c.float()
    sa, out = att.self, att.output
    q = x.float() @ f(sa.query.weight).t() + sa.query.bias
    k = y.float() @ f(sa.key.weight).t() + sa.key.bias
    v = y.float() @ f(sa.value.weight).t() + sa.value.bias
    B, Lq, Lk = x.shape[0], x.shape[1], y.shape[1]
    sp = lambda t, L: t.view(B, L, H, D // H).permute(0, 2, 1, 3)
    s = sp(q, Lq) @ sp(k, Lk).transpose(-1, -2) / math.sqrt(D // H)
    if mask is not None:
        s = s + mask[:, None, None, :]
    ctx = (torch.softmax(s, -1) @ sp(v, Lk)).permute(0, 2, 1, 3).reshape(B, Lq, D)
    o = ctx @ f(out.dense.weight).t() + out.dense.bias + x.float()
    ln = out.LayerNorm
    return torch.nn.functional.layer_norm(o, (D,), ln.weight, ln.bias, ln.eps)


def run(att, x, y, mask, fused, pdrop=0.0):
    old = ops.XATTN
    ops.XATTN = "auto" if fused else "off"
    try:
        with torch.no_grad():
            return att(x, None, y, mask, pdrop=pdrop).float()
    finally:
        ops.XATTN = old


def rms(t):
    return t.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("B,I", [(3, 577), (1, 577), (2, 145), (5, 325), (2, 640), (130, 33)])
@pytest.mark.parametrize("direction", ["txt<-img", "img<-txt"])
def test_fused_cross_attention_matches_reference_formulation(B, I, direction):
    att, _ = make(seed=B + I)
    T = 32
    xt = torch.randn(B, T, D, device="cuda").to(torch.bfloat16)
    xi = torch.randn(B, I, D, device="cuda").to(torch.bfloat16)
    mt = torch.zeros(B, T, device="cuda")
    mt[:, T - 9:] = -10000.0          # the reference's (1 - mask) * -10000.0
    mt[0, 5:] = -10000.0
    mi = torch.zeros(B, I, device="cuda")
    mi[B - 1, I - I // 5:] = -10000.0
    cases = [(xt, xi, None), (xt, xi, mi)] if direction == "txt<-img" else [(xi, xt, mt), (xi, xt, None)]
    for x, y, mask in cases:
        assert ops.xattn_supported(x.view(-1, D), x.shape[1], y.view(-1, D), y.shape[1], mask, att.block_params())
        ref = reference(att, x, y, mask)
        u, f = run(att, x, y, mask, False), run(att, x, y, mask, True)
        assert torch.isfinite(f).all()
        eu, ef = rms(u - ref), rms(f - ref)
        assert ef < max(1.5 * eu, 8e-3), f"rms error fused {ef:.5f} vs unfused {eu:.5f}"
        assert (f - ref).abs().max().item() < 0.08
        # the masked keys get exactly zero weight: changing the other stream at masked positions changes nothing
        if mask is not None:
            y2 = y.clone()
            y2[mask < 0] = 7.0
            assert torch.equal(run(att, x, y2, mask, True), f)


@pytest.mark.parametrize("direction", ["txt<-img", "img<-txt"])
def test_fused_cross_attention_dropout_uses_the_same_masks_as_the_unfused_path(direction):
    att, _ = make(seed=5)
    B, T, I = 3, 32, 577
    xt = torch.randn(B, T, D, device="cuda").to(torch.bfloat16)
    xi = torch.randn(B, I, D, device="cuda").to(torch.bfloat16)
    mt = torch.zeros(B, T, device="cuda")
    mt[:, T - 9:] = -10000.0
    x, y, mask = (xt, xi, None) if direction == "txt<-img" else (xi, xt, mt)
    ev = run(att, x, y, mask, True)
    ops.set_dropout_seed(77)
    u = run(att, x, y, mask, False, 0.1)
    ops.set_dropout_seed(77)
    f = run(att, x, y, mask, True, 0.1)
    ops.set_dropout_seed(78)
    f2 = run(att, x, y, mask, True, 0.1)
    assert rms(f - u) < 0.012, f"same seeds: rms {rms(f - u):.5f}"
    assert rms(f - ev) > 0.2 and rms(f - f2) > 0.2   # dropout really happened, and depends on the seed


def test_fused_cross_attention_falls_back_on_uncovered_shapes():
    att, _ = make(seed=6)
    P = att.block_params()
    x = torch.randn(2 * 64, D, device="cuda").to(torch.bfloat16)      # 64 text tokens (pre-training): not covered
    y = torch.randn(2 * 577, D, device="cuda").to(torch.bfloat16)
    assert not ops.xattn_supported(x, 64, y, 577, None, P)
    y2 = torch.randn(2 * 1025, D, device="cuda").to(torch.bfloat16)   # 512 px: 1025 image tokens
    x2 = torch.randn(2 * 32, D, device="cuda").to(torch.bfloat16)
    assert not ops.xattn_supported(x2, 32, y2, 1025, None, P)
    out = run(att, x.view(2, 64, D), y.view(2, 577, D), None, True)  # "auto" takes the composition
    assert torch.isfinite(out).all()
