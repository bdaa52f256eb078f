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
    ops.XATTN = "always" if fused else "off"     # "always": also where the product's "auto" rule prefers the composition (64 text tokens)
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


def test_fused_cross_attention_coverage_and_fallback(monkeypatch):
    """Image queries (csrc/xflash.hip) cover 32 and 64 text keys and any number of image tokens; text queries (the whole-row
    score tile) 32 or 64 text tokens and <= 640 image tokens, forward and backward; everything else takes the composition."""
    att, _ = make(seed=6)
    P = att.block_params()
    x = torch.randn(2 * 64, D, device="cuda").to(torch.bfloat16)      # 64 text tokens (pre-training, config.py:121-147)
    y = torch.randn(2 * 577, D, device="cuda").to(torch.bfloat16)
    # covered by the kernels, but the product's "auto" rule leaves 64 text tokens to the composition (no FLOPs saved there)
    assert not ops.xattn_supported(x, 64, y, 577, None, P) and not ops.xattn_supported(y, 577, x, 64, None, P)
    monkeypatch.setattr(ops, "XATTN", "always")
    assert ops.xattn_supported(x, 64, y, 577, None, P, backward=True)
    assert ops.xattn_supported(y, 577, x, 64, None, P, backward=True)
    y2 = torch.randn(2 * 1025, D, device="cuda").to(torch.bfloat16)   # 512 px: 1025 image tokens (configs[4])
    x2 = torch.randn(2 * 32, D, device="cuda").to(torch.bfloat16)
    assert not ops.xattn_supported(x2, 32, y2, 1025, None, P)         # text queries over 1025 keys: the composition
    assert ops.xattn_supported(y2, 1025, x2, 32, None, P, backward=True)
    x3 = torch.randn(2 * 48, D, device="cuda").to(torch.bfloat16)     # 48 text tokens: not covered either way
    assert not ops.xattn_supported(x3, 48, y, 577, None, P) and not ops.xattn_supported(y, 577, x3, 48, None, P)
    out = run(att, x2.view(2, 32, D), y2.view(2, 1025, D), None, True)   # "auto" takes the composition
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("B,T,I", [(2, 64, 577), (3, 32, 1025), (2, 64, 1025), (1, 64, 65)])
def test_image_queries_at_pretraining_and_512px_shapes(B, T, I, monkeypatch):
    """m3ae_xattn_supported widened (round 3): 64 text keys (pre-training) and 1025 image tokens (configs[4]) through the
    one-launch image-query kernel, against the fp32 reference of the reference's formulation."""
    att, _ = make(seed=B + T + I)
    xt = torch.randn(B, T, D, device="cuda").to(torch.bfloat16)
    xi = torch.randn(B, I, D, device="cuda").to(torch.bfloat16)
    mt = torch.zeros(B, T, device="cuda")
    mt[:, T - 11:] = -10000.0
    mt[0, 3:] = -10000.0
    cases = [(xi, xt, mt), (xi, xt, None)]
    if I <= 640:      # text queries (64 tokens) over the image keys: covered up to 640 keys
        mi = torch.zeros(B, I, device="cuda")
        mi[B - 1, I - I // 5:] = -10000.0
        cases += [(xt, xi, None), (xt, xi, mi)]
    monkeypatch.setattr(ops, "XATTN", "always")
    for x, y, mask in cases:
        assert ops.xattn_supported(x.view(-1, D), x.shape[1], y.view(-1, D), y.shape[1], mask, att.block_params())
        ref = reference(att, x, y, mask)
        u, f = run(att, x, y, mask, False), run(att, x, y, mask, True)
        eu, ef = rms(u - ref), rms(f - ref)
        assert torch.isfinite(f).all()
        assert ef < max(1.5 * eu, 8e-3), f"rms error fused {ef:.5f} vs unfused {eu:.5f}"
        assert (f - ref).abs().max().item() < 0.08
        if mask is not None:
            y2 = y.clone()
            y2[mask < 0] = 7.0
            assert torch.equal(run(att, x, y2, mask, True), f)


@pytest.mark.parametrize("pdrop", [0.0, 0.1])
@pytest.mark.parametrize("B,I", [(3, 577), (2, 130), (37, 33)])
def test_one_launch_image_query_kernel_is_bit_identical_to_the_round2_chain(B, I, pdrop):
    """csrc/xflash.hip keeps the scores and probabilities on chip; the round-2 chain (score GEMM + softmax epilogue, P through
    HBM, P V' GEMM) rounds the same values to bf16 at the same places and reduces in the same order -- so the pre-LayerNorm sum,
    the output and the saved probabilities are bit for bit the same, with and without dropout."""
    att, _ = make(seed=B * 7 + I)
    P = att.block_params()
    T = 32
    xt = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
    xi = torch.randn(B * I, D, device="cuda").to(torch.bfloat16)
    mt = torch.zeros(B, T, device="cuda")
    mt[:, T - 9:] = -10000.0
    res = []
    try:
        for legacy in (False, True):
            ops.XATTN_LEGACY_CHAIN = legacy
            ops.set_dropout_seed(1234)
            out, saved = ops.xattn_fwd(xi, B, I, xt, T, mt, P, pdrop, need_bwd=True)
            t = saved[4]
            res.append((out.clone(), t["s"].clone(), t["probs"].clone(), t["probs_drop"].clone() if pdrop > 0 else None))
    finally:
        ops.XATTN_LEGACY_CHAIN = False
    (o1, s1, p1, d1), (o2, s2, p2, d2) = res
    assert torch.equal(p1, p2)
    if pdrop > 0:
        assert torch.equal(d1, d2)
    assert torch.equal(s1, s2) and torch.equal(o1, o2)
    # a forward-only call (no saves) computes the same output
    ops.set_dropout_seed(1234)
    o3, _ = ops.xattn_fwd(xi, B, I, xt, T, mt, P, pdrop, need_bwd=False)
    assert torch.equal(o3, o1)


@pytest.mark.parametrize("stream,I", [("text", 577), ("image", 577), ("image", 145), ("text", 33)])
@pytest.mark.parametrize("pdrop", [0.0, 0.1])
def test_fused_cross_attention_backward_matches_the_composition(stream, I, pdrop):
    """Training through a whole BertCrossLayer (bert_model.py:445-503) with the cross-attention sub-block on the fused
    forward + fused backward (m3ae_xattn_fwd / m3ae_xattn_bwd) against the same layer on the composition (GEMMs + flash
    attention, whose backward is tested against torch in test_gpu_ops.py), same dropout seeds: output, both input gradients
    and every parameter gradient.  The two formulations round different intermediates to bf16, so gradients agree to bf16
    noise: relative L2 error < 2 % per tensor (4 % under dropout); key.bias -- whose gradient is exactly zero (the key bias
    drops out of the softmax) and pure rounding noise in the composition -- is held to that noise level."""
    from m3ae_amd.modules.bert_model import BertCrossLayer
    torch.manual_seed(11)
    layer = BertCrossLayer(D, H, 4 * D, drop_rate=pdrop)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if p.dim() == 2:
                p.copy_(torch.randn_like(p) * (1.5 / math.sqrt(p.shape[1])))
            elif "LayerNorm.weight" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(0.1 * torch.randn_like(p))
    cfg = dict(learning_rate=1e-3, weight_decay=0.01, lr_multiplier_head=1, lr_multiplier_multi_modal=1)
    store = ParamStore(layer, cfg, "cuda", torch.bfloat16, weight_units=layer.weight_units)
    layer.train(pdrop > 0)
    B, T = 3, 32
    xt = torch.randn(B, T, D, device="cuda").to(torch.bfloat16)
    xi = torch.randn(B, I, D, device="cuda").to(torch.bfloat16)
    mt = torch.zeros(B, T, device="cuda")
    mt[:, T - 9:] = -10000.0
    h0, e0, ms, mo = (xt, xi, mt, None) if stream == "text" else (xi, xt, None, mt)
    dy = torch.randn_like(h0)
    res = []
    old = ops.XATTN_TRAIN
    try:
        for mode in ("auto", "off"):
            ops.XATTN_TRAIN = mode
            store.zero_grad()
            ops.set_dropout_seed(99)
            h, e = h0.clone().requires_grad_(True), e0.clone().requires_grad_(True)
            y = layer(h, e, ms, mo)
            y.backward(dy)
            res.append((y.detach().float().clone(), h.grad.float().clone(), e.grad.float().clone(),
                        {n: q.grad.clone() for n, q in layer.named_parameters()}))
    finally:
        ops.XATTN_TRAIN = old
    (yf, dhf, def_, gf), (yu, dhu, deu, gu) = res
    tol = 0.04 if pdrop > 0 else 0.02
    rel = lambda a, b: ((a - b).double().norm() / (b.double().norm() + 1e-30)).item()
    assert rel(yf, yu) < 0.01, ("y", rel(yf, yu))
    assert rel(dhf, dhu) < tol, ("d hidden", rel(dhf, dhu))
    assert rel(def_, deu) < tol, ("d other stream", rel(def_, deu))
    assert torch.isfinite(dhf).all() and torch.isfinite(def_).all()
    for n in gu:
        if n.endswith("self.key.bias"):   # both attentions: exactly zero in exact arithmetic, rounding noise here
            scale = max(gu[n.replace("key.bias", "value.bias")].abs().max().item(), 1e-6)
            assert gf[n].abs().max().item() <= 0.05 * scale and gu[n].abs().max().item() <= 0.05 * scale, n
            continue
        assert rel(gf[n], gu[n]) < tol, (n, rel(gf[n], gu[n]))


def test_fused_training_path_batch_rule():
    """Training takes the fused sub-block from ops.XATTN_TRAIN_MIN_BATCH samples per call on (default 96: measured slower than
    the composition at 32 / 64, faster at 128 / 256, profiles/r02_xattn_batch_rule.log); forward-only calls always take it."""
    from m3ae_amd.modules.bert_model import BertCrossLayer
    layer = BertCrossLayer(D, H, 4 * D, drop_rate=0.0)
    cfg = dict(learning_rate=1e-3, weight_decay=0.01, lr_multiplier_head=1, lr_multiplier_multi_modal=1)
    ParamStore(layer, cfg, "cuda", torch.bfloat16, weight_units=layer.weight_units)
    old = ops.XATTN_TRAIN_MIN_BATCH
    try:
        ops.XATTN_TRAIN_MIN_BATCH = 96
        for B, want in ((4, False), (96, True)):
            xt = torch.randn(B, 32, D, device="cuda").to(torch.bfloat16).requires_grad_(True)
            xi = torch.randn(B, 145, D, device="cuda").to(torch.bfloat16)
            layer(xt, xi, None, None)
            assert layer._bp.fused_cross is want, (B, layer._bp.fused_cross)
            with torch.no_grad():
                layer(xt, xi, None, None)
            assert layer._bp.fused_cross is True
    finally:
        ops.XATTN_TRAIN_MIN_BATCH = old


def _fp32_reference_with_gradients(att, x, y, mask, dy, pdrop, seeds):
    """The reference's formulation (bert_model.py:253-350 cross branch, :353-364) in fp32 torch autograd on the bf16-rounded
    inputs and weights, with the library's exported dropout masks (attention probabilities: rows (b H + h) Lq + q, columns Lk;
    hidden: rows B Lq, columns D).  Returns (out, dx, dy_other, {parameter name: gradient})."""
    sa, so = att.self, att.output
    leaf = {}
    for n, p_ in att.named_parameters():
        leaf[n] = (p_.m3ae_c.float() if p_.dim() == 2 else p_.data.float()).clone().requires_grad_(True)
    xf, yf = x.float().clone().requires_grad_(True), y.float().clone().requires_grad_(True)
    B, Lq, Lk = x.shape[0], x.shape[1], y.shape[1]
    q = xf @ leaf["self.query.weight"].t() + leaf["self.query.bias"]
    k = yf @ leaf["self.key.weight"].t() + leaf["self.key.bias"]
    v = yf @ leaf["self.value.weight"].t() + leaf["self.value.bias"]
    sp = lambda t, L: t.view(B, L, H, D // H).permute(0, 2, 1, 3)
    sc = sp(q, Lq) @ sp(k, Lk).transpose(-1, -2) / math.sqrt(D // H)
    if mask is not None:
        sc = sc + mask[:, None, None, :]
    pr = torch.softmax(sc, -1)
    if pdrop > 0:
        keep = ops.dropout_keep_mask(B * H * Lq, Lk, pdrop, seeds[0]).float().view(B, H, Lq, Lk)
        pr = pr * keep / (1.0 - pdrop)
    ctx = (pr @ sp(v, Lk)).permute(0, 2, 1, 3).reshape(B, Lq, D)
    dense = ctx @ leaf["output.dense.weight"].t() + leaf["output.dense.bias"]
    if pdrop > 0:
        keep_h = ops.dropout_keep_mask(B * Lq, D, pdrop, seeds[1]).float().view(B, Lq, D)
        dense = dense * keep_h / (1.0 - pdrop)
    out = torch.nn.functional.layer_norm(dense + xf, (D,), leaf["output.LayerNorm.weight"], leaf["output.LayerNorm.bias"],
                                         so.LayerNorm.eps)
    out.backward(dy.float())
    return out.detach(), xf.grad, yf.grad, {n: t.grad for n, t in leaf.items()}


@pytest.mark.parametrize("pdrop", [0.0, 0.1])
@pytest.mark.parametrize("B,I,T", [(3, 577, 32), (128, 577, 32), (32, 145, 32), (3, 577, 64), (2, 1025, 32)])
@pytest.mark.parametrize("direction", ["txt<-img", "img<-txt"])
def test_fused_cross_attention_backward_against_fp32_autograd(direction, B, I, T, pdrop):
    """m3ae_xattn_fwd + m3ae_xattn_bwd against fp32 torch autograd of the reference's formulation -- at B = 3 (one reduction
    split), B = 32 (ksplit 4) and at B = 128, a batch the product takes this path at (ops.XATTN_TRAIN_MIN_BATCH = 96: the
    split-K fp32-atomic weight gradients run with ksplit >= 10 there), at 64 text tokens (pre-training) and at 1025 image tokens
    (configs[4]; image queries), with the -10000 key mask, with dropout through the exported masks.  The composition (GEMMs + flash attention) runs on the same seeds as a yardstick: the fused path must be within
    1.5 x its error against the fp32 reference or 2 % relative L2 (4 % under dropout), per tensor."""
    if direction == "txt<-img" and I > 640:
        pytest.skip("text queries over more than 640 image keys take the composition")
    att, store = make(seed=B + I, wscale=1.5)
    att.train(pdrop > 0)
    P = att.block_params()
    xt = torch.randn(B, T, D, device="cuda").to(torch.bfloat16)
    xi = torch.randn(B, I, D, device="cuda").to(torch.bfloat16)
    mt = torch.zeros(B, T, device="cuda")
    mt[:, T - 9:] = -10000.0
    mt[0, 5:] = -10000.0
    x, y, mask = (xt, xi, None) if direction == "txt<-img" else (xi, xt, mt)
    L, Lo = x.shape[1], y.shape[1]
    dy = torch.randn(B * L, D, device="cuda").to(torch.bfloat16)
    seed = 4321
    seeds = ((seed << 32) | 1, (seed << 32) | 2)     # ops.next_dropout_seed: attention probabilities, then hidden
    ref = _fp32_reference_with_gradients(att, x, y, mask, dy.view(B, L, D), pdrop, seeds)
    got = {}
    for fused in (True, False):
        store.zero_grad()
        ops.set_dropout_seed(seed)
        old = ops.XATTN
        ops.XATTN = "always" if fused else "off"
        try:
            out, saved = ops._attn_sub_fwd(x.view(B * L, D), B, L, y.view(B * Lo, D), Lo, mask, P, pdrop, fused_cross=True)
            assert isinstance(saved[0], str) is fused
            dx, dother = ops._attn_sub_bwd(dy, saved, B, L, Lo, P)
        finally:
            ops.XATTN = old
        got[fused] = (out.float().view(B, L, D), dx.float().view(B, L, D), dother.float().view(B, Lo, D),
                      {n: p_.grad.clone() for n, p_ in att.named_parameters()})
    rel = lambda a, b: ((a - b).double().norm() / (b.double().norm() + 1e-30)).item()
    tol = 0.04 if pdrop > 0 else 0.02
    names = ["out", "dx", "dother"]
    for idx, name in enumerate(names):
        ef, ec = rel(got[True][idx], ref[idx]), rel(got[False][idx], ref[idx])
        assert torch.isfinite(got[True][idx]).all()
        assert ef < max(1.5 * ec, tol), (name, ef, ec)
    for n, g_ref in ref[3].items():
        gf, gc = got[True][3][n], got[False][3][n]
        if n == "self.key.bias":     # exactly zero in exact arithmetic (b_k drops out of the softmax); the fused path never touches it
            scale = max(ref[3]["self.value.bias"].abs().max().item(), 1e-6)
            assert gf.abs().max().item() <= 0.05 * scale, n
            continue
        ef, ec = rel(gf, g_ref), rel(gc, g_ref)
        assert ef < max(1.5 * ec, tol), (n, ef, ec)
