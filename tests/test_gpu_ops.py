"""GPU: every kernel family through the C ABI against a plain PyTorch fp32 reference of the same op.
Tolerances: fp32 paths 1e-4 (fp32 FMA vs ATen); bf16 paths are compared with an fp32 reference computed from the
SAME bf16-rounded inputs, tolerance = bf16 output rounding (2^-8 relative) + accumulation noise."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from m3ae_amd import ops  # noqa: E402


def dev():
    return torch.device("cuda")


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + int(np.prod(shape)) % 1000)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def close(a, b, rtol, atol, msg=""):
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol)
    assert not bad.any(), f"{msg}: {int(bad.sum())}/{a.numel()} off, max err {err.max().item():.3e} (ref max {b.abs().max().item():.3e})"


def test_selftest_hardware_idioms():
    res = ops.selftest()
    assert res[0] == 0, f"mismatches [total, mfma16, mfma32, tr_read, acc_as_operand, lds_dma] = {res}"


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (200, 136, 128), (577 * 2, 768, 768), (33, 498, 1536), (1154, 2304, 768)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nt_bias_act_residual(M, N, K, dtype):
    x, w = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
    b, r = rnd(N, seed=3), rnd(M, N, dtype=dtype, seed=4)
    for act, tact in [(ops.ACT_NONE, lambda t: t), (ops.ACT_GELU, torch.nn.functional.gelu),
                      (ops.ACT_QUICKGELU, lambda t: t * torch.sigmoid(1.702 * t)), (ops.ACT_TANH, torch.tanh)]:
        y, pre = ops.mm_nt(x, K, M, w, bias=b, act=act, residual=r, want_preact=True)
        ref_pre = x.float() @ w.float().t() + b
        ref = tact(ref_pre) + r.float()
        tol = (1e-4, 1e-4) if dtype == torch.float32 else (1e-2, 2e-2)
        close(pre, ref_pre, *tol, msg=f"preact act={act} path={ops.last_gemm_path()}")
        close(y, ref, *tol, msg=f"y act={act} path={ops.last_gemm_path()}")
    if dtype == torch.bfloat16:
        expect = "mfma_nt" if (N % 4 == 0 and K % 64 == 0) else "generic"
        assert ops.last_gemm_path() == expect


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5])
def test_gemm_mfma_matches_generic_bitwise_shape_sweep(variant):
    """The MFMA NT kernel variants against the generic kernel on identical bf16 inputs, ragged M / N tails."""
    from m3ae_amd import _lib
    _lib.lib().m3ae_set_tuning(0, variant)
    for M, N, K in [(1, 128, 64), (127, 132, 192), (129, 260, 64), (300, 8, 3072), (1000, 3072, 768), (577, 768, 128),
                    (2308, 2304, 768)]:
        x, w = rnd(M, K, dtype=torch.bfloat16, seed=5), rnd(N, K, dtype=torch.bfloat16, scale=K ** -0.5, seed=6)
        y1, _ = ops.mm_nt(x, K, M, w, out_dtype=torch.float32)
        assert ops.last_gemm_path() == "mfma_nt"
        y2, _ = ops.mm_nt(x, K, M, w, out_dtype=torch.float32, force_generic=True)
        assert ops.last_gemm_path() == "generic"
        close(y1, y2, 1e-5, 1e-5, msg=f"{M}x{N}x{K}")
    _lib.lib().m3ae_set_tuning(0, -1)


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (577 * 2, 768, 768), (1000, 2304, 768), (4616, 768, 3072), (37, 128, 256),
                                   (4616, 256, 512), (9000, 768, 768)])
def test_gemm_wgrad_tn(M, N, K):
    """dW[N,K] += dY[M,N]^T X[M,K]: MFMA TN kernel (transposing LDS reads, split-K atomics), fp32 accumulate."""
    dy, x = rnd(M, N, dtype=torch.bfloat16, seed=7), rnd(M, K, dtype=torch.bfloat16, seed=8)
    g = torch.ones(N, K, dtype=torch.float32, device=dev())
    db = torch.full((N,), 2.0, dtype=torch.float32, device=dev())
    ops.gemm(dy, 1, N, x, K, 1, g, K, N, K, M, accumulate=True, a_rowsum=db)
    assert ops.last_gemm_path() == "mfma_tn"
    ref = 1.0 + dy.float().t() @ x.float()
    close(g, ref, 1e-4, 1e-3 * math.sqrt(M), msg="wgrad")
    close(db, 2.0 + dy.float().sum(0), 1e-4, 1e-3 * math.sqrt(M), msg="fused bias grad")
    g32 = torch.zeros(N, K, dtype=torch.float32, device=dev())
    db32 = torch.zeros(N, dtype=torch.float32, device=dev())
    ops.gemm(dy.float(), 1, N, x.float(), K, 1, g32, K, N, K, M, accumulate=True, a_rowsum=db32)
    assert ops.last_gemm_path() == "generic"
    close(g32, ref - 1.0, 1e-4, 1e-4 * math.sqrt(M), msg="wgrad fp32")
    close(db32, dy.float().sum(0), 1e-4, 1e-4 * math.sqrt(M), msg="fused bias grad fp32")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_dgrad_with_fused_dact(dtype):
    M, N, K = 300, 512, 128
    dy, w, u = rnd(M, N, dtype=dtype, seed=9), rnd(N, K, dtype=dtype, scale=N ** -0.5, seed=10), rnd(M, K, dtype=dtype, seed=11)
    wt = w.t().contiguous()

    class P:  # minimal parameter stand-in
        pass
    p = P()
    p.m3ae_c, p.m3ae_t = w, (wt if dtype == torch.bfloat16 else None)
    dx = ops.mm_dgrad(dy, p, dact_aux=u, dact=ops.ACT_GELU)
    uf = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uf).backward(dy.float() @ w.float())
    tol = (1e-4, 1e-4) if dtype == torch.float32 else (1e-2, 2e-2)
    close(dx, uf.grad, *tol, msg="dgrad*gelu'")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,eps,act", [(768, 1e-12, ops.ACT_NONE), (768, 1e-5, ops.ACT_NONE), (1536, 1e-5, ops.ACT_GELU), (128, 1e-12, ops.ACT_NONE)])
def test_layernorm_fwd_bwd(dtype, D, eps, act):
    M = 1154
    x = rnd(M, D, dtype=dtype, seed=12).requires_grad_(True)
    gamma = (1 + 0.1 * rnd(D, seed=13)).requires_grad_(True)
    beta = (0.1 * rnd(D, seed=14)).requires_grad_(True)
    gamma.grad = torch.zeros_like(gamma)
    beta.grad = torch.zeros_like(beta)
    y = ops.layer_norm(x, gamma, beta, eps, act=act)
    dy = rnd(M, D, dtype=dtype, seed=15)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    gr, br = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, eps)
    if act == ops.ACT_GELU:
        yr = torch.nn.functional.gelu(yr)
    yr.backward(dy.float())
    tol = (1e-4, 1e-4) if dtype == torch.float32 else (1e-2, 2e-2)
    close(y, yr, *tol, msg="ln y")
    close(x.grad, xr.grad, *tol, msg="ln dx")
    gt = (1e-3, 1e-3 * math.sqrt(M)) if dtype == torch.float32 else (2e-2, 2e-2 * math.sqrt(M))
    close(gamma.grad, gr.grad, *gt, msg="ln dgamma")
    close(beta.grad, br.grad, *gt, msg="ln dbeta")


def _attn_ref(q, k, v, H, mask):
    B, Lq, D = q.shape
    Lk = k.shape[1]
    dh = D // H
    qh = q.view(B, Lq, H, dh).permute(0, 2, 1, 3)
    kh = k.view(B, Lk, H, dh).permute(0, 2, 1, 3)
    vh = v.view(B, Lk, H, dh).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if mask is not None:
        s = s + mask[:, None, None, :]
    p = torch.softmax(s, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, Lq, D)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Lq,Lk,masked", [(32, 32, True), (577, 577, False), (32, 577, False), (577, 32, True), (17, 17, False), (100, 45, True)])
def test_attention_fwd_bwd(dtype, Lq, Lk, masked):
    B, H, dh = 2, 3, 64
    D = H * dh
    cross = Lq != Lk
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, device=dev())
        mask[1, Lk // 2:] = -10000.0
        mask[0, Lk - 3:] = -10000.0
    if cross:
        q = rnd(B, Lq, D, dtype=dtype, seed=16).requires_grad_(True)
        kv = rnd(B, Lk, 2 * D, dtype=dtype, seed=17).requires_grad_(True)
        o = ops.cross_attention(q, kv, mask, H)
        qr, kvr = q.detach().float().requires_grad_(True), kv.detach().float().requires_grad_(True)
        oref = _attn_ref(qr, kvr[..., :D], kvr[..., D:], H, mask)
        leaves, rleaves = (q, kv), (qr, kvr)
    else:
        qkv = rnd(B, Lq, 3 * D, dtype=dtype, seed=18).requires_grad_(True)
        o = ops.self_attention(qkv, mask, H)
        qkvr = qkv.detach().float().requires_grad_(True)
        oref = _attn_ref(qkvr[..., :D], qkvr[..., D:2 * D], qkvr[..., 2 * D:], H, mask)
        leaves, rleaves = (qkv,), (qkvr,)
    do = rnd(B, Lq, D, dtype=dtype, seed=19)
    o.backward(do)
    oref.backward(do.float())
    tol = (1e-4, 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)
    close(o, oref, *tol, msg="attn o")
    for a, r in zip(leaves, rleaves):
        close(a.grad, r.grad, tol[0], tol[1] * 2, msg="attn grad")


def test_attention_fully_masked_row_is_finite_and_matches():
    """All keys of one sample carry the additive -10000 (not -inf): softmax is shift invariant, result is finite."""
    B, H, L, D = 2, 2, 32, 128
    qkv = rnd(B, L, 3 * D, dtype=torch.bfloat16, seed=20)
    mask = torch.zeros(B, L, device=dev())
    mask[1, :] = -10000.0
    o = ops.self_attention(qkv, mask, H)
    f = qkv.float()
    ref = _attn_ref(f[..., :D], f[..., D:2 * D], f[..., 2 * D:], H, mask)
    assert torch.isfinite(o.float()).all()
    close(o, ref, 2e-2, 2e-2, msg="fully masked")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_roberta_embed_and_vit_tokens(dtype):
    B, S, D, V = 3, 32, 128, 500
    ids = torch.randint(3, V, (B, S), generator=torch.Generator().manual_seed(0))
    ids[1, 20:] = 1
    ids[2, 5:] = 1
    ids = ids.to(dev())
    word, pos, typ = (rnd(V, D, seed=21).requires_grad_(True), rnd(514, D, seed=22).requires_grad_(True),
                      rnd(1, D, seed=23).requires_grad_(True))
    out = ops.roberta_embed(ids, word, pos, typ, 1, dtype)
    ne = (ids != 1).long()
    pid = torch.cumsum(ne, 1) * ne + 1
    ref = word[ids] + typ[0] + pos[pid]
    tol = (1e-6, 1e-6) if dtype == torch.float32 else (1e-2, 1e-2)
    close(out, ref, *tol, msg="embed")
    do = rnd(B, S, D, dtype=dtype, seed=24)
    out.backward(do)
    gw, gp, gt = torch.autograd.grad(ref, (word, pos, typ), do.float())
    close(word.grad, gw, 1e-4, 1e-4, msg="dword")
    close(pos.grad, gp, 1e-4, 1e-4, msg="dpos")
    close(typ.grad, gt, 1e-4, 1e-3, msg="dtype")
    # ViT tokens
    R, P, W = 64, 16, 128
    img = rnd(B, 3, R, R, seed=25)
    conv = rnd(W, 3, P, P, scale=0.02, seed=26).requires_grad_(True)
    cls, pe = rnd(W, seed=27).requires_grad_(True), rnd(17, W, seed=28).requires_grad_(True)
    if dtype == torch.bfloat16:
        conv.m3ae_c = conv.detach().to(torch.bfloat16)
    tok = ops.vit_tokens(img, conv, cls, pe, dtype)
    x = torch.nn.functional.conv2d(img, conv, stride=P).reshape(B, W, -1).permute(0, 2, 1)
    ref = torch.cat([cls.view(1, 1, W).expand(B, 1, W), x], 1) + pe
    tol = (1e-4, 1e-4) if dtype == torch.float32 else (2e-2, 2e-2)
    close(tok, ref, *tol, msg="vit tokens")
    dt = rnd(B, 17, W, dtype=dtype, seed=29)
    tok.backward(dt)
    gc, gcl, gpe = torch.autograd.grad(ref, (conv, cls, pe), dt.float())
    close(conv.grad, gc, tol[0], tol[1] * 4, msg="dconv")
    close(cls.grad, gcl, tol[0], tol[1], msg="dcls")
    close(pe.grad, gpe, tol[0], tol[1], msg="dpos")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_losses(dtype):
    B, Cc = 6, 498
    x = rnd(B, Cc, dtype=dtype, seed=30).requires_grad_(True)
    z = torch.zeros(B, Cc, device=dev())
    z[torch.arange(B), torch.arange(B) * 7] = 1.0
    loss = ops.bce_with_logits_loss(x, z)
    loss.backward()
    xr = x.detach().float().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, z) * Cc
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-4 * ref.item()
    close(x.grad, xr.grad, 1e-2, 1e-5, msg="bce grad")
    V = 1000
    lg = rnd(B, 8, V, dtype=dtype, seed=31).requires_grad_(True)
    lab = torch.randint(0, V, (B, 8), generator=torch.Generator().manual_seed(1))
    lab[0, :5] = -100
    lab = lab.to(dev())
    l2 = ops.cross_entropy(lg, lab)
    l2.backward()
    lr_ = lg.detach().float().requires_grad_(True)
    r2 = torch.nn.functional.cross_entropy(lr_.view(-1, V), lab.view(-1), ignore_index=-100)
    r2.backward()
    assert abs(l2.item() - r2.item()) < 1e-4 * r2.item()
    close(lg.grad, lr_.grad, 1e-2, 1e-6, msg="xent grad")


def test_adamw_matches_hf_semantics():
    import ctypes as C
    from m3ae_amd import _lib
    from oracle import m3ae_oracle as O
    n = 4096
    p, g = rnd(n, seed=32), rnd(n, seed=33)
    m, v = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    sh = torch.zeros(n, dtype=torch.bfloat16, device=dev())
    pr, mr, vr = p.cpu().clone(), torch.zeros(n), torch.zeros(n)
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for step in (1, 2, 3):
        _lib.check(L.m3ae_adamw(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                C.c_void_p(v.data_ptr()), C.c_void_p(sh.data_ptr()), n, 1e-3, 0.9, 0.98, 1e-8, 0.01,
                                step, 0.5, s), "adamw")
        O.adamw_step(pr, g.cpu() * 0.5, mr, vr, step, 1e-3, 0.01)
    close(p, pr, 1e-5, 1e-6, msg="adamw p")
    close(sh, pr, 1e-2, 1e-3, msg="adamw shadow")
