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


def test_build_then_first_launch_in_a_fresh_process():
    """`__graft_entry__.build()` (which loads the library) followed by the first launch in the SAME fresh process: the library must
    not pull the system HIP runtime in before PyTorch-ROCm's own (one runtime per process; the wrong order ended in
    hipErrorNoDevice on the first launch).  One child process, nothing else on the GPU meanwhile."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import __graft_entry__ as g; g.build(); from m3ae_amd import ops; r = ops.selftest(); "
            "assert r[0] == 0, r; print('child ok')")
    p = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "child ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


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


@pytest.mark.parametrize("variant", [0, 4, 7, 9])
def test_gemm_mfma_matches_generic_bitwise_shape_sweep(variant):
    """The MFMA NT kernel variants against the generic kernel on identical bf16 inputs, ragged M / N tails."""
    from m3ae_amd import _lib
    ops.GEMM_NT_VARIANT = variant
    for M, N, K in [(1, 128, 64), (127, 132, 192), (129, 260, 64), (300, 8, 3072), (1000, 3072, 768), (577, 768, 128),
                    (2308, 2304, 768)]:
        x, w = rnd(M, K, dtype=torch.bfloat16, seed=5), rnd(N, K, dtype=torch.bfloat16, scale=K ** -0.5, seed=6)
        y1, _ = ops.mm_nt(x, K, M, w, out_dtype=torch.float32)
        assert ops.last_gemm_path() in ("mfma_nt", "mfma_nt_pp", "mfma_nt_pp2")
        y2, _ = ops.mm_nt(x, K, M, w, out_dtype=torch.float32, force_generic=True)
        assert ops.last_gemm_path() == "generic"
        close(y1, y2, 1e-5, 1e-5, msg=f"{M}x{N}x{K}")
    ops.GEMM_NT_VARIANT = -1


@pytest.mark.parametrize("variant", [7, 8, 9, 10, -1])
@pytest.mark.parametrize("kind", ["plain", "bias+res", "gelu+deriv", "dmul", "dropout+res", "tanh+pre+res", "dgelu+res"])
def test_gemm_nt_pingpong_and_persistent_vs_fp32_reference_large_ragged(variant, kind):
    """The step's dominant kernel -- gemm_nt_pp2_kernel (round 4: 9, its persistent launch 10; what the auto rule (-1) picks for
    large shapes) and its predecessors gemm_nt_pp_kernel (7) / persistent (8) -- against an INDEPENDENT fp32 torch reference on the same bf16-rounded operands, every epilogue class,
    at a ragged shape with >= 512 tiles (36965 x 3072 x 768: 145 x 12 = 1740 tiles, last row tile 101 rows)."""
    from m3ae_amd import _lib
    L = _lib.lib()
    M, N, K = 36965, 3072, 768
    x, w = rnd(M, K, dtype=torch.bfloat16, seed=31), rnd(N, K, dtype=torch.bfloat16, scale=K ** -0.5, seed=32)
    b = rnd(N, seed=33)
    aux = rnd(M, N, dtype=torch.bfloat16, seed=34)
    pre = x.float() @ w.float().t()
    ops.GEMM_NT_VARIANT = variant
    try:
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        extra = None
        if kind == "plain":
            ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K)
            ref = pre
        elif kind == "bias+res":
            ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K, bias=b, residual=aux)
            ref = pre + b + aux.float()
        elif kind == "gelu+deriv":
            extra = torch.empty_like(y)
            ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K, bias=b, act=ops.ACT_GELU, preact=extra, preact_grad=True)
            u = (pre + b).double()
            ref = torch.nn.functional.gelu(u).float()
            cdf = 0.5 * (1 + torch.erf(u / math.sqrt(2.0)))
            dref = (cdf + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)).float()
            close(extra, dref, 1e-2, 1e-2, msg="gelu derivative")
        elif kind == "dmul":
            ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K, dact_aux=aux, dact=ops.ACT_MULAUX)
            ref = pre * aux.float()
        elif kind == "tanh+pre+res":   # the catch-all epilogue class (EPI_ANY: the instantiation that keeps compiler-visible loads)
            extra = torch.empty_like(y)
            ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K, bias=b, act=ops.ACT_TANH, preact=extra, residual=aux)
            ref = torch.tanh(pre + b) + aux.float()
            close(extra, pre + b, 1e-2, 2e-2, msg="pre-activation")
        elif kind == "dgelu+res":      # derivative operand prefetched, residual read inside the row passes
            aux2 = rnd(M, N, dtype=torch.bfloat16, seed=35)
            ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K, dact_aux=aux, dact=ops.ACT_GELU, residual=aux2)
            u = aux.double()
            cdf = 0.5 * (1 + torch.erf(u / math.sqrt(2.0)))
            ref = (pre + aux2.float()) * (cdf + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)).float()   # m3ae_hip.h: residual first
        else:
            keep = ops.dropout_keep_mask(M, N, 0.1, 4242)   # the exported mask of (p, seed) on an [M, N] array
            ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K, bias=b, residual=aux, dropout=(0.1, 4242))
            ref = (pre + b) * keep.float() / 0.9 + aux.float()
            assert 0.88 < keep.float().mean().item() < 0.92
        assert ops.last_gemm_path() == ("mfma_nt_pp" if variant in (7, 8) else "mfma_nt_pp2")
        close(y, ref, 1e-2, 2e-2, msg=f"variant {variant} {kind}")
    finally:
        ops.GEMM_NT_VARIANT = -1


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
    # second copy of the state: the same three steps with the step's {lr, step size} read from DEVICE memory (ABI 3 hyper_dev:
    # what a hipGraph-captured step uses) while the host arguments hold stale values
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    hyper = torch.zeros(2, device=dev())
    for step in (1, 2, 3):
        _lib.check(L.m3ae_adamw(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                C.c_void_p(v.data_ptr()), C.c_void_p(sh.data_ptr()), n, 1e-3, 0.9, 0.98, 1e-8, 0.01,
                                step, 0.5, None, s), "adamw")
        hyper.copy_(torch.tensor([1e-3, 1e-3 * math.sqrt(1 - 0.98 ** step) / (1 - 0.9 ** step)]))
        _lib.check(L.m3ae_adamw(C.c_void_p(p2.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m2.data_ptr()),
                                C.c_void_p(v2.data_ptr()), None, n, 123.0, 0.9, 0.98, 1e-8, 0.01,
                                77, 0.5, C.c_void_p(hyper.data_ptr()), s), "adamw hyper_dev")
        O.adamw_step(pr, g.cpu() * 0.5, mr, vr, step, 1e-3, 0.01)
    close(p, pr, 1e-5, 1e-6, msg="adamw p")
    close(sh, pr, 1e-2, 1e-3, msg="adamw shadow")
    close(p2, p, 1e-6, 1e-7, msg="adamw with device-side hyper-parameters")


# ----------------------------------------------------------------------------------------------------------
# dropout (training mode; bert_model.py:334/:362/:440 and the RoBERTa embeddings).  The library's mask is a
# counter hash of (seed, linear element index); the mask is exported and the op is checked against torch with
# that exact mask, so these are deterministic parity tests, not statistical ones.
# ----------------------------------------------------------------------------------------------------------
def test_dropout_mask_statistics_and_determinism():
    p = 0.1
    for rows, cols in ((1024, 1024), (1733, 577)):  # 577: odd row length (the index stride is padded to 580)
        n = rows * cols
        m1 = ops.dropout_keep_mask(rows, cols, p, 1234)
        m2 = ops.dropout_keep_mask(rows, cols, p, 1234)
        m3 = ops.dropout_keep_mask(rows, cols, p, 1235)
        assert torch.equal(m1, m2)
        assert (m1 != m3).float().mean().item() > 0.1
        keep = m1.float().mean().item()
        assert abs(keep - (1 - p)) < 5 * math.sqrt(p * (1 - p) / n), keep
        # four neighbours share one hash (a different byte each): keep bits must still be uncorrelated at lags
        # 1..4 along a row and between rows
        f = m1.float() - keep
        for lag in (1, 2, 3, 4):
            assert abs((f[:, lag:] * f[:, :-lag]).mean().item() / (p * (1 - p))) < 0.01, lag
        assert abs((f[1:] * f[:-1]).mean().item() / (p * (1 - p))) < 0.01
    rows, cols = 1024, 1024
    for dtype in (torch.float32, torch.bfloat16):
        x = rnd(rows, cols, dtype=dtype, seed=40).requires_grad_(True)
        ops.set_dropout_seed(7)
        y = ops.dropout(x, p)
        seed = ((7 << 32) | 1)
        mk = ops.dropout_keep_mask(rows, cols, p, seed).float()
        close(y, x.detach().float() * mk / (1 - p), 1e-2 if dtype == torch.bfloat16 else 1e-6, 1e-6, msg="dropout fwd")
        y.backward(torch.ones_like(y))
        close(x.grad, mk / (1 - p), 1e-2, 1e-6, msg="dropout bwd")
    assert ops.dropout(x, p, training=False) is x and ops.dropout(x, 0.0) is x


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1154, 768, 768), (200, 136, 128), (4616, 768, 3072)])
def test_gemm_epilogue_dropout_matches_exported_mask(dtype, M, N, K):
    p, seed = 0.1, 0xABCDEF0123
    x, w = rnd(M, K, dtype=dtype, seed=41), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=42)
    b, r = rnd(N, seed=43), rnd(M, N, dtype=dtype, seed=44)
    y, _ = ops.mm_nt(x, K, M, w, bias=b, residual=r, dropout=(p, seed))
    mk = ops.dropout_keep_mask(M, N, p, seed).float()
    ref = (x.float() @ w.float().t() + b) * mk / (1 - p) + r.float()
    tol = (1e-4, 1e-4) if dtype == torch.float32 else (1e-2, 2e-2)
    close(y, ref, *tol, msg=f"gemm dropout path={ops.last_gemm_path()}")
    # dropped elements are EXACTLY the residual
    dropped = mk == 0
    assert torch.equal(y[dropped], r[dropped])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_bwd_drop_second_output(dtype):
    M, D, p, seed = 1154, 768, 0.1, 99
    x = rnd(M, D, dtype=dtype, seed=45)
    ln = torch.nn.LayerNorm(D, eps=1e-12).to(dev())
    ln.weight.data = 1 + 0.1 * rnd(D, seed=46)
    ln.bias.data = 0.1 * rnd(D, seed=47)
    for q in (ln.weight, ln.bias):
        q.grad = torch.zeros_like(q)
    _, mean, rstd = ops.ln_fwd_raw(x, ln)
    dy = rnd(M, D, dtype=dtype, seed=48)
    dx0 = ops.ln_bwd_raw(dy, x, ln, mean, rstd)
    g0, b0 = ln.weight.grad.clone(), ln.bias.grad.clone()
    ln.weight.grad.zero_(), ln.bias.grad.zero_()
    dx, dxd = ops.ln_bwd_raw(dy, x, ln, mean, rstd, drop=(p, seed))
    mk = ops.dropout_keep_mask(M, D, p, seed).float()
    assert torch.equal(dx, dx0)
    close(dxd, dx0.float() * mk / (1 - p), 1e-2 if dtype == torch.bfloat16 else 1e-6, 1e-7, msg="dx_drop")
    close(ln.weight.grad, g0, 1e-5, 1e-4, msg="dgamma")
    close(ln.bias.grad, b0, 1e-5, 1e-4, msg="dbeta")


def _attn_ref_drop(q, k, v, H, mask, keep, p):
    B, Lq, D = q.shape
    Lk = k.shape[1]
    dh = D // H
    qh = q.view(B, Lq, H, dh).permute(0, 2, 1, 3)
    kh = k.view(B, Lk, H, dh).permute(0, 2, 1, 3)
    vh = v.view(B, Lk, H, dh).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh)
    if mask is not None:
        s = s + mask[:, None, None, :]
    pr = torch.softmax(s, dim=-1) * keep / (1 - p)  # nn.Dropout on attention_probs (bert_model.py:334)
    return (pr @ vh).permute(0, 2, 1, 3).reshape(B, Lq, D)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Lq,Lk,masked", [(32, 32, True), (577, 577, False), (32, 577, False), (577, 32, True), (100, 45, True)])
def test_attention_dropout_fwd_bwd_matches_exported_mask(dtype, Lq, Lk, masked):
    B, H, dh, p, seed = 2, 3, 64, 0.1, 0x1234567
    D = H * dh
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, device=dev())
        mask[1, Lk // 2:] = -10000.0
    keep = ops.dropout_keep_mask(B * H * Lq, Lk, p, seed).float().view(B, H, Lq, Lk)
    do = rnd(B, Lq, D, dtype=dtype, seed=52)
    if Lq != Lk:
        q = rnd(B, Lq, D, dtype=dtype, seed=50).requires_grad_(True)
        kv = rnd(B, Lk, 2 * D, dtype=dtype, seed=51).requires_grad_(True)
        o = ops.cross_attention(q, kv, mask, H, (p, seed))
        qr, kvr = q.detach().float().requires_grad_(True), kv.detach().float().requires_grad_(True)
        oref = _attn_ref_drop(qr, kvr[..., :D], kvr[..., D:], H, mask, keep, p)
        leaves, rleaves = (q, kv), (qr, kvr)
    else:
        qkv = rnd(B, Lq, 3 * D, dtype=dtype, seed=53).requires_grad_(True)
        o = ops.self_attention(qkv, mask, H, (p, seed))
        qkvr = qkv.detach().float().requires_grad_(True)
        oref = _attn_ref_drop(qkvr[..., :D], qkvr[..., D:2 * D], qkvr[..., 2 * D:], H, mask, keep, p)
        leaves, rleaves = (qkv,), (qkvr,)
    o.backward(do)
    oref.backward(do.float())
    tol = (1e-4, 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)
    close(o, oref, *tol, msg="attn dropout o")
    for a, r in zip(leaves, rleaves):
        close(a.grad, r.grad, tol[0], tol[1] * 2, msg="attn dropout grad")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["cross", "self"])
def test_fused_block_with_dropout_matches_op_level_composition(dtype, kind):
    """Training-mode BertCrossLayer / RobertaLayer: the fused node (epilogue dropout, LN-backward second output,
    in-kernel attention dropout) against the op-level composition with standalone dropout kernels on the same seeds."""
    from types import SimpleNamespace
    from m3ae_amd import synth
    from m3ae_amd.modules.bert_model import BertCrossLayer, BertSelfLayer
    from m3ae_amd.param_store import ParamStore
    B, L, Lo, D, H = 2, 40, 101, 128, 2
    layer = (BertCrossLayer(D, H, 4 * D, drop_rate=0.1) if kind == "cross" else BertSelfLayer(D, H, 4 * D, drop_rate=0.1))
    synth.fill_deterministic(layer)
    cfg = dict(learning_rate=1e-3, weight_decay=0.01, lr_multiplier_head=1, lr_multiplier_multi_modal=1)
    store = ParamStore(layer, cfg, "cuda", dtype, weight_units=layer.weight_units)
    layer.train()
    h0 = rnd(B, L, D, dtype=dtype, seed=60)
    e0 = rnd(B, Lo, D, dtype=dtype, seed=61)
    mask = torch.zeros(B, L, device=dev())
    mask[1, L - 7:] = -10000.0
    dy = rnd(B, L, D, dtype=dtype, seed=62)
    res = []
    for fused in (True, False):
        store.zero_grad()
        ops.set_dropout_seed(321)
        h, e = h0.clone().requires_grad_(True), e0.clone().requires_grad_(True)
        if kind == "cross":
            y = (layer if fused else layer.forward_unfused)(h, e, mask, None)
        else:
            y = (layer if fused else layer.forward_unfused)(h, mask)
        y.backward(dy)
        res.append(SimpleNamespace(y=y.detach().clone(), dh=h.grad.clone(), de=None if e.grad is None else e.grad.clone(),
                                   g={n: q.grad.clone() for n, q in layer.named_parameters()}))
    f, u = res
    tol = (1e-4, 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)
    close(f.y, u.y, *tol, msg="y")
    close(f.dh, u.dh, tol[0], tol[1] * 2, msg="dh")
    if kind == "cross":
        close(f.de, u.de, tol[0], tol[1] * 2, msg="d encoder states")
    for n in f.g:
        if n.endswith("key.bias"):  # identically zero in exact arithmetic (softmax shift invariance): pure rounding noise
            continue
        scale = u.g[n].abs().max().item() + 1e-12
        close(f.g[n], u.g[n], tol[0] * 2, tol[1] * scale * 2, msg=n)
    # dropout really happened: a different seed gives a different output, eval() gives the deterministic one
    ops.set_dropout_seed(322)
    y2 = layer(h0, e0, mask, None) if kind == "cross" else layer(h0, mask)
    assert (y2.float() - f.y.float()).abs().max().item() > 0.05
    layer.eval()
    y3 = layer(h0, e0, mask, None) if kind == "cross" else layer(h0, mask)
    y4 = layer(h0, e0, mask, None) if kind == "cross" else layer(h0, mask)
    assert torch.equal(y3, y4)


@pytest.mark.parametrize("M,N,K", [(4616, 256, 512), (9000, 768, 768), (36928, 768, 3072), (5000, 1536, 256), (4100, 512, 768)])
def test_gemm_wgrad_tn_pingpong_variant(M, N, K):
    """TN variant 5 (gemm_tn_pp_kernel: 256x256 tile, staggered wave rows) incl. ragged reduction tails and the fused
    bias gradient; repeated to screen the ring for races (every run must agree with the fp32 reference)."""
    from m3ae_amd import _lib
    ops.GEMM_TN_VARIANT = 5
    try:
        dy, x = rnd(M, N, dtype=torch.bfloat16, seed=70), rnd(M, K, dtype=torch.bfloat16, seed=71)
        ref = dy.float().t() @ x.float()
        refb = dy.float().sum(0)
        for it in range(6):
            g = torch.zeros(N, K, dtype=torch.float32, device=dev())
            db = torch.zeros(N, dtype=torch.float32, device=dev())
            ops.gemm(dy, 1, N, x, K, 1, g, K, N, K, M, accumulate=True, a_rowsum=db)
            assert ops.last_gemm_path() == "mfma_tn"
            close(g, ref, 1e-4, 1e-3 * math.sqrt(M), msg=f"wgrad pp iter {it}")
            close(db, refb, 1e-4, 1e-3 * math.sqrt(M), msg=f"bias grad pp iter {it}")
    finally:
        ops.GEMM_TN_VARIANT = -1


def test_vocab_projection_padded_mfma_path_for_odd_vocabulary():
    """MLM head with a vocabulary that is no multiple of 128 (RoBERTa: 50265): zero-padded operand copies keep forward,
    dgrad and wgrad on the MFMA kernels; logits view + strided cross-entropy + gradients against torch."""
    from m3ae_amd import synth
    from m3ae_amd.modules.prediction_heads import MLMHead
    from m3ae_amd.param_store import ParamStore
    V, D, B, S = 1001, 128, 3, 50
    head = MLMHead(D, V)
    synth.fill_deterministic(head)
    head.bias.data = 0.1 * torch.randn(V)
    cfg = dict(learning_rate=1e-3, weight_decay=0.01, lr_multiplier_head=1, lr_multiplier_multi_modal=1)
    store = ParamStore(head, cfg, "cuda", torch.bfloat16, weight_units=head.weight_units)
    store.zero_grad()
    x = rnd(B, S, D, dtype=torch.bfloat16, seed=90).requires_grad_(True)
    labels = torch.randint(0, V, (B, S), device=dev())
    labels[0, :7] = -100
    logits = ops.vocab_linear(x, head.decoder.weight, head.bias)
    assert logits.shape == (B, S, V) and logits.stride(-2) == 1024 and ops.last_gemm_path().startswith("mfma")
    loss = ops.cross_entropy(logits, labels)
    loss.backward()
    w = head.decoder.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    bb = head.bias.detach().clone().requires_grad_(True)
    xr = x.detach().float().requires_grad_(True)
    lr = xr @ w.t() + bb
    lossr = torch.nn.functional.cross_entropy(lr.view(-1, V), labels.view(-1), ignore_index=-100)
    lossr.backward()
    close(logits, lr, 1e-2, 2e-2, msg="vocab logits")
    assert abs(loss.item() - lossr.item()) < 2e-3 * lossr.item()
    close(x.grad, xr.grad, 3e-2, 2e-2 * xr.grad.abs().max().item(), msg="dx")
    close(head.decoder.weight.grad, w.grad, 3e-2, 2e-2 * w.grad.abs().max().item(), msg="dW")
    close(head.bias.grad, bb.grad, 3e-2, 2e-2 * bb.grad.abs().max().item(), msg="db")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1154, 3072, 768), (200, 136, 128), (5000, 1024, 256)])
def test_gemm_saved_activation_derivative_scheme(dtype, M, N, K):
    """Forward epilogue with preact_grad: second output = act'(pre-activation); backward epilogue with
    dact=ACT_MULAUX: dX-GEMM output times that saved derivative.  Against torch autograd, all three activations."""
    x, w = rnd(M, K, dtype=dtype, seed=95), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=96)
    b = rnd(N, seed=97)
    dy = rnd(M, 64, dtype=dtype, seed=98)
    w2 = rnd(N, 64, dtype=dtype, scale=0.1, seed=99)   # dU = (dY . W2^T) * act'(U)
    tol = (1e-4, 1e-4) if dtype == torch.float32 else (1e-2, 2e-2)
    for act, tact in [(ops.ACT_GELU, torch.nn.functional.gelu), (ops.ACT_QUICKGELU, lambda t: t * torch.sigmoid(1.702 * t)),
                      (ops.ACT_RELU, torch.relu)]:
        y, dsaved = ops.mm_nt(x, K, M, w, bias=b, act=act, want_preact=True, preact_grad=True)
        u = (x.float() @ w.float().t() + b).requires_grad_(True)
        yr = tact(u)
        (dref,) = torch.autograd.grad(yr.sum(), u)
        close(y, yr, *tol, msg=f"y act={act}")
        if act == ops.ACT_RELU:   # derivative at exactly 0 is a convention; compare away from the kink
            far = u.detach().abs() > 1e-3
            close(dsaved[far], dref[far], *tol, msg="relu'")
        else:
            close(dsaved, dref, *tol, msg=f"act' act={act}")
        du, _ = ops.mm_nt(dy, 64, M, w2, dact_aux=dsaved, dact=ops.ACT_MULAUX)
        close(du, (dy.float() @ w2.float().t()) * dsaved.float(), *tol, msg=f"mulaux act={act}")


@pytest.mark.parametrize("causal", [False, True])
def test_attention_dropout_with_position_bias_and_causal(causal):
    """T5 attention in train mode: relative-position bias (+ causal in the decoder) with attention-weight dropout,
    forward / backward incl. the bias gradient, against torch with the library's exported mask."""
    B, H, L, dh, p, seed = 2, 4, 70, 64, 0.1, 0xBEEF01
    D = H * dh
    dt = torch.bfloat16
    q, k, v = (rnd(B, L, D, dtype=dt, seed=s_).requires_grad_(True) for s_ in (101, 102, 103))
    bias = (0.5 * rnd(H, L, L, seed=104)).contiguous()
    keep = ops.dropout_keep_mask(B * H * L, L, p, seed).float().view(B, H, L, L)
    o, lse = ops.attn_forward(q, k, v, H, None, bias, scale=1.0, causal=causal, dropout=(p, seed))
    do = rnd(B, L, D, dtype=dt, seed=105)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    dbias = torch.zeros_like(bias)
    ops.attn_backward(q, k, v, o, lse, do, dq, dk, dv, H, None, bias, scale=1.0, causal=causal, d_pos_bias=dbias,
                      dropout=(p, seed))
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    br = bias.clone().requires_grad_(True)
    sh = lambda t: t.view(B, L, H, dh).permute(0, 2, 1, 3)
    s = sh(qr) @ sh(kr).transpose(-1, -2) + br[None]
    if causal:
        s = s.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool, device=dev())), float("-inf"))
    pr = torch.softmax(s, dim=-1) * keep / (1 - p)
    oref = (pr @ sh(vr)).permute(0, 2, 1, 3).reshape(B, L, D)
    oref.backward(do.float())
    close(o, oref, 2e-2, 2e-2, msg="o")
    for a, r, n in ((dq, qr.grad, "dq"), (dk, kr.grad, "dk"), (dv, vr.grad, "dv")):
        close(a, r, 2e-2, 2e-2 * r.abs().max().item(), msg=n)   # unscaled scores (T5): gradients of magnitude ~20
    close(dbias, br.grad, 3e-2, 3e-2 * br.grad.abs().max().item(), msg="d_pos_bias")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogue_dropout_after_activation_and_in_dgrad(dtype):
    """T5DenseReluDense in train mode: wi GEMM with ReLU + saved derivative + dropout AFTER the activation; the matching
    dgrad GEMM multiplies by the saved derivative and re-applies the same mask."""
    M, N, K, p, seed = 1200, 512, 256, 0.1, 0x77AA
    x, w = rnd(M, K, dtype=dtype, seed=110), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=111)
    y, d = ops.mm_nt(x, K, M, w, act=ops.ACT_RELU, want_preact=True, preact_grad=True, dropout=(p, seed))
    mk = ops.dropout_keep_mask(M, N, p, seed).float()
    u = x.float() @ w.float().t()
    tol = (1e-4, 1e-4) if dtype == torch.float32 else (1e-2, 2e-2)
    close(y, torch.relu(u) * mk / (1 - p), *tol, msg="relu + dropout")
    far = u.abs() > 1e-3
    close(d[far], (u > 0).float()[far], *tol, msg="saved derivative is NOT dropped")
    dy = rnd(M, 64, dtype=dtype, seed=112)
    w2 = rnd(N, 64, dtype=dtype, scale=0.1, seed=113)
    du, _ = ops.mm_nt(dy, 64, M, w2, dact_aux=d, dact=ops.ACT_MULAUX, dropout=(p, seed))
    close(du, (dy.float() @ w2.float().t()) * d.float() * mk / (1 - p), *tol, msg="dgrad: derivative x mask")



@pytest.mark.parametrize("kind", ["plain", "bias+res", "gelu+deriv", "dmul"])
def test_gemm_nt_kernels_agree_bit_for_bit_on_the_large_shapes(kind):
    """Every NT kernel the auto path or a tuning key can select for large shapes -- 256x256 2-stage (4), ping-pong (7), its
    persistent form (8, >= 512 tiles), the second-generation kernel of round 4 (9, persistent launch 10, auto -1) -- accumulates in the same order: on the
    path's own large shapes and epilogue classes the bf16 outputs are identical, ragged row tail included."""
    from m3ae_amd import _lib
    L = _lib.lib()
    M, N, K = 147712 // 4 + 37, 3072, 768          # 145 row tiles x 12 column tiles (> 512), last row tile ragged
    x = rnd(M, K, dtype=torch.bfloat16, seed=21)
    w = rnd(N, K, dtype=torch.bfloat16, scale=K ** -0.5, seed=22)
    b = rnd(N, dtype=torch.float32, seed=23)
    aux = rnd(M, N, dtype=torch.bfloat16, seed=24)
    outs = {}
    try:
        for v in (4, 7, 8, 9, 10, -1):
            ops.GEMM_NT_VARIANT = v
            y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            extra = None
            if kind == "plain":
                ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K)
            elif kind == "bias+res":
                ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K, bias=b, residual=aux)
            elif kind == "gelu+deriv":
                extra = torch.empty_like(y)
                ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K, bias=b, act=ops.ACT_GELU, preact=extra, preact_grad=True)
            else:
                ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K, dact_aux=aux, dact=ops.ACT_MULAUX)
            outs[v] = (y, extra, ops.last_gemm_path())
        assert outs[8][2] == "mfma_nt_pp" and outs[9][2] == outs[10][2] == outs[-1][2] == "mfma_nt_pp2"
        for v in (7, 8, 9, 10, -1):
            assert torch.equal(outs[v][0].view(torch.int16), outs[4][0].view(torch.int16)), (kind, v)
            if outs[4][1] is not None:
                assert torch.equal(outs[v][1].view(torch.int16), outs[4][1].view(torch.int16)), (kind, v, "derivative")
    finally:
        ops.GEMM_NT_VARIANT = -1


def test_mim_bookkeeping_kernels_match_the_reference_formulas():
    """random_masking's index work (m3ae_module.py:153-183), the MIM target (patchify :185-192 +
    objectives.py:52-56) and the masked MSE with its gradient (objectives.py:58-62) against the reference's own torch
    expressions, evaluated on the CPU in fp32.  Index outputs exact; ties in the noise included."""
    g = torch.Generator().manual_seed(3)
    B, L, keep = 5, 36, 9
    noise = torch.rand(B, L, generator=g)
    noise[1, 7] = noise[1, 3]                          # a tie: ranks follow the index order (stable argsort)
    ids_restore, keep_rows, mask = ops.mask_ranks(noise.to(dev()), keep)
    shuf = torch.argsort(noise, dim=1, stable=True)
    restore = torch.argsort(shuf, dim=1)
    ref_mask = torch.ones(B, L)
    ref_mask[:, :keep] = 0
    ref_mask = torch.gather(ref_mask, 1, restore)
    assert torch.equal(ids_restore.cpu(), restore) and torch.equal(mask.cpu(), ref_mask)
    ref_rows = torch.cat([torch.zeros(B, 1, dtype=torch.long), shuf[:, :keep] + 1], 1) + torch.arange(B).view(B, 1) * (L + 1)
    assert torch.equal(keep_rows.cpu().view(B, keep + 1), ref_rows)

    P, C, H = 4, 3, 24
    img = torch.randn(B, C, H, H, generator=g)
    x = img.reshape(B, C, H // P, P, H // P, P)
    pat = torch.einsum("nchpwq->nhwpqc", x).reshape(B, (H // P) ** 2, P * P * C)
    assert torch.equal(ops.mim_targets(img.to(dev()), P, False).cpu(), pat)
    tgt = (pat - pat.mean(-1, keepdim=True)) / (pat.var(-1, keepdim=True) + 1.e-6) ** .5
    close(ops.mim_targets(img.to(dev()), P, True), tgt, 1e-5, 1e-5, "norm-pix target")

    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-2)):
        full = torch.randn(B, L + 1, P * P * C, generator=g).to(dtype)
        xr = full.float().clone().requires_grad_(True)
        per = ((xr[:, 1:, :] - tgt) ** 2).mean(-1)
        ref = (per * ref_mask).sum() / ref_mask.sum()
        (3.0 * ref).backward()
        xd = full.to(dev()).requires_grad_(True)
        loss = ops.mim_loss(xd, tgt.to(dev()), mask)
        (3.0 * loss).backward()
        assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()) + 1e-6
        close(xd.grad, xr.grad, tol, 1e-7, f"mim dlogits {dtype}")
        assert float(xd.grad[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(577, 577, False), (145, 145, True), (32, 577, False), (577, 32, True), (33, 65, True)])
@pytest.mark.parametrize("drop", [None, (0.1, 77)])
def test_attention_backward_generations_agree_bit_for_bit(shape, drop):
    """Round-4 backward kernels (csrc/attention.hip: attn_bwd_dq2_kernel / attn_bwd_dkdv2_kernel -- ONE LDS image per streamed
    operand read by rows and transposed, two tiles per barrier) against the round-3 kernels (launch_flags =
    M3AE_ATTN_LEGACY_KERNELS): same math and the same accumulation order, so dq / dk / dv are identical, ragged tails, key masks
    and dropout included.  (Both generations are held to the fp32 torch reference by test_attention_fwd_bwd above.)"""
    Lq, Lk, masked = shape
    B, H, D = 3, 12, 768
    q = rnd(B, Lq, D, dtype=torch.bfloat16, seed=51)
    kv = rnd(B, Lk, 2 * D, dtype=torch.bfloat16, seed=52)
    k, v = kv[..., :D], kv[..., D:]
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, device="cuda")
        mask[:, Lk - 3:] = -10000.0
    o, lse = ops.attn_forward(q, k, v, H, mask, dropout=drop)
    do = rnd(B, Lq, D, dtype=torch.bfloat16, seed=53)
    outs = []
    try:
        for legacy in (False, True):
            ops.ATTN_LEGACY = legacy
            dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
            ops.attn_backward(q, k, v, o, lse, do, dq, dkv[..., :D], dkv[..., D:], H, mask, dropout=drop)
            outs.append((dq, dkv))
    finally:
        ops.ATTN_LEGACY = False
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][0].float().abs().max().item() > 0
