"""Autograd operators of the hot path, each a thin host wrapper over the C ABI (include/m3ae_hip.h).

PyTorch supplies device memory, the stream and the autograd tape; every FLOP on the path runs in libm3ae_hip.so.
Parameter gradients are ACCUMULATED IN PLACE into `param.grad` (fp32, normally a view of ParamStore's flat
gradient buffer) by the wgrad / reduction kernels themselves, and the Functions return None for them: no autograd
accumulation kernels, and the data-parallel reducer (m3ae_amd/ddp.py) is told the moment a gradient is complete
through `grad_ready_hook`.
"""
import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_MULAUX, ACT_NONE, ACT_QUICKGELU, ACT_RELU, ACT_TANH, BF16, F32, AttnDesc, GemmDesc, XattnDesc,
                   check)

grad_ready_hook = None  # callable(param) set by the DDP reducer
# forward activation GEMMs save act'(pre-activation) for their backward GEMM (A/B knob: M3AE_SAVE_DACT=0 saves the
# pre-activation and re-evaluates the derivative in the backward epilogue, the first scheme of this repo)
SAVE_DACT = os.environ.get("M3AE_SAVE_DACT", "1") != "0"
# host-side launch policy, set by ddp.FlatGradReducer while collectives run next to backward (per-call flag in the GEMM
# descriptor: the library itself keeps no state)
NT_NO_PERSISTENT = os.environ.get("M3AE_NT_NO_PERSISTENT", "0") == "1"   # (set by ddp.FlatGradReducer.attach; the env default is for A/B runs)
# diagnostic per-call kernel selectors (m3ae_gemm_desc.launch_flags; -1 / 0 = by shape): tests compare kernel variants bit for
# bit, tools time them; the product path never sets them
GEMM_NT_VARIANT, GEMM_TN_VARIANT, GEMM_COL_GROUP = int(os.environ.get("M3AE_GEMM_NT_VARIANT", -1)), int(os.environ.get("M3AE_GEMM_TN_VARIANT", -1)), int(os.environ.get("M3AE_GEMM_COL_GROUP", 0))   # (env: A/B runs of tools)


ATTN_LEGACY = os.environ.get("M3AE_ATTN_LEGACY", "0") == "1"   # round-3 attention kernels (tests / tools compare the generations)
GEMM_ST_POLICY = int(os.environ.get("M3AE_GEMM_ST_POLICY", 0))   # output-store cache policy selector (0: the kernel's default)


def _gemm_flags():
    return ((1 if NT_NO_PERSISTENT else 0) | (((GEMM_NT_VARIANT + 1) & 0xF) << 8) | (((GEMM_TN_VARIANT + 1) & 0xF) << 12)
            | ((GEMM_COL_GROUP & 0xF) << 16) | ((GEMM_ST_POLICY & 0x3) << 20))
PROFILE = None  # when a list: every GEMM / attention launch is bracketed by HIP events on the launch stream


def _prof_begin():
    if PROFILE is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_end(e0, kind, dims):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROFILE.append((kind, dims, e0, e1))


# dropout seeds: every dropout site of every forward call draws a fresh 64-bit seed from this counter stream; the
# site stores it for its backward.  `set_dropout_seed` makes a run reproducible.
_drop_base, _drop_ctr = 0x5EED, 0


def set_dropout_seed(seed):
    global _drop_base, _drop_ctr
    _drop_base, _drop_ctr = int(seed) & 0xFFFFFFFF, 0


# Dropout salt (ABI 3): a device uint32 every dropout kernel folds into its mask key.  None in eager runs (the host draws a fresh
# seed per site and step); graph.GraphedStep points it at its per-replay counter while it captures, so the frozen seeds of the
# captured launches still give new masks at every replay.
DROPOUT_SALT = None


def _salt():
    return None if DROPOUT_SALT is None else C.c_void_p(DROPOUT_SALT.data_ptr())


def next_dropout_seed():
    global _drop_ctr
    _drop_ctr += 1
    return ((_drop_base << 32) | (_drop_ctr & 0xFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_dev_index = None


def _stream():
    """The caller's current HIP stream as a raw handle.  (torch.cuda.current_stream() builds a Stream object through three
    layers of device-index helpers: 8 us a call, 14 % of the host time of a step at ~1600 calls -- tools/host_profile.py.)"""
    global _dev_index
    if _raw_stream is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()   # one process drives one GPU (bench.py / trainer: torch.cuda.set_device first)
    return C.c_void_p(_raw_stream(_dev_index))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_cuda(t):
    if not t.is_cuda:
        raise _lib.M3AEHipError("m3ae_amd ops run on the GPU only (no CPU fallback); got a CPU tensor")
    # _stream() hands out the raw current stream of ONE device (the first one an op ran on: one process drives one GPU): a
    # tensor of another device would be launched on that device's stream handle
    if _dev_index is not None and t.device.index != _dev_index:
        raise _lib.M3AEHipError(f"m3ae_amd ops were first used on cuda:{_dev_index}; got a tensor on {t.device} "
                                f"(one process drives one GPU: call torch.cuda.set_device before the first op)")


def compute_weight(w):
    """The tensor a GEMM reads for parameter `w`: its bf16 shadow in perf mode, itself in fp32 mode."""
    return getattr(w, "m3ae_c", w)


# ----------------------------------------------------------------------------------------------------------
# raw GEMM
# ----------------------------------------------------------------------------------------------------------
def gemm(a, a_sm, a_sk, b, b_sk, b_sn, c, c_sm, M, N, K, *, alpha=1.0, accumulate=False, bias=None, act=ACT_NONE,
         preact=None, residual=None, dact_aux=None, dact=ACT_NONE, force_generic=False, batch=(1, 1),
         a_sb=(0, 0), b_sb=(0, 0), c_sb=(0, 0), a_rowsum=None, dropout=None, preact_grad=False):
    _need_cuda(c)
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.batch1, d.batch2 = batch
    d.A, d.a_sm, d.a_sk, d.a_sb1, d.a_sb2 = a.data_ptr(), a_sm, a_sk, a_sb[0], a_sb[1]
    d.B, d.b_sk, d.b_sn, d.b_sb1, d.b_sb2 = b.data_ptr(), b_sk, b_sn, b_sb[0], b_sb[1]
    d.C, d.c_sm, d.c_sn, d.c_sb1, d.c_sb2 = c.data_ptr(), c_sm, 1, c_sb[0], c_sb[1]
    d.dtype_a, d.dtype_b, d.dtype_c = _dt(a), _dt(b), _dt(c)
    d.alpha = alpha
    d.accumulate = int(accumulate)
    if bias is not None:
        assert bias.dtype == torch.float32
        d.bias = bias.data_ptr()
    d.act = act
    for name, t in (("preact", preact), ("residual", residual), ("dact_aux", dact_aux)):
        if t is not None:
            assert t.dtype == c.dtype and t.stride(-1) == 1 and t.stride(-2) == c_sm, name
            setattr(d, name, t.data_ptr())
    d.dact = dact
    d.preact_grad = int(preact_grad)
    d.launch_flags = _gemm_flags()
    d.force_generic = int(force_generic)
    if a_rowsum is not None:
        assert a_rowsum.dtype == torch.float32 and a_rowsum.numel() >= M and batch == (1, 1)
        d.a_rowsum = a_rowsum.data_ptr()
    if dropout is not None and dropout[0] > 0:
        d.dropout_p, d.dropout_seed = dropout
        d.dropout_salt = _salt()
    e0 = _prof_begin()
    check(_lib.lib().m3ae_gemm(C.byref(d), _stream()), "m3ae_gemm")
    if e0 is not None:
        _prof_end(e0, "gemm:" + last_gemm_path(), (M, N, K, batch[0] * batch[1]))


def last_gemm_path():
    return _lib.lib().m3ae_last_gemm_path().decode()


def _rows(x):
    """View x [..., K] as 2-D rows (M, K) without copying; returns (tensor2d, M, K, row_stride)."""
    K = x.shape[-1]
    if x.dim() == 2:
        assert x.stride(1) == 1
        return x, x.shape[0], K, x.stride(0)
    if not x.is_contiguous():
        x = x.contiguous()
    x2 = x.view(-1, K)
    return x2, x2.shape[0], K, K


_LAUNCH_STREAM = None


def launch_stream():
    """The process-wide HIGH-priority HIP stream the step is issued on (bench.py, trainer.py).  The model's text half runs on a second,
    normal-priority stream beside the image half (modules/m3ae_module.py); with the caller's stream at high priority the hardware
    dispatches the image kernels' workgroups first and the text kernels fill what they leave -- measured +0.75 / +1.4 % on the step at
    per-GPU batch 256, same box, alternating runs (profiles/r04_launch_stream_priority_ab.log; a high-priority SIDE stream: nothing)."""
    global _LAUNCH_STREAM
    if _LAUNCH_STREAM is None:
        _LAUNCH_STREAM = torch.cuda.Stream(priority=-1)
    return _LAUNCH_STREAM


def use_launch_stream():
    """Make launch_stream() this thread's current stream, ordered behind everything already queued on the device; returns the stream
    that was current (torch.cuda.set_stream(prev) restores it).  M3AE_LAUNCH_PRIORITY=normal keeps the caller's stream (A/B runs)."""
    prev = torch.cuda.current_stream()
    if os.environ.get("M3AE_LAUNCH_PRIORITY", "high") != "high":
        return prev
    torch.cuda.synchronize()
    torch.cuda.set_stream(launch_stream())
    return prev


def mm_nt(x2, ldx, M, w, bias=None, act=ACT_NONE, residual=None, want_preact=False, out_dtype=None, dact_aux=None,
          dact=ACT_NONE, force_generic=False, alpha=1.0, dropout=None, preact_grad=False):
    """y[M,N] = epi(alpha * x2[M,K] . w[N,K]^T).  want_preact + preact_grad: the second output is act'(pre-activation)
    (consumed by a backward GEMM with dact=ACT_MULAUX) instead of the pre-activation itself."""
    N, K = w.shape
    y = torch.empty((M, N), dtype=out_dtype or x2.dtype, device=x2.device)
    pre = torch.empty_like(y) if want_preact else None
    gemm(x2, ldx, 1, w, 1, w.stride(0), y, N, M, N, K, bias=bias, act=act, preact=pre, residual=residual,
         dact_aux=dact_aux, dact=dact, force_generic=force_generic, alpha=alpha, dropout=dropout,
         preact_grad=preact_grad and want_preact)
    return y, pre


def mm_dgrad(dy, w_param, dact_aux=None, dact=ACT_NONE, residual=None, alpha=1.0, dropout=None):
    """dx[M,K] = dy[M,N] . W[N,K]  (bf16: NT against the transposed shadow; fp32: strided generic)."""
    M, N = dy.shape
    wt = getattr(w_param, "m3ae_t", None)
    if wt is not None:
        K = wt.shape[0]
        dx = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
        gemm(dy, dy.stride(0), 1, wt, 1, wt.stride(0), dx, K, M, K, N, dact_aux=dact_aux, dact=dact, residual=residual,
             alpha=alpha, dropout=dropout)
    else:
        w = compute_weight(w_param)
        K = w.shape[1]
        dx = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
        gemm(dy, dy.stride(0), 1, w, w.stride(0), 1, dx, K, M, K, N, dact_aux=dact_aux, dact=dact, residual=residual,
             alpha=alpha, dropout=dropout)
    return dx


def mm_wgrad(dy, x2, ldx, w_param, b_param=None, alpha=1.0):
    """w.grad[N,K] += dy[M,N]^T . x2[M,K]  (fp32 accumulate in place); with b_param also b.grad[N] += colsum(dy),
    fused into the same kernel (row sums of the A operand dy^T)."""
    want_b = b_param is not None and b_param.requires_grad
    if not w_param.requires_grad:
        if want_b:
            bias_grad(dy, b_param)
        return
    g = _grad_buf(w_param)
    M, N = dy.shape
    K = g.shape[1]
    gemm(dy, 1, dy.stride(0), x2, ldx, 1, g, g.stride(0), N, K, M, accumulate=True, alpha=alpha,
         a_rowsum=_grad_buf(b_param) if want_b else None)
    _done(w_param)
    if want_b:
        _done(b_param)


def bias_grad(dy, b_param):
    if b_param is None or not b_param.requires_grad:
        return
    g = _grad_buf(b_param)
    M, N = dy.shape
    check(_lib.lib().m3ae_colsum(_p(dy), _p(g), M, N, dy.stride(0), _dt(dy), 1, _stream()), "m3ae_colsum")
    _done(b_param)


def act_bwd(dy, pre, act):
    dx = torch.empty_like(dy)
    check(_lib.lib().m3ae_act_bwd(_p(dy), _p(pre), _p(dx), dy.numel(), act, _dt(dy), _stream()), "m3ae_act_bwd")
    return dx


# ----------------------------------------------------------------------------------------------------------
# Linear / MLP
# ----------------------------------------------------------------------------------------------------------
def _members(p):
    return getattr(p, "members", None) or ([p] if p is not None else [])


def _done(p):
    if grad_ready_hook is not None and p is not None:
        for m in _members(p):
            grad_ready_hook(m)


def _grad_buf(p):
    if getattr(p, "members", None) is not None:  # PackedParam: members' grads are adjacent views of the flat buffer
        g = p.grad
        if g is None:
            raise _lib.M3AEHipError("packed parameters need ParamStore-managed gradients")
        return g
    if p.grad is None:
        p.grad = torch.zeros_like(p, dtype=torch.float32)
    return p.grad


def add(a, b, out=None):
    """out = a + b on the library's streaming add (same shape / dtype, contiguous)."""
    _need_cuda(a)
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a) if out is None else out
    check(_lib.lib().m3ae_add(_p(a), _p(b), _p(out), a.numel(), _dt(a), _stream()), "m3ae_add")
    return out


class Fork2Fn(torch.autograd.Function):
    """Identity with two outputs for a tensor that feeds two consumers (a fusion layer's x / y feed the text layer AND the image
    layer of the pair, m3ae_module.py:269-278): the two gradients meet here and are summed by the library's add instead of by
    autograd's accumulation (12 ATen adds per step in round 3).  Runs on the stream of its forward, i.e. the producer's."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None:
            return gb
        if gb is None:
            return ga
        return add(ga, gb)


def fork2(x):
    return Fork2Fn.apply(x) if (x.requires_grad and torch.is_grad_enabled()) else (x, x)


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b + extra_bias) (+ residual).  nn.Linear sites of clip_model.py / bert_model.py /
    m3ae_module.py.  `weight` / `bias` are Parameters or PackedParams; `anchors` are the underlying Parameters of a
    PackedParam (graph recording only)."""

    @staticmethod
    def forward(ctx, x, residual, extra_bias, weight, bias, act, alpha, *anchors):
        x2, M, K, ldx = _rows(x)
        w = compute_weight(weight)
        b = None if bias is None else (bias.data if hasattr(bias, "members") else bias.detach())
        if extra_bias is not None:  # e.g. + modality_type_embeddings row (m3ae_module.py:260-263)
            b = extra_bias.detach().float() if b is None else b + extra_bias.detach().float()
        res2 = None
        if residual is not None:
            res2 = residual.contiguous().view(M, -1)
        y, pre = mm_nt(x2, ldx, M, w, bias=b, act=act, residual=res2, want_preact=(act != ACT_NONE), alpha=alpha)
        ctx.save_for_backward(x2, pre)
        ctx.weight, ctx.bias, ctx.act, ctx.ldx, ctx.alpha = weight, bias, act, ldx, alpha
        ctx.x_shape, ctx.has_res = x.shape, residual is not None
        ctx.x_needs = x.requires_grad
        ctx.extra_needs = extra_bias is not None and extra_bias.requires_grad
        ctx.n_anchor = len(anchors)
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, pre = ctx.saved_tensors
        N = dy.shape[-1]
        dy2 = dy.contiguous().view(-1, N)
        dres = dy if ctx.has_res else None
        dz = act_bwd(dy2, pre, ctx.act) if ctx.act != ACT_NONE else dy2
        dextra = None
        dx = None
        if ctx.x_needs:   # before the weight gradient reports the parameter: an optimizer-in-backward update of this weight's
            dx = mm_dgrad(dz, ctx.weight, alpha=ctx.alpha).view(ctx.x_shape)   # bucket is then ordered behind this read of it
        if ctx.extra_needs:
            mm_wgrad(dz, x2, ctx.ldx, ctx.weight, alpha=ctx.alpha)
            dextra = torch.empty(N, dtype=torch.float32, device=dz.device)
            check(_lib.lib().m3ae_colsum(_p(dz), _p(dextra), dz.shape[0], N, dz.stride(0), _dt(dz), 0, _stream()),
                  "m3ae_colsum")
            if ctx.bias is not None and ctx.bias.requires_grad:
                _grad_buf(ctx.bias).add_(dextra)
                _done(ctx.bias)
        else:
            mm_wgrad(dz, x2, ctx.ldx, ctx.weight, ctx.bias, alpha=ctx.alpha)
        return (dx, dres, dextra, None, None, None, None) + (None,) * ctx.n_anchor


class GatherLinearFn(torch.autograd.Function):
    """y = act(x[:, 0] W^T + b): Pooler (prediction_heads.py:15-18).  The token-0 rows are addressed in place
    through the GEMM's row stride; the backward scatters into a zeroed [B, L, D] gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        B, L, D = x.shape
        xc = x.contiguous()
        w = compute_weight(weight)
        y, pre = mm_nt(xc, L * D, B, w, bias=bias, act=act, want_preact=(act != ACT_NONE))
        ctx.save_for_backward(xc, pre)
        ctx.weight, ctx.bias, ctx.act = weight, bias, act
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, pre = ctx.saved_tensors
        B, L, D = xc.shape
        dy2 = dy.contiguous()
        dz = act_bwd(dy2, pre, ctx.act) if ctx.act != ACT_NONE else dy2
        mm_wgrad(dz, xc, L * D, ctx.weight, ctx.bias)
        dx = torch.zeros_like(xc)
        wt = getattr(ctx.weight, "m3ae_t", None)
        N = dz.shape[1]
        if wt is not None:
            gemm(dz, N, 1, wt, 1, wt.stride(0), dx, L * D, B, D, N)
        else:
            w = compute_weight(ctx.weight)
            gemm(dz, N, 1, w, w.stride(0), 1, dx, L * D, B, D, N)
        return dx, None, None, None


class MLPFn(torch.autograd.Function):
    """y = act(x W1^T + b1) W2^T + b2 (+ residual).  BertIntermediate + BertOutput.dense (bert_model.py:416-440)
    and the CLIP mlp (clip_model.py:46-50).  The activation derivative is fused into the dgrad GEMM's epilogue."""

    @staticmethod
    def forward(ctx, x, residual, w1, b1, w2, b2, act):
        x2, M, K, ldx = _rows(x)
        g, u = mm_nt(x2, ldx, M, compute_weight(w1), bias=b1, act=act, want_preact=True, preact_grad=SAVE_DACT)
        res2 = residual.contiguous().view(M, -1) if residual is not None else None
        y, _ = mm_nt(g, g.stride(0), M, compute_weight(w2), bias=b2, residual=res2)
        ctx.save_for_backward(x2, u, g)
        ctx.p = (w1, b1, w2, b2)
        ctx.act, ctx.ldx, ctx.x_shape, ctx.has_res = act, ldx, x.shape, residual is not None
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, u, g = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.p
        dy2 = dy.contiguous().view(-1, dy.shape[-1])
        dres = dy if ctx.has_res else None
        mm_wgrad(dy2, g, g.stride(0), w2, b2)
        du = mm_dgrad(dy2, w2, dact_aux=u, dact=ACT_MULAUX if SAVE_DACT else ctx.act)  # dU = (dY W2) * act'(U), act'(U) saved by the forward
        mm_wgrad(du, x2, ctx.ldx, w1, b1)
        dx = mm_dgrad(du, w1).view(ctx.x_shape)
        return dx, dres, None, None, None, None, None


def linear(x, weight, bias=None, act=ACT_NONE, residual=None, extra_bias=None, alpha=1.0):
    anchors = tuple(_members(weight)) if hasattr(weight, "members") else ()
    if hasattr(bias, "members"):
        anchors = anchors + tuple(bias.members)
    return LinearFn.apply(x, residual, extra_bias, weight, bias, act, alpha, *anchors)


def mlp(x, w1, b1, w2, b2, act, residual=None):
    return MLPFn.apply(x, residual, w1, b1, w2, b2, act)


# ----------------------------------------------------------------------------------------------------------
# LayerNorm
# ----------------------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act, rms):
        xc = x.contiguous()
        D = xc.shape[-1]
        M = xc.numel() // D
        y = torch.empty_like(xc)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        check(_lib.lib().m3ae_layernorm_fwd(_p(xc), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, D, eps, _dt(xc),
                                            act, int(rms), _stream()), "m3ae_layernorm_fwd")
        ctx.save_for_backward(xc, mean, rstd)
        ctx.gamma, ctx.beta, ctx.act, ctx.rms = gamma, beta, act, rms
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, mean, rstd = ctx.saved_tensors
        D = xc.shape[-1]
        M = xc.numel() // D
        dyc = dy.contiguous()
        dx = torch.empty_like(xc)
        L = _lib.lib()
        nblk = L.m3ae_layernorm_bwd_blocks(M)
        ws = torch.empty(2 * nblk * D, dtype=torch.float32, device=xc.device)
        gg = _grad_buf(ctx.gamma)
        gb = _grad_buf(ctx.beta) if ctx.beta is not None else None
        check(L.m3ae_layernorm_bwd(_p(dyc), _p(xc), _p(ctx.gamma), _p(ctx.beta), _p(mean), _p(rstd), _p(dx), None,
                                   _p(gg), _p(gb), _p(ws), M, D, _dt(xc), ctx.act, int(ctx.rms), _stream()),
              "m3ae_layernorm_bwd")
        _done(ctx.gamma)
        _done(ctx.beta)
        return dx, None, None, None, None, None


def layer_norm(x, gamma, beta, eps, act=ACT_NONE, rms=False):
    return LayerNormFn.apply(x, gamma, beta, eps, act, rms)


# ----------------------------------------------------------------------------------------------------------
# raw LayerNorm helpers (no autograd) for the fused block functions
# ----------------------------------------------------------------------------------------------------------
def ln_fwd_raw(x2, ln, act=ACT_NONE, rms=False):
    M, D = x2.shape
    y = torch.empty_like(x2)
    mean = None if rms else torch.empty(M, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x2.device)
    check(_lib.lib().m3ae_layernorm_fwd(_p(x2), _p(ln.weight), _p(ln.bias), _p(y), _p(mean), _p(rstd), M, D, ln.eps,
                                        _dt(x2), act, int(rms), _stream()), "m3ae_layernorm_fwd")
    return y, mean, rstd


def ln_bwd_raw(dy, x2, ln, mean, rstd, dx_add=None, act=ACT_NONE, rms=False, drop=None):
    """dx = LN'(dy) (+ dx_add); ln.weight.grad / ln.bias.grad accumulate in place.  With drop = (p, seed) returns
    (dx, dx_drop): dx_drop is dx under the dropout mask of the dense layer that fed this LayerNorm."""
    M, D = x2.shape
    L = _lib.lib()
    dx = torch.empty_like(x2)
    nblk = L.m3ae_layernorm_bwd_blocks(M)
    ws = torch.empty(2 * nblk * D, dtype=torch.float32, device=x2.device)
    train = ln.weight.requires_grad
    gg = _grad_buf(ln.weight) if train else None
    gb = _grad_buf(ln.bias) if (train and ln.bias is not None) else None
    if drop is not None and drop[0] > 0:
        assert dx_add is None and act == ACT_NONE and not rms
        dxd = torch.empty_like(x2)
        check(L.m3ae_layernorm_bwd_drop(_p(dy), _p(x2), _p(ln.weight), _p(ln.bias), _p(mean), _p(rstd), _p(dx), _p(dxd),
                                        drop[0], drop[1], _salt(), _p(gg), _p(gb), _p(ws), M, D, _dt(x2), _stream()),
              "m3ae_layernorm_bwd_drop")
        if train:
            _done(ln.weight)
            _done(ln.bias)
        return dx, dxd
    check(L.m3ae_layernorm_bwd(_p(dy), _p(x2), _p(ln.weight), _p(ln.bias), _p(mean), _p(rstd), _p(dx), _p(dx_add),
                               _p(gg), _p(gb), _p(ws), M, D, _dt(x2), act, int(rms), _stream()), "m3ae_layernorm_bwd")
    if train:
        _done(ln.weight)
        _done(ln.bias)
    return dx


# ----------------------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------------------
def _attn_desc(B, H, Lq, Lk, Dh, q, k, v, o, key_mask, pos_bias, scale, causal, lse, lse_stride, dtype):
    d = AttnDesc()
    d.B, d.H, d.Lq, d.Lk, d.Dh = B, H, Lq, Lk, Dh
    d.q, d.q_sb, d.q_sl = q.data_ptr(), q.stride(0), q.stride(1)
    d.k, d.k_sb, d.k_sl = k.data_ptr(), k.stride(0), k.stride(1)
    d.v, d.v_sb, d.v_sl = v.data_ptr(), v.stride(0), v.stride(1)
    d.o, d.o_sb, d.o_sl = o.data_ptr(), o.stride(0), o.stride(1)
    d.key_mask = key_mask.data_ptr() if key_mask is not None else None
    d.pos_bias = pos_bias.data_ptr() if pos_bias is not None else None
    d.scale, d.causal = scale, int(causal)
    d.lse = lse.data_ptr() if lse is not None else None
    d.lse_stride = lse_stride
    d.dtype = dtype
    d.launch_flags = _lib.ATTN_LEGACY_KERNELS if ATTN_LEGACY else 0
    return d


def _attn_ws(d, backward, device):
    n = _lib.lib().m3ae_attn_workspace_bytes(C.byref(d), int(backward))
    if n <= 0:
        return None
    ws = torch.empty(n, dtype=torch.uint8, device=device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), n
    return ws


def attn_forward(q, k, v, heads, key_mask=None, pos_bias=None, scale=None, causal=False, dropout=None):
    """q [B,Lq,D] / k,v [B,Lk,D] (last dim contiguous, any batch/token strides) -> o [B,Lq,D], lse."""
    _need_cuda(q)
    B, Lq, D = q.shape
    Lk = k.shape[1]
    Dh = D // heads
    if q.dtype == torch.bfloat16 and Dh != 64:
        # the MFMA attention kernels are specialised for 64-wide heads; other widths (the decoder head's 96) take the fp32
        # materialised-softmax kernels on fp32 copies of the projections: short sequences, a few % of that head's work
        o32, lse = attn_forward(q.float(), k.float(), v.float(), heads, key_mask, pos_bias, scale, causal, dropout)
        return o32.to(torch.bfloat16), lse
    scale = (1.0 / math.sqrt(Dh)) if scale is None else scale
    o = torch.empty((B, Lq, D), dtype=q.dtype, device=q.device)
    lse_stride = (Lq + 31) // 32 * 32
    # bf16 kernels write every row of the table, padding rows (>= Lq) included; the fp32 path leaves it untouched
    lse = (torch.empty if q.dtype == torch.bfloat16 else torch.zeros)((B, heads, lse_stride), dtype=torch.float32, device=q.device)
    d = _attn_desc(B, heads, Lq, Lk, Dh, q, k, v, o, key_mask, pos_bias, scale, causal, lse, lse_stride, _dt(q))
    if dropout is not None and dropout[0] > 0:
        d.dropout_p, d.dropout_seed = dropout
        d.dropout_salt = _salt()
    ws = _attn_ws(d, False, q.device)
    e0 = _prof_begin()
    check(_lib.lib().m3ae_attn_fwd(C.byref(d), _stream()), "m3ae_attn_fwd")
    _prof_end(e0, "attn_fwd", (B, heads, Lq, Lk, Dh))
    del ws
    return o, lse


def attn_backward(q, k, v, o, lse, do, dq, dk, dv, heads, key_mask=None, pos_bias=None, scale=None, causal=False,
                  d_pos_bias=None, dropout=None):
    B, Lq, D = q.shape
    Lk = k.shape[1]
    Dh = D // heads
    if q.dtype == torch.bfloat16 and Dh != 64:   # fp32 detour, as in attn_forward
        f = [t.float() for t in (q, k, v, o, do)]
        g = [torch.empty_like(t) for t in f[:3]]
        attn_backward(f[0], f[1], f[2], f[3].contiguous(), lse, f[4].contiguous(), g[0], g[1], g[2], heads, key_mask,
                      pos_bias, scale, causal, d_pos_bias, dropout)
        for dst, src in zip((dq, dk, dv), g):
            dst.copy_(src)
        return
    scale = (1.0 / math.sqrt(Dh)) if scale is None else scale
    assert do.stride() == o.stride() and dq.stride() == q.stride() and dk.stride() == k.stride() and dv.stride() == v.stride()
    d = _attn_desc(B, heads, Lq, Lk, Dh, q, k, v, o, key_mask, pos_bias, scale, causal, lse, lse.shape[-1], _dt(q))
    # rows Lq .. lse_stride-1 are padding and must stay finite (0 * NaN in the dK/dV tile): the dQ kernel writes them as 0
    delta = torch.empty_like(lse) if q.dtype == torch.bfloat16 else torch.zeros_like(lse)
    d.d_o, d.dq, d.dk, d.dv, d.delta = do.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr()
    d.d_pos_bias = d_pos_bias.data_ptr() if d_pos_bias is not None else None
    if dropout is not None and dropout[0] > 0:
        d.dropout_p, d.dropout_seed = dropout
        d.dropout_salt = _salt()
    ws = _attn_ws(d, True, q.device)
    e0 = _prof_begin()
    check(_lib.lib().m3ae_attn_bwd(C.byref(d), _stream()), "m3ae_attn_bwd")
    _prof_end(e0, "attn_bwd", (B, heads, Lq, Lk, Dh))
    del ws, delta


class SelfAttnFn(torch.autograd.Function):
    """softmax(QK^T / sqrt(dh) + mask) V on a packed [B, L, 3D] projection (rows Q | K | V)."""

    @staticmethod
    def forward(ctx, qkv, key_mask, heads, dropout=None, causal=False):
        D = qkv.shape[-1] // 3
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        o, lse = attn_forward(q, k, v, heads, key_mask, dropout=dropout, causal=causal)
        ctx.save_for_backward(qkv, o, lse, key_mask)
        ctx.heads, ctx.dropout, ctx.causal = heads, dropout, causal
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, key_mask = ctx.saved_tensors
        D = qkv.shape[-1] // 3
        dqkv = torch.empty_like(qkv)
        attn_backward(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], o, lse, do.contiguous(), dqkv[..., :D],
                      dqkv[..., D:2 * D], dqkv[..., 2 * D:], ctx.heads, key_mask, dropout=ctx.dropout, causal=ctx.causal)
        return dqkv, None, None, None, None


class CrossAttnFn(torch.autograd.Function):
    """Q from this stream [B, Lq, D]; packed K | V [B, Lk, 2D] from the other stream (bert_model.py:275-278)."""

    @staticmethod
    def forward(ctx, q, kv, key_mask, heads, dropout=None):
        D = q.shape[-1]
        o, lse = attn_forward(q, kv[..., :D], kv[..., D:], heads, key_mask, dropout=dropout)
        ctx.save_for_backward(q, kv, o, lse, key_mask)
        ctx.heads, ctx.dropout = heads, dropout
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse, key_mask = ctx.saved_tensors
        D = q.shape[-1]
        # same strides as the (possibly sliced) inputs: the kernels address dq / dk / dv with the q / k / v strides
        dq = torch.empty_strided(q.shape, q.stride(), dtype=q.dtype, device=q.device)
        dkv = torch.empty_strided(kv.shape, kv.stride(), dtype=kv.dtype, device=kv.device)
        attn_backward(q, kv[..., :D], kv[..., D:], o, lse, do.contiguous(), dq, dkv[..., :D], dkv[..., D:], ctx.heads,
                      key_mask, dropout=ctx.dropout)
        return dq, dkv, None, None, None


def self_attention(qkv, key_mask, heads, dropout=None, causal=False):
    """dropout = (p, seed): attention-probability dropout (mask rows (b*H + h)*Lq + q, columns k)."""
    return SelfAttnFn.apply(qkv, key_mask, heads, dropout, causal)


def cross_attention(q, kv, key_mask, heads, dropout=None):
    return CrossAttnFn.apply(q, kv, key_mask, heads, dropout)


# ----------------------------------------------------------------------------------------------------------
# fused transformer blocks: ONE autograd node per block, hand-written backward, every gradient join of the block
# (residual branches) folded into a dgrad-GEMM or LayerNorm-backward epilogue -- no autograd add / copy kernels.
# `P` is a plain namespace of parameter references built by the module (m3ae_amd/modules/*); `anchors` are the same
# parameters as tensors so that autograd records the node.
# ----------------------------------------------------------------------------------------------------------
def _bdata(b):
    return None if b is None else (b.data if hasattr(b, "members") else b.detach())


# the fused cross-attention sub-block (csrc/xattn.hip, csrc/xflash.hip): "auto" = whenever the shapes are covered AND absorbing
# the long side's projection saves work (heads x text tokens < hidden width: 32 text tokens; at 64 the absorbed products are as
# large as the projections they replace and the fused path measures SLOWER than the composition -- forward 459 vs 331 us,
# training +25 %, profiles/r03_xattn_T64_fused_vs_composition.log); "always" = whenever covered (tests, tools); "off" = always the
# composition q / kv GEMM + flash attention + output GEMM + LayerNorm
XATTN = os.environ.get("M3AE_XATTN", "auto")
# training (a backward will be asked): "auto" = fused forward + fused backward, "off" = the composition
XATTN_TRAIN = os.environ.get("M3AE_XATTN_TRAIN", "auto")
# ... from this per-call batch on: measured on one MI355X (profiles/r02_xattn_batch_rule.log), fused forward + backward
# against the composition per sub-block: B = 32 +25..30 % slower, B = 64 +9..13 % slower, B = 128 8..12 % faster, B = 256
# 11..12 % faster -- the per-sample products of the absorbed form need the batch to fill the chip
XATTN_TRAIN_MIN_BATCH = int(os.environ.get("M3AE_XATTN_TRAIN_MIN_BATCH", 96))
# A/B measurements (tools/, tests): the round-2 dir-1 forward chain (P through HBM) instead of the one-launch kernel
XATTN_LEGACY_CHAIN = False


def _xattn_desc(h2, B, L, other2, Lo, mask, P, pdrop, seeds):
    D = h2.shape[1]
    d = XattnDesc()
    d.dir = 0 if L <= Lo else 1
    d.B, d.Lq, d.Lk, d.D, d.H = B, L, Lo, D, P.heads
    d.x, d.y = h2.data_ptr(), other2.data_ptr()
    d.key_mask = mask.data_ptr() if mask is not None else None
    wq, wkv, wo = P.w_q, P.w_kv, P.w_o
    d.wq, d.wkv, d.wo = compute_weight(wq).data_ptr(), compute_weight(wkv).data_ptr(), compute_weight(wo).data_ptr()
    d.wq_t, d.wkv_t, d.wo_t = wq.m3ae_t.data_ptr(), wkv.m3ae_t.data_ptr(), wo.m3ae_t.data_ptr()
    d.bq, d.bkv, d.bo = _bdata(P.b_q).data_ptr(), _bdata(P.b_kv).data_ptr(), _bdata(P.b_o).data_ptr()
    d.ln_g, d.ln_b, d.ln_eps = P.ln.weight.data_ptr(), P.ln.bias.data_ptr(), P.ln.eps
    if pdrop > 0:
        d.dropout_p, d.seed_attn, d.seed_hidden = pdrop, seeds[0], seeds[1]
        d.dropout_salt = _salt()
    d.launch_flags = (_lib.XATTN_NO_PERSISTENT if NT_NO_PERSISTENT else 0) | (_lib.XATTN_LEGACY_CHAIN if XATTN_LEGACY_CHAIN else 0)
    return d


def xattn_supported(h2, L, other2, Lo, mask, P, backward=False):
    """The fused sub-block covers these shapes (forward); backward=True: and m3ae_xattn_bwd does, and every projection
    parameter is trainable (the fused backward accumulates all six gradients in place; a layer with frozen projections
    whose input still needs a gradient takes the composition)."""
    if XATTN == "off" or h2.dtype != torch.bfloat16 or other2.shape[1] != h2.shape[1]:
        return False
    if XATTN != "always" and P.heads * min(L, Lo) >= h2.shape[1]:
        return False                      # covered, but not profitable (see XATTN above)
    if getattr(P.w_q, "m3ae_t", None) is None or getattr(P.w_kv, "m3ae_t", None) is None or getattr(P.w_o, "m3ae_t", None) is None:
        return False
    d = XattnDesc()
    d.dir = 0 if L <= Lo else 1
    d.B, d.Lq, d.Lk, d.D, d.H = 1, L, Lo, h2.shape[1], P.heads
    d.launch_flags = _lib.XATTN_LEGACY_CHAIN if XATTN_LEGACY_CHAIN else 0
    if not backward:
        return bool(_lib.lib().m3ae_xattn_supported(C.byref(d)))
    if not all(p.requires_grad for p in (P.w_q, P.w_kv, P.w_o, P.b_q, P.b_kv, P.b_o)):
        return False
    return bool(_lib.lib().m3ae_xattn_bwd_supported(C.byref(d)))


def xattn_fwd(h2, B, L, other2, Lo, mask, P, pdrop=0.0, need_bwd=True):
    """BertAttention as crossattention (bert_model.py:480-488) through the fused kernels.  Returns (out, saved).
    need_bwd=False (forward-only call): the image-query direction keeps scores and probabilities on chip."""
    _need_cuda(h2)
    dev, D, H = h2.device, h2.shape[1], P.heads
    seeds = (next_dropout_seed(), next_dropout_seed()) if pdrop > 0 else None
    d = _xattn_desc(h2, B, L, other2, Lo, mask, P, pdrop, seeds)
    bf = torch.bfloat16
    T, R = min(L, Lo), H * min(L, Lo)
    e = lambda *shape, dt=bf: torch.empty(shape, dtype=dt, device=dev)
    t = {}
    if d.dir == 0:
        t["proj"] = e(B * L, D)
        t["prime"] = e(B, R, D)
        t["probs"] = e(B, R, 640)
        t["zctx"] = e(B, R, D)
        t["ctx"] = e(B * L, D)
        if pdrop > 0:
            t["probs_drop"] = e(B, R, 640)
            t["rowsum"] = e(B, R, dt=torch.float32)
    else:
        t["proj"] = e(B * Lo, 2 * D)
        t["prime"] = e(2, B, R, D)
        t["colbias"] = e(B, R, dt=torch.float32)
        if need_bwd or XATTN_LEGACY_CHAIN:
            t["probs"] = e(B, L, R)
            if pdrop > 0:
                t["probs_drop"] = e(B, L, R)
    t["s"] = e(B * L, D)
    t["out"] = e(B * L, D)
    t["mean"] = e(B * L, dt=torch.float32)
    t["rstd"] = e(B * L, dt=torch.float32)
    for k, v in t.items():
        setattr(d, k, v.data_ptr())
    e0 = _prof_begin()
    check(_lib.lib().m3ae_xattn_fwd(C.byref(d), _stream()), "m3ae_xattn_fwd")
    _prof_end(e0, "xattn_fwd", (B, H, L, Lo, D // H))
    return t["out"], ("xattn", h2, other2, mask, t, seeds, pdrop)


def xattn_bwd(dy, saved, B, L, Lo, P, need_dother=True):
    """Backward of xattn_fwd (m3ae_xattn_bwd): returns (dx, dother); parameter gradients accumulate in place."""
    _, h2, other2, mask, t, seeds, pdrop = saved
    dev, D, H = h2.device, h2.shape[1], P.heads
    d = _xattn_desc(h2, B, L, other2, Lo, mask, P, pdrop, seeds)
    for k, v in t.items():
        setattr(d, k, v.data_ptr())
    bf = torch.bfloat16
    e = lambda *shape, dt=bf: torch.empty(shape, dtype=dt, device=dev)
    dyc = dy.contiguous()
    dx = e(B * L, D)
    dother = e(B * Lo, D) if need_dother else None
    L_ = _lib.lib()
    w = {"ws_ds": e(B * L, D), "ws_dscores": torch.empty_like(t["probs"]), "ws_dprime": torch.empty_like(t["prime"]),
         "ws_dproj": torch.empty_like(t["proj"]), "ws_vec": e(3 * B * H * min(L, Lo), dt=torch.float32),
         "ws_ln": e(2 * L_.m3ae_layernorm_bwd_blocks(B * L) * D, dt=torch.float32)}
    if pdrop > 0:
        w["ws_dsd"] = e(B * L, D)
    if d.dir == 0:
        w["ws_dz"] = torch.empty_like(t["zctx"])
        w["ws_dctx"] = e(B * L, D)
    for k, v in w.items():
        setattr(d, k, v.data_ptr())
    d.d_out, d.dx = dyc.data_ptr(), dx.data_ptr()
    d.dy = dother.data_ptr() if dother is not None else None
    train = P.ln.weight.requires_grad
    grads = {"g_wq": P.w_q, "g_wkv": P.w_kv, "g_wo": P.w_o, "g_bq": P.b_q, "g_bkv": P.b_kv, "g_bo": P.b_o}
    for k, prm in grads.items():
        if not prm.requires_grad:
            raise _lib.M3AEHipError("the fused cross-attention backward needs trainable projection parameters")
        setattr(d, k, _grad_buf(prm).data_ptr())
    if train:
        d.g_ln_g, d.g_ln_b = _grad_buf(P.ln.weight).data_ptr(), _grad_buf(P.ln.bias).data_ptr()
    e0 = _prof_begin()
    check(L_.m3ae_xattn_bwd(C.byref(d), _stream()), "m3ae_xattn_bwd")
    _prof_end(e0, "xattn_bwd", (B, H, L, Lo, D // H))
    for prm in (P.w_o, P.b_o, P.w_kv, P.b_kv, P.w_q, P.b_q) + ((P.ln.weight, P.ln.bias) if train else ()):
        _done(prm)
    return dx, dother


def _attn_sub_fwd(h2, B, L, other2, Lo, mask, P, pdrop=0.0, fused_cross=False):
    """BertAttention (bert_model.py:367-413) on 2-D token-major activations. Returns (y, saved).
    pdrop > 0 (training): attention-probability dropout (:334) and hidden dropout on the output dense (:362)."""
    heads = P.heads
    D = h2.shape[1]
    if other2 is not None and fused_cross:
        need_bwd = getattr(P, "need_bwd", True)
        if xattn_supported(h2, L, other2, Lo, mask, P, backward=need_bwd):
            return xattn_fwd(h2, B, L, other2, Lo, mask, P, pdrop, need_bwd=need_bwd)
    da = (pdrop, next_dropout_seed()) if pdrop > 0 else None
    dh = (pdrop, next_dropout_seed()) if pdrop > 0 else None
    if other2 is None:
        qkv, _ = mm_nt(h2, D, B * L, compute_weight(P.w_qkv), bias=_bdata(P.b_qkv))
        v3 = qkv.view(B, L, 3 * D)
        o, lse = attn_forward(v3[..., :D], v3[..., D:2 * D], v3[..., 2 * D:], heads, mask, dropout=da)
        proj = (qkv,)
    else:
        q, _ = mm_nt(h2, D, B * L, compute_weight(P.w_q), bias=_bdata(P.b_q))
        kv, _ = mm_nt(other2, other2.shape[1], B * Lo, compute_weight(P.w_kv), bias=_bdata(P.b_kv))
        kv3 = kv.view(B, Lo, 2 * D)
        o, lse = attn_forward(q.view(B, L, D), kv3[..., :D], kv3[..., D:], heads, mask, dropout=da)
        proj = (q, kv)
    o2 = o.view(B * L, D)
    s, _ = mm_nt(o2, D, B * L, compute_weight(P.w_o), bias=_bdata(P.b_o), residual=h2, dropout=dh)
    y, mean, rstd = ln_fwd_raw(s, P.ln)
    return y, (h2, other2, proj, o, lse, s, mean, rstd, mask, da, dh)


def _attn_sub_bwd(dy, saved, B, L, Lo, P, need_dother=True):
    if isinstance(saved[0], str):   # ("xattn", ...): the fused sub-block
        return xattn_bwd(dy, saved, B, L, Lo, P, need_dother)
    h2, other2, proj, o, lse, s, mean, rstd, mask, da, dh = saved
    D = h2.shape[1]
    if dh is not None:
        ds, dsd = ln_bwd_raw(dy, s, P.ln, mean, rstd, drop=dh)  # dsd: gradient of the (dropped) dense output
    else:
        ds = dsd = ln_bwd_raw(dy, s, P.ln, mean, rstd)
    o2 = o.view(B * L, D)
    mm_wgrad(dsd, o2, D, P.w_o, P.b_o)
    dctx = mm_dgrad(dsd, P.w_o)
    if other2 is None:
        (qkv,) = proj
        v3 = qkv.view(B, L, 3 * D)
        dqkv = torch.empty_like(qkv)
        d3 = dqkv.view(B, L, 3 * D)
        attn_backward(v3[..., :D], v3[..., D:2 * D], v3[..., 2 * D:], o, lse, dctx.view(B, L, D), d3[..., :D],
                      d3[..., D:2 * D], d3[..., 2 * D:], P.heads, mask, dropout=da)
        mm_wgrad(dqkv, h2, D, P.w_qkv, P.b_qkv)
        dhid = mm_dgrad(dqkv, P.w_qkv, residual=ds)  # + residual-branch gradient, fused
        return dhid, None
    q, kv = proj
    kv3 = kv.view(B, Lo, 2 * D)
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    dkv3 = dkv.view(B, Lo, 2 * D)
    attn_backward(q.view(B, L, D), kv3[..., :D], kv3[..., D:], o, lse, dctx.view(B, L, D), dq.view(B, L, D),
                  dkv3[..., :D], dkv3[..., D:], P.heads, mask, dropout=da)
    mm_wgrad(dq, h2, D, P.w_q, P.b_q)
    mm_wgrad(dkv, other2, other2.shape[1], P.w_kv, P.b_kv)
    dhid = mm_dgrad(dq, P.w_q, residual=ds)
    dother = mm_dgrad(dkv, P.w_kv) if need_dother else None
    return dhid, dother


def _ffn_sub_fwd(h2, P, pdrop=0.0):
    """BertIntermediate + BertOutput (bert_model.py:416-442, 500-503); pdrop: hidden dropout on the output dense (:440)."""
    M, D = h2.shape
    dh = (pdrop, next_dropout_seed()) if pdrop > 0 else None
    g, u = mm_nt(h2, D, M, compute_weight(P.w1), bias=_bdata(P.b1), act=ACT_GELU, want_preact=True, preact_grad=SAVE_DACT)
    s, _ = mm_nt(g, g.shape[1], M, compute_weight(P.w2), bias=_bdata(P.b2), residual=h2, dropout=dh)
    y, mean, rstd = ln_fwd_raw(s, P.ln)
    return y, (h2, u, g, s, mean, rstd, dh)


def _ffn_sub_bwd(dy, saved, P):
    h2, u, g, s, mean, rstd, dh = saved
    if dh is not None:
        ds, dsd = ln_bwd_raw(dy, s, P.ln, mean, rstd, drop=dh)
    else:
        ds = dsd = ln_bwd_raw(dy, s, P.ln, mean, rstd)
    mm_wgrad(dsd, g, g.shape[1], P.w2, P.b2)
    du = mm_dgrad(dsd, P.w2, dact_aux=u, dact=ACT_MULAUX if SAVE_DACT else ACT_GELU)  # u holds gelu'(pre-activation)
    mm_wgrad(du, h2, h2.shape[1], P.w1, P.b1)
    return mm_dgrad(du, P.w1, residual=ds)


class BertCrossLayerFn(torch.autograd.Function):
    """BertCrossLayer.forward (bert_model.py:457-498): self-attn -> cross-attn -> FFN as one node."""

    @staticmethod
    def forward(ctx, h, other, mask_self, mask_other, P, *anchors):
        B, L, D = h.shape
        Lo = other.shape[1]
        h2 = h.contiguous().view(B * L, D)
        other2 = other.contiguous().view(B * Lo, other.shape[2])
        pd = getattr(P, "pdrop", 0.0)
        a, s1 = _attn_sub_fwd(h2, B, L, None, L, mask_self, P.attn, pd)
        # the fused cross-attention sub-block (csrc/xattn.hip) where the shapes are covered: forward-only calls always,
        # training with its fused backward (ops.XATTN_TRAIN)
        c, s2 = _attn_sub_fwd(a, B, L, other2, Lo, mask_other, P.cross, pd, fused_cross=getattr(P, "fused_cross", False))
        y, s3 = _ffn_sub_fwd(c, P.ffn, pd)
        ctx.saved = (s1, s2, s3)
        ctx.P, ctx.dims, ctx.n_anchor = P, (B, L, Lo, D), len(anchors)
        ctx.need_other = other.requires_grad
        return y.view(B, L, D)

    @staticmethod
    def backward(ctx, dy):
        B, L, Lo, D = ctx.dims
        s1, s2, s3 = ctx.saved
        ctx.saved = None
        P = ctx.P
        dc = _ffn_sub_bwd(dy.contiguous().view(B * L, D), s3, P.ffn)
        da, dother = _attn_sub_bwd(dc, s2, B, L, Lo, P.cross, need_dother=ctx.need_other)
        dh, _ = _attn_sub_bwd(da, s1, B, L, L, P.attn)
        return (dh.view(B, L, D), None if dother is None else dother.view(B, Lo, -1), None, None, None) + \
               (None,) * ctx.n_anchor


class BertSelfLayerFn(torch.autograd.Function):
    """BertSelfLayer == HF RobertaLayer (bert_model.py:506-546; m3ae_module.py:233-234)."""

    @staticmethod
    def forward(ctx, h, mask, P, *anchors):
        B, L, D = h.shape
        h2 = h.contiguous().view(B * L, D)
        pd = getattr(P, "pdrop", 0.0)
        a, s1 = _attn_sub_fwd(h2, B, L, None, L, mask, P.attn, pd)
        y, s3 = _ffn_sub_fwd(a, P.ffn, pd)
        ctx.saved = (s1, s3)
        ctx.P, ctx.dims, ctx.n_anchor = P, (B, L, D), len(anchors)
        return y.view(B, L, D)

    @staticmethod
    def backward(ctx, dy):
        B, L, D = ctx.dims
        s1, s3 = ctx.saved
        ctx.saved = None
        da = _ffn_sub_bwd(dy.contiguous().view(B * L, D), s3, ctx.P.ffn)
        dh, _ = _attn_sub_bwd(da, s1, B, L, L, ctx.P.attn)
        return (dh.view(B, L, D), None, None) + (None,) * ctx.n_anchor


class ClipBlockFn(torch.autograd.Function):
    """ResidualAttentionBlock.forward (clip_model.py:60-63), pre-LN: x += MHA(LN1(x)); x += MLP(LN2(x))."""

    @staticmethod
    def forward(ctx, x, P, *anchors):
        B, L, D = x.shape
        M = B * L
        x2 = x.contiguous().view(M, D)
        h1, m1, r1 = ln_fwd_raw(x2, P.ln1)
        qkv, _ = mm_nt(h1, D, M, compute_weight(P.w_in), bias=_bdata(P.b_in))
        v3 = qkv.view(B, L, 3 * D)
        o, lse = attn_forward(v3[..., :D], v3[..., D:2 * D], v3[..., 2 * D:], P.heads, None)
        xa, _ = mm_nt(o.view(M, D), D, M, compute_weight(P.w_out), bias=_bdata(P.b_out), residual=x2)
        h2, m2, r2 = ln_fwd_raw(xa, P.ln2)
        g, u = mm_nt(h2, D, M, compute_weight(P.w_fc), bias=_bdata(P.b_fc), act=ACT_QUICKGELU, want_preact=True,
                     preact_grad=SAVE_DACT)
        y, _ = mm_nt(g, g.shape[1], M, compute_weight(P.w_proj), bias=_bdata(P.b_proj), residual=xa)
        ctx.saved = (x2, m1, r1, h1, qkv, o, lse, xa, m2, r2, h2, u, g)
        ctx.P, ctx.dims, ctx.n_anchor = P, (B, L, D), len(anchors)
        return y.view(B, L, D)

    @staticmethod
    def backward(ctx, dy):
        B, L, D = ctx.dims
        M = B * L
        x2, m1, r1, h1, qkv, o, lse, xa, m2, r2, h2, u, g = ctx.saved
        ctx.saved = None
        P = ctx.P
        dy2 = dy.contiguous().view(M, D)
        mm_wgrad(dy2, g, g.shape[1], P.w_proj, P.b_proj)
        du = mm_dgrad(dy2, P.w_proj, dact_aux=u, dact=ACT_MULAUX if SAVE_DACT else ACT_QUICKGELU)
        mm_wgrad(du, h2, D, P.w_fc, P.b_fc)
        dh2 = mm_dgrad(du, P.w_fc)
        dxa = ln_bwd_raw(dh2, xa, P.ln2, m2, r2, dx_add=dy2)  # + residual branch, fused into LN backward
        mm_wgrad(dxa, o.view(M, D), D, P.w_out, P.b_out)
        dctx = mm_dgrad(dxa, P.w_out)
        v3 = qkv.view(B, L, 3 * D)
        dqkv = torch.empty_like(qkv)
        d3 = dqkv.view(B, L, 3 * D)
        attn_backward(v3[..., :D], v3[..., D:2 * D], v3[..., 2 * D:], o, lse, dctx.view(B, L, D), d3[..., :D],
                      d3[..., D:2 * D], d3[..., 2 * D:], P.heads, None)
        mm_wgrad(dqkv, h1, D, P.w_in, P.b_in)
        dh1 = mm_dgrad(dqkv, P.w_in)
        dx = ln_bwd_raw(dh1, x2, P.ln1, m1, r1, dx_add=dxa)
        return (dx.view(B, L, D), None) + (None,) * ctx.n_anchor


# ----------------------------------------------------------------------------------------------------------
# T5 pre-norm blocks (HF T5Block restated; m3ae_t5_mm_encoder_input.py:202,244): RMSNorm, no linear biases, no
# 1/sqrt(d) scaling, additive relative-position bias, ReLU FFN.  One autograd node per block.
# ----------------------------------------------------------------------------------------------------------
def _drop_raw(x2, drop):
    """dropout of a 2-D activation with an explicit (p, seed): the gradient entering a dropped sub-layer output."""
    if drop is None:
        return x2
    x2 = x2.contiguous()
    out = torch.empty_like(x2)
    check(_lib.lib().m3ae_dropout(_p(x2), _p(out), None, x2.shape[0], x2.shape[1], drop[0], drop[1], _salt(), _dt(x2), _stream()),
          "m3ae_dropout")
    return out


def _t5_attn_fwd(h2, B, L, src2, Ls, P, bias, causal, pdrop=0.0, kv=None):
    # kv: the projected keys / values of `src2` computed earlier ([B * Ls, 2 inner]; generation re-uses them every step)
    # HF T5 (third party, transformers 4.6.0): attention-weight dropout inside T5Attention and `hidden + dropout(attn)`
    da = (pdrop, next_dropout_seed()) if pdrop > 0 else None
    dh = (pdrop, next_dropout_seed()) if pdrop > 0 else None
    n, _, rstd = ln_fwd_raw(h2, P.ln, rms=True)
    D = n.shape[1]
    inner = P.w_o.shape[1]
    if src2 is None and kv is None:
        qkv, _ = mm_nt(n, D, B * L, compute_weight(P.w_qkv))
        v3 = qkv.view(B, L, 3 * inner)
        o, lse = attn_forward(v3[..., :inner], v3[..., inner:2 * inner], v3[..., 2 * inner:], P.heads, None, bias,
                              scale=1.0, causal=causal, dropout=da)
        proj = (qkv,)
    else:
        q, _ = mm_nt(n, D, B * L, compute_weight(P.w_q))
        if kv is None:
            kv, _ = mm_nt(src2, src2.shape[1], B * Ls, compute_weight(P.w_kv))
        kv3 = kv.view(B, Ls, 2 * inner)
        o, lse = attn_forward(q.view(B, L, inner), kv3[..., :inner], kv3[..., inner:], P.heads, None, bias, scale=1.0,
                              causal=causal, dropout=da)
        proj = (q, kv)
    y, _ = mm_nt(o.view(B * L, inner), inner, B * L, compute_weight(P.w_o), residual=h2, dropout=dh)
    return y, (h2, rstd, n, proj, o, lse, src2, da, dh)


def t5_self_attn_step(h2, B, P, bias_row, cache, t, pdrop=0.0):
    """Generation: the T5 self-attention sub-layer for ONE new position t of every sequence.  h2 [B, D]; `cache` [B, Tmax,
    2 inner] holds the keys | values of positions < t and receives row t; bias_row [H, 1, t + 1] is the relative-position
    bias of the new query.  Same kernels, same order as _t5_attn_fwd on the whole prefix (whose last row this equals)."""
    da = (pdrop, next_dropout_seed()) if pdrop > 0 else None
    dh = (pdrop, next_dropout_seed()) if pdrop > 0 else None
    n, _, _ = ln_fwd_raw(h2, P.ln, rms=True)
    D = n.shape[1]
    inner = P.w_o.shape[1]
    qkv, _ = mm_nt(n, D, B, compute_weight(P.w_qkv))
    cache[:, t].copy_(qkv[:, inner:])
    q = qkv[:, :inner].unsqueeze(1)                        # [B, 1, inner], batch stride 3 inner
    o, _ = attn_forward(q, cache[:, :t + 1, :inner], cache[:, :t + 1, inner:], P.heads, None, bias_row, scale=1.0,
                        causal=False, dropout=da)
    y, _ = mm_nt(o.view(B, inner), inner, B, compute_weight(P.w_o), residual=h2, dropout=dh)
    return y


def _t5_attn_bwd(dy, saved, B, L, Ls, P, bias, causal, dbias, need_dh=True, need_dsrc=True):
    h2, rstd, n, proj, o, lse, src2, da, dh_drop = saved
    need_dh = need_dh or P.ln.weight.requires_grad  # the RMSNorm scale gradient comes out of the same kernel
    D = n.shape[1]
    inner = P.w_o.shape[1]
    dyd = _drop_raw(dy, dh_drop)  # gradient of the (dropped) sub-layer output; the residual branch keeps dy itself
    mm_wgrad(dyd, o.view(B * L, inner), inner, P.w_o)
    dctx = mm_dgrad(dyd, P.w_o).view(B, L, inner)
    dsrc = None
    if src2 is None:
        (qkv,) = proj
        v3 = qkv.view(B, L, 3 * inner)
        dqkv = torch.empty_like(qkv)
        d3 = dqkv.view(B, L, 3 * inner)
        attn_backward(v3[..., :inner], v3[..., inner:2 * inner], v3[..., 2 * inner:], o, lse, dctx, d3[..., :inner],
                      d3[..., inner:2 * inner], d3[..., 2 * inner:], P.heads, None, bias, scale=1.0, causal=causal,
                      d_pos_bias=dbias, dropout=da)
        mm_wgrad(dqkv, n, D, P.w_qkv)
        dn = mm_dgrad(dqkv, P.w_qkv) if need_dh else None
    else:
        q, kv = proj
        kv3 = kv.view(B, Ls, 2 * inner)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        dkv3 = dkv.view(B, Ls, 2 * inner)
        attn_backward(q.view(B, L, inner), kv3[..., :inner], kv3[..., inner:], o, lse, dctx, dq.view(B, L, inner),
                      dkv3[..., :inner], dkv3[..., inner:], P.heads, None, bias, scale=1.0, causal=causal,
                      d_pos_bias=dbias, dropout=da)
        mm_wgrad(dq, n, D, P.w_q)
        mm_wgrad(dkv, src2, src2.shape[1], P.w_kv)
        dn = mm_dgrad(dq, P.w_q) if need_dh else None
        dsrc = mm_dgrad(dkv, P.w_kv) if need_dsrc else None
    dh = ln_bwd_raw(dn, h2, P.ln, None, rstd, dx_add=dy, rms=True) if need_dh else None
    return dh, dsrc


def _t5_ff_fwd(h2, P, pdrop=0.0):
    # HF T5DenseReluDense: wo(dropout(relu(wi(x)))); T5LayerFF: hidden + dropout(ff)
    d1 = (pdrop, next_dropout_seed()) if pdrop > 0 else None
    d2 = (pdrop, next_dropout_seed()) if pdrop > 0 else None
    n, _, rstd = ln_fwd_raw(h2, P.ln, rms=True)
    M, D = n.shape
    g, u = mm_nt(n, D, M, compute_weight(P.w1), act=ACT_RELU, want_preact=True, preact_grad=SAVE_DACT, dropout=d1)
    y, _ = mm_nt(g, g.shape[1], M, compute_weight(P.w2), residual=h2, dropout=d2)
    return y, (h2, rstd, n, u, g, d1, d2)


def _t5_ff_bwd(dy, saved, P):
    h2, rstd, n, u, g, d1, d2 = saved
    dyd = _drop_raw(dy, d2)
    mm_wgrad(dyd, g, g.shape[1], P.w2)                     # g is the dropped activation the forward multiplied by W2
    du = mm_dgrad(dyd, P.w2, dact_aux=u, dact=ACT_MULAUX if SAVE_DACT else ACT_RELU, dropout=d1)
    mm_wgrad(du, n, n.shape[1], P.w1)
    dn = mm_dgrad(du, P.w1)
    return ln_bwd_raw(dn, h2, P.ln, None, rstd, dx_add=dy, rms=True)


class T5EncBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, pos_bias, P, *anchors):
        B, L, D = h.shape
        h2 = h.contiguous().view(B * L, D)
        bias = pos_bias.detach() if pos_bias is not None else None
        pd = getattr(P, "pdrop", 0.0)
        a, s1 = _t5_attn_fwd(h2, B, L, None, L, P.attn, bias, False, pd)
        y, s2 = _t5_ff_fwd(a, P.ffn, pd)
        ctx.saved = (s1, s2, bias)
        ctx.P, ctx.dims, ctx.n_anchor = P, (B, L, D), len(anchors)
        ctx.need_h = h.requires_grad
        ctx.need_bias = pos_bias is not None and pos_bias.requires_grad
        return y.view(B, L, D)

    @staticmethod
    def backward(ctx, dy):
        B, L, D = ctx.dims
        s1, s2, bias = ctx.saved
        ctx.saved = None
        dbias = torch.zeros_like(bias) if ctx.need_bias else None
        da = _t5_ff_bwd(dy.contiguous().view(B * L, D), s2, ctx.P.ffn)
        dh, _ = _t5_attn_bwd(da, s1, B, L, L, ctx.P.attn, bias, False, dbias, need_dh=ctx.need_h)
        return (None if dh is None else dh.view(B, L, D), dbias, None) + (None,) * ctx.n_anchor


class T5DecBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, enc, pos_bias, P, *anchors):
        B, T, D = h.shape
        Ls = enc.shape[1]
        h2 = h.contiguous().view(B * T, D)
        enc2 = enc.contiguous().view(B * Ls, enc.shape[2])
        bias = pos_bias.detach() if pos_bias is not None else None
        pd = getattr(P, "pdrop", 0.0)
        a, s1 = _t5_attn_fwd(h2, B, T, None, T, P.attn, bias, True, pd)
        c, s2 = _t5_attn_fwd(a, B, T, enc2, Ls, P.cross, None, False, pd)
        y, s3 = _t5_ff_fwd(c, P.ffn, pd)
        ctx.saved = (s1, s2, s3, bias)
        ctx.P, ctx.dims, ctx.n_anchor = P, (B, T, Ls, D), len(anchors)
        ctx.need_h, ctx.need_enc = h.requires_grad, enc.requires_grad
        ctx.need_bias = pos_bias is not None and pos_bias.requires_grad
        return y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        B, T, Ls, D = ctx.dims
        s1, s2, s3, bias = ctx.saved
        ctx.saved = None
        P = ctx.P
        dbias = torch.zeros_like(bias) if ctx.need_bias else None
        dc = _t5_ff_bwd(dy.contiguous().view(B * T, D), s3, P.ffn)
        da, denc = _t5_attn_bwd(dc, s2, B, T, Ls, P.cross, None, False, None, need_dsrc=ctx.need_enc)
        dh, _ = _t5_attn_bwd(da, s1, B, T, T, P.attn, bias, True, dbias, need_dh=ctx.need_h)
        return (None if dh is None else dh.view(B, T, D), None if denc is None else denc.view(B, Ls, -1), dbias,
                None) + (None,) * ctx.n_anchor


class EmbedRowsFn(torch.autograd.Function):
    """rows = table[ids] (T5 `shared` lookup for the teacher-forced decoder input).  `table` is the compute-dtype
    view of `weight`; the gradient (only when the embedding is trainable) is scatter-added into weight.grad."""

    @staticmethod
    def forward(ctx, ids, weight, table):
        ids = ids.contiguous()   # the kernel walks ids linearly: a strided view (e.g. prefix[:, -1:]) would read its neighbours
        out = torch.empty((ids.numel(), table.shape[1]), dtype=table.dtype, device=table.device)
        check(_lib.lib().m3ae_gather_rows(_p(table), _p(ids), _p(out), ids.numel(), table.shape[1], _dt(table),
                                          _stream()), "m3ae_gather_rows")
        ctx.save_for_backward(ids)
        ctx.weight = weight
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        if ctx.weight.requires_grad:  # rare path: the embedding is frozen in the reference's recipe
            _grad_buf(ctx.weight).index_add_(0, ids, dout.float())
            _done(ctx.weight)
        return None, None, None


# ----------------------------------------------------------------------------------------------------------
# embeddings / tokens
# ----------------------------------------------------------------------------------------------------------
class RobertaEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, word, pos, typ, pad_id, dtype):
        _need_cuda(ids)
        B, S = ids.shape
        D = word.shape[1]
        out = torch.empty((B, S, D), dtype=dtype, device=ids.device)
        check(_lib.lib().m3ae_roberta_embed_fwd(_p(ids), _p(word), _p(pos), _p(typ), _p(out), B, S, D, pad_id, _dt(out),
                                                _stream()), "m3ae_roberta_embed_fwd")
        ctx.save_for_backward(ids)
        ctx.p, ctx.pad_id = (word, pos, typ), pad_id
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        word, pos, typ = ctx.p
        B, S = ids.shape
        D = word.shape[1]
        d = dout.contiguous()
        check(_lib.lib().m3ae_roberta_embed_bwd(_p(ids), _p(d), _p(_grad_buf(word)), _p(_grad_buf(pos)),
                                                _p(_grad_buf(typ)), B, S, D, ctx.pad_id, _dt(d), _stream()),
              "m3ae_roberta_embed_bwd")
        for p in (word, pos, typ):
            _done(p)
        return None, None, None, None, None, None


def roberta_embed(ids, word, pos, typ, pad_id, dtype):
    return RobertaEmbedFn.apply(ids, word, pos, typ, pad_id, dtype)


class VitTokensFn(torch.autograd.Function):
    """conv1 (k = s = patch, no bias) as im2col + GEMM, prepend class_embedding, optional + positional_embedding
    (clip_model.py:94-99 / :110-116)."""

    @staticmethod
    def forward(ctx, img, conv_w, cls, pos, dtype, add_pos):
        _need_cuda(img)
        L = _lib.lib()
        B, _, R, _ = img.shape
        width, _, P, _ = conv_w.shape
        g = R // P
        G = g * g
        imgc = img.contiguous().float()
        patches = torch.empty((B * G, 3 * P * P), dtype=dtype, device=img.device)
        check(L.m3ae_patchify(_p(imgc), _p(patches), B, R, P, _dt(patches), _stream()), "m3ae_patchify")
        w2 = compute_weight(conv_w).view(width, -1)
        pe, _ = mm_nt(patches, patches.stride(0), B * G, w2)
        out = torch.empty((B, G + 1, width), dtype=dtype, device=img.device)
        posz = pos if add_pos else torch.zeros_like(pos)
        check(L.m3ae_vit_tokens_fwd(_p(pe), _p(cls), _p(posz), _p(out), B, G, width, _dt(out), _stream()),
              "m3ae_vit_tokens_fwd")
        ctx.save_for_backward(patches)
        ctx.p, ctx.dims, ctx.add_pos = (conv_w, cls, pos), (B, G, width), add_pos
        return out

    @staticmethod
    def backward(ctx, dout):
        (patches,) = ctx.saved_tensors
        conv_w, cls, pos = ctx.p
        B, G, width = ctx.dims
        d = dout.contiguous()
        dpe = torch.empty((B * G, width), dtype=d.dtype, device=d.device)
        gpos = _grad_buf(pos) if ctx.add_pos else torch.zeros_like(pos)
        check(_lib.lib().m3ae_vit_tokens_bwd(_p(d), _p(dpe), _p(_grad_buf(cls)), _p(gpos), B, G, width, _dt(d),
                                             _stream()), "m3ae_vit_tokens_bwd")
        _done(cls)
        if ctx.add_pos:
            _done(pos)
        if conv_w.requires_grad:
            g = _grad_buf(conv_w).view(width, -1)
            gemm(dpe, 1, dpe.stride(0), patches, patches.stride(0), 1, g, g.stride(0), width, g.shape[1], B * G,
                 accumulate=True)
            _done(conv_w)
        return None, None, None, None, None, None


def vit_tokens(img, conv_w, cls, pos, dtype, add_pos=True):
    return VitTokensFn.apply(img, conv_w, cls, pos, dtype, add_pos)


# ----------------------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------------------
class BCELossFn(torch.autograd.Function):
    """F.binary_cross_entropy_with_logits(x, z) * z.shape[1]  (objectives.py:201)."""

    @staticmethod
    def forward(ctx, logits, targets):
        x = logits.contiguous()
        B, Cc = x.shape
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        check(_lib.lib().m3ae_bce_logits(_p(x), _p(targets), _p(loss), _p(dx), B, Cc, 1.0, _dt(x), _stream()),
              "m3ae_bce_logits")
        ctx.save_for_backward(dx)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g.to(dx.dtype), None


def bce_with_logits_loss(logits, targets):
    return BCELossFn.apply(logits, targets)


class XentFn(torch.autograd.Function):
    """F.cross_entropy(logits, labels, ignore_index=-100) (objectives.py:19-23, :101)."""

    @staticmethod
    def forward(ctx, logits, labels):
        # rows may be strided (a [..., :V] view of vocabulary-padded logits, see VocabProjFn): no compaction copy
        x = logits
        Cc = x.shape[-1]
        uniform = x.dim() >= 2 and x.stride(-1) == 1 and all(
            x.stride(i) == x.stride(i + 1) * x.shape[i + 1] for i in range(x.dim() - 2))
        if not uniform:
            x = x.contiguous()
        ld = x.stride(-2) if x.dim() >= 2 else Cc
        rows = x.numel() // Cc
        lab = labels.contiguous().view(-1)
        loss = torch.zeros(1, dtype=torch.float32, device=x.device)
        ws = torch.empty(4, dtype=torch.float32, device=x.device)
        dx = (torch.zeros if ld != Cc else torch.empty)((rows, ld), dtype=x.dtype, device=x.device)
        check(_lib.lib().m3ae_xent(_p(x), _p(lab), _p(loss), _p(dx), _p(ws), rows, Cc, ld, 1.0, _dt(x), _stream()),
              "m3ae_xent")
        ctx.save_for_backward(dx)
        ctx.shape, ctx.cols = logits.shape, Cc
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        d = dx * g.to(dx.dtype)
        return d.view(*ctx.shape[:-1], dx.shape[-1])[..., :ctx.cols], None


def cross_entropy(logits, labels):
    return XentFn.apply(logits, labels)


class VocabProjFn(torch.autograd.Function):
    """logits = x . W^T + b for a vocabulary that is not a multiple of 128 (RoBERTa: 50265), perf mode.  The MFMA
    kernels need N % 4 == 0 (forward), K % 64 == 0 (dgrad) and N1 % 128 == 0 (wgrad); the odd size would send all three
    to the generic kernel (14 % of a pre-training step).  Zero-padded bf16 operand copies [Vp, K] / [K, Vp] (Vp = V rounded
    up to 128, refreshed from the weight's bf16 shadows on every call: two 77-MB copies) keep them on the MFMA path.
    Returns PADDED logits [..., Vp] (pad columns = 0); callers slice [..., :V] (ops.vocab_linear)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2, M, K, ldx = _rows(x)
        V = weight.shape[0]
        Vp = (V + 127) // 128 * 128
        st = getattr(weight, "_vocab_pad", None)
        if st is None:
            dev = x.device
            st = (torch.zeros((Vp, K), dtype=torch.bfloat16, device=dev), torch.zeros((K, Vp), dtype=torch.bfloat16, device=dev),
                  torch.zeros(Vp, dtype=torch.float32, device=dev))
            weight._vocab_pad = st
        Wp, WpT, bp = st
        Wp[:V].copy_(compute_weight(weight))
        WpT[:, :V].copy_(weight.m3ae_t)
        if bias is not None:
            bp[:V].copy_(bias.detach())
        y, _ = mm_nt(x2, ldx, M, Wp, bias=bp)
        ctx.save_for_backward(x2)
        ctx.meta = (weight, bias, M, K, V, Vp, ldx, x.shape)
        return y.view(*x.shape[:-1], Vp)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        weight, bias, M, K, V, Vp, ldx, xshape = ctx.meta
        _, WpT, _ = weight._vocab_pad
        dy2 = dy.contiguous().view(M, Vp)
        dx = torch.empty((M, K), dtype=dy2.dtype, device=dy2.device)
        gemm(dy2, Vp, 1, WpT, 1, Vp, dx, K, M, K, Vp)                       # dX = dY . W   (K-contiguous transposed copy)
        if weight.requires_grad:
            dWp = torch.zeros((Vp, K), dtype=torch.float32, device=dy2.device)
            dbp = torch.zeros(Vp, dtype=torch.float32, device=dy2.device)
            gemm(dy2, 1, Vp, x2, ldx, 1, dWp, K, Vp, K, M, accumulate=True, a_rowsum=dbp)   # dW = dY^T . X  (+ column sums)
            _grad_buf(weight).add_(dWp[:V])
            _done(weight)
            if bias is not None and bias.requires_grad:
                _grad_buf(bias).add_(dbp[:V])
                _done(bias)
        return dx.view(xshape), None, None


def vocab_linear(x, weight, bias):
    """Vocabulary projection (MLM head, prediction_heads.py:33): MFMA path for any vocabulary size in perf mode."""
    V = weight.shape[0]
    if x.dtype == torch.bfloat16 and V % 128 != 0 and getattr(weight, "m3ae_t", None) is not None:
        return VocabProjFn.apply(x, weight, bias)[..., :V]
    return linear(x, weight, bias)


# ----------------------------------------------------------------------------------------------------------
# row gather (MIM masking) -- differentiable in the source
# ----------------------------------------------------------------------------------------------------------
class GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx):
        idx = idx.contiguous()
        s2 = src.contiguous().view(-1, src.shape[-1])
        out = torch.empty((idx.numel(), s2.shape[1]), dtype=src.dtype, device=src.device)
        check(_lib.lib().m3ae_gather_rows(_p(s2), _p(idx), _p(out), idx.numel(), s2.shape[1], _dt(s2), _stream()),
              "m3ae_gather_rows")
        ctx.save_for_backward(idx)
        ctx.shape = src.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        d = dout.contiguous()
        dsrc = torch.zeros(ctx.shape, dtype=d.dtype, device=d.device)
        check(_lib.lib().m3ae_scatter_add_rows(_p(d), _p(idx), _p(dsrc), idx.numel(), d.shape[1], _dt(d), _stream()),
              "m3ae_scatter_add_rows")
        return dsrc, None


def gather_rows(src, flat_idx):
    return GatherRowsFn.apply(src, flat_idx)


class DropoutFn(torch.autograd.Function):
    """nn.Dropout as a standalone op (RoBERTa embeddings, HF RobertaEmbeddings.dropout; m3ae_module.py:230)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        xc = x.contiguous()
        y = torch.empty_like(xc)
        cols = xc.shape[-1]
        check(_lib.lib().m3ae_dropout(_p(xc), _p(y), None, xc.numel() // cols, cols, p, seed, _salt(), _dt(xc), _stream()),
              "m3ae_dropout")
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        d = dy.contiguous()
        dx = torch.empty_like(d)
        cols = d.shape[-1]
        check(_lib.lib().m3ae_dropout(_p(d), _p(dx), None, d.numel() // cols, cols, ctx.p, ctx.seed, _salt(), _dt(d), _stream()),
              "m3ae_dropout")
        return dx, None, None


def dropout(x, p, training=True):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, p, next_dropout_seed())


def dropout_keep_mask(rows, cols, p, seed, device="cuda"):
    """uint8 [rows, cols] keep-mask of the library's counter hash: the mask every dropout site applies for (p, seed)
    on a [rows, cols] array -- GEMM epilogue (M, N), LayerNorm backward (M, D), attention ((b*H + h)*Lq + q, Lk)."""
    m = torch.empty((rows, cols), dtype=torch.uint8, device=device)
    check(_lib.lib().m3ae_dropout(None, None, _p(m), rows, cols, p, seed, _salt(), F32, _stream()), "m3ae_dropout")
    return m


def selftest():
    out = torch.zeros(8 + 256, dtype=torch.int32, device="cuda")
    check(_lib.lib().m3ae_selftest(_p(out), _stream()), "m3ae_selftest")
    return out[:6].cpu().tolist()


# ------------------------------------------------------------------------------------------------------------
# masked-image-modelling bookkeeping (pre-training)
# ------------------------------------------------------------------------------------------------------------
def mask_ranks(noise, len_keep):
    """random_masking's index work (m3ae_module.py:153-183) -> ids_restore [B, L] int64, keep_rows [B * (len_keep + 1)]
    int64 (flat token-row ids, class row first), mask [B, L] fp32 (1 = removed)."""
    _need_cuda(noise)
    B, L = noise.shape
    n = noise.contiguous().float()
    ids_restore = torch.empty((B, L), dtype=torch.long, device=noise.device)
    keep_rows = torch.empty((B, len_keep + 1), dtype=torch.long, device=noise.device)
    mask = torch.empty((B, L), dtype=torch.float32, device=noise.device)
    check(_lib.lib().m3ae_mask_ranks(_p(n), _p(ids_restore), _p(keep_rows), _p(mask), B, L, len_keep, _stream()),
          "m3ae_mask_ranks")
    return ids_restore, keep_rows.view(-1), mask


def mim_targets(img, patch, norm_pix):
    """patchify (m3ae_module.py:185-192) [+ per-patch standardisation, objectives.py:52-56]: [B, C, H, W] -> [B, L, P*P*C]."""
    _need_cuda(img)
    B, Cc, H, W = img.shape
    x = img.contiguous().float()
    out = torch.empty((B, (H // patch) * (W // patch), patch * patch * Cc), dtype=torch.float32, device=img.device)
    check(_lib.lib().m3ae_mim_targets(_p(x), _p(out), B, Cc, H, W, patch, int(bool(norm_pix)), _stream()), "m3ae_mim_targets")
    return out


class MimLossFn(torch.autograd.Function):
    """objectives.py:58-62 on the decoder output WITH its class row (x [B, L + 1, D]): masked per-patch MSE."""

    @staticmethod
    def forward(ctx, x, target, mask):
        _need_cuda(x)
        xc = x.contiguous()
        B, L1, D = xc.shape
        t, m = target.contiguous().float(), mask.contiguous().float()
        acc = torch.empty(2, dtype=torch.float32, device=x.device)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        check(_lib.lib().m3ae_mim_loss_fwd(_p(xc), _p(t), _p(m), _p(acc), _p(loss), B, L1 - 1, D, _dt(xc), _stream()),
              "m3ae_mim_loss_fwd")
        ctx.save_for_backward(xc, t, m, acc)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        xc, t, m, acc = ctx.saved_tensors
        B, L1, D = xc.shape
        dx = torch.empty_like(xc)
        go = g.detach().reshape(1).float().contiguous()
        check(_lib.lib().m3ae_mim_loss_bwd(_p(xc), _p(t), _p(m), _p(acc), _p(go), _p(dx), B, L1 - 1, D, _dt(xc), _stream()),
              "m3ae_mim_loss_bwd")
        return dx, None, None


def mim_loss(x_with_cls, target, mask):
    return MimLossFn.apply(x_with_cls, target, mask)
