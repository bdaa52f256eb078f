"""CLIP ViT image tower on the MI355X kernels -- module tree / parameter names of the reference's
m3ae/modules/vision_encoders/clip_model.py:27-196 (weights of upstream M3AE / CLIP checkpoints load by name).

Pre-LN block = LayerNorm(fp32 stats) -> packed in-proj GEMM -> flash attention -> out-proj GEMM (+residual)
            -> LayerNorm -> GEMM(+bias+QuickGELU) -> GEMM(+bias+residual).
Activations stay [B, L, D] (the reference's NLD->LND permutes at clip_model.py:102-104 are layout only).
The tower runs `layers - 1` blocks (clip_model.py:71) -- the METER/M3AE quirk the checkpoints depend on.
"""
from types import SimpleNamespace as NS

import numpy as np
import torch
import torch.nn as nn

from .. import ops


class LayerNorm(nn.LayerNorm):
    """clip_model.py:27-33 (fp32 statistics; the kernel always computes them in fp32)."""


class QuickGELU(nn.Module):
    """clip_model.py:36-38; fused into the c_fc GEMM epilogue (ACT_QUICKGELU)."""


class _MHAParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names (clip_model.py:44): in_proj_weight [3d, d] (rows Q, K,
    V), in_proj_bias, out_proj.{weight,bias}."""

    def __init__(self, d_model, n_head):
        super().__init__()
        self.num_heads = n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.normal_(self.in_proj_weight, std=d_model ** -0.5)


class ResidualAttentionBlock(nn.Module):
    """clip_model.py:41-63."""

    def __init__(self, d_model, n_head):
        super().__init__()
        self.attn = _MHAParams(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d_model, d_model * 4))
        self.mlp.add_module("gelu", QuickGELU())
        self.mlp.add_module("c_proj", nn.Linear(d_model * 4, d_model))
        self.ln_2 = LayerNorm(d_model)
        self._bp = None

    def forward(self, x):
        if self._bp is None:
            a, m = self.attn, self.mlp
            self._bp = NS(heads=a.num_heads, ln1=self.ln_1, ln2=self.ln_2, w_in=a.in_proj_weight, b_in=a.in_proj_bias,
                          w_out=a.out_proj.weight, b_out=a.out_proj.bias, w_fc=m.c_fc.weight, b_fc=m.c_fc.bias,
                          w_proj=m.c_proj.weight, b_proj=m.c_proj.bias)
            self._anchors = tuple(self.parameters())
        return ops.ClipBlockFn.apply(x, self._bp, *self._anchors)

    def forward_unfused(self, x):
        h = ops.layer_norm(x, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps)
        qkv = ops.linear(h, self.attn.in_proj_weight, self.attn.in_proj_bias)
        ctx = ops.self_attention(qkv, None, self.attn.num_heads)
        x = ops.linear(ctx, self.attn.out_proj.weight, self.attn.out_proj.bias, residual=x)
        h = ops.layer_norm(x, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
        return ops.mlp(h, self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias,
                       ops.ACT_QUICKGELU, residual=x)

    def weight_units(self):
        return [self.attn.in_proj_weight, self.attn.out_proj.weight, self.mlp.c_fc.weight, self.mlp.c_proj.weight]


class Transformer(nn.Module):
    """clip_model.py:66-76: `layers - 1` blocks."""

    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers - 1)])

    def forward(self, x):
        for blk in self.resblocks:
            x = blk(x)
        return x

    def weight_units(self):
        u = []
        for blk in self.resblocks:
            u += blk.weight_units()
        return u


class VisualTransformer(nn.Module):
    """clip_model.py:79-128."""

    def __init__(self, patch_size, width, layers, heads, resolution_after):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((resolution_after // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)

    def forward_patch_embed(self, x, dtype):
        return ops.vit_tokens(x, self.conv1.weight, self.class_embedding, self.positional_embedding, dtype, add_pos=False)

    def forward_trans(self, x):
        x = ops.layer_norm(x, self.ln_pre.weight, self.ln_pre.bias, self.ln_pre.eps)
        x = self.transformer(x)
        return ops.layer_norm(x, self.ln_post.weight, self.ln_post.bias, self.ln_post.eps)

    def forward(self, x, dtype):
        x = ops.vit_tokens(x, self.conv1.weight, self.class_embedding, self.positional_embedding, dtype, add_pos=True)
        return self.forward_trans(x)

    def weight_units(self):
        return [self.conv1.weight] + self.transformer.weight_units()


class CLIP(nn.Module):
    """clip_model.py:131-196.  The text-tower leftovers (token_embedding, positional_embedding, ln_final) are kept so
    that state_dict keys match the reference; they never receive a gradient."""

    def __init__(self, vision_layers, vision_width, vision_patch_size, resolution_after, context_length=77,
                 vocab_size=49408, transformer_width=512):
        super().__init__()
        self.visual = VisualTransformer(vision_patch_size, vision_width, vision_layers, vision_width // 64,
                                        resolution_after)
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.initialize_parameters()

    def initialize_parameters(self):
        """clip_model.py:169-180."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        t = self.visual.transformer
        proj_std = (t.width ** -0.5) * ((2 * t.layers) ** -0.5)
        attn_std = t.width ** -0.5
        fc_std = (2 * t.width) ** -0.5
        for block in t.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)

    def forward(self, image, dtype):
        return self.visual(image, dtype)

    def forward_patch_embed(self, image, dtype):
        return self.visual.forward_patch_embed(image, dtype)

    def forward_trans(self, x):
        return self.visual.forward_trans(x)

    def weight_units(self):
        return self.visual.weight_units()


def adapt_position_encoding(model, patch_size=32, after=384, suffix="visual.positional_embedding"):
    """clip_model.py:224-251: bicubic resize of the patch grid of the positional embedding in a state_dict (host-side,
    load time only)."""
    keys = [k for k in model if k.endswith(suffix)]
    assert len(keys) == 1
    key = keys[0]
    origin = model[key]
    dim2 = origin.dim() == 2
    if dim2:
        origin = origin.unsqueeze(0)
    grid_before = int(np.sqrt(origin.shape[1] - 1))
    grid_after = after // patch_size
    assert after % patch_size == 0
    dim = origin.shape[-1]
    pe = origin[0, 1:, :].reshape(grid_before, grid_before, dim)
    pe = torch.nn.functional.interpolate(pe.permute(2, 0, 1).unsqueeze(0).float(), size=(grid_after, grid_after),
                                         mode="bicubic")
    pe = pe.squeeze(0).permute(1, 2, 0).reshape(-1, dim).to(origin.dtype)
    pe = torch.cat((origin[0, 0:1, :], pe), dim=0).unsqueeze(0)
    model[key] = pe.squeeze(0) if dim2 else pe
    return model


def build_model(name, resolution_after=224, vision_width=768, vision_layers=12, patch_size=16):
    """Counterpart of clip_model.py:259-313 without the URL download: the architecture is given by the config
    (`m3ae_amd.config.resolve_arch`); weights arrive through `load_state_dict` (checkpoint) or synth init."""
    return CLIP(vision_layers, vision_width, patch_size, resolution_after)
