"""Pooler / MLM / MIM / ITM heads -- names and arithmetic of the reference's m3ae/modules/prediction_heads.py."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .clip_model import LayerNorm, Transformer


class Pooler(nn.Module):
    """prediction_heads.py:9-19: tanh(Linear(h[:, 0])); the token-0 gather is folded into the GEMM's row stride."""

    def __init__(self, hidden_size):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)

    def forward(self, hidden_states):
        return ops.GatherLinearFn.apply(hidden_states, self.dense.weight, self.dense.bias, ops.ACT_TANH)

    def weight_units(self):
        return [self.dense.weight]


class _HeadTransform(nn.Module):
    """transformers BertPredictionHeadTransform (prediction_heads.py:25): dense -> gelu -> LayerNorm."""

    def __init__(self, hidden, eps):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)


class MLMHead(nn.Module):
    """prediction_heads.py:22-34."""

    def __init__(self, hidden, vocab, eps=1e-12):
        super().__init__()
        self.transform = _HeadTransform(hidden, eps)
        self.decoder = nn.Linear(hidden, vocab, bias=False)
        self.bias = nn.Parameter(torch.zeros(vocab))

    def forward(self, x):
        t = self.transform
        h = ops.linear(x, t.dense.weight, t.dense.bias, act=ops.ACT_GELU)
        h = ops.layer_norm(h, t.LayerNorm.weight, t.LayerNorm.bias, t.LayerNorm.eps)
        return ops.vocab_linear(h, self.decoder.weight, self.bias)

    def weight_units(self):
        return [self.transform.dense.weight, self.decoder.weight]


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """position_embeddings.py:21-69 (MAE 2-D sin-cos table; a constant computed once on the host)."""
    def one_d(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float64) / (dim / 2.0)
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size, grid_size])
    emb = np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


class MIMHead(nn.Module):
    """prediction_heads.py:37-86 (MAE-style decoder; `decoder` runs decoder_num_layers blocks, :57 + clip :71)."""

    def __init__(self, config):
        super().__init__()
        self.hidden_size = config["hidden_size"]
        self.patch_size = config["patch_size"]
        self.num_patches = (config["image_size"] // config["patch_size"]) ** 2
        dh = config["mim_decoder_hidden_size"]
        self.decoder_embed = nn.Linear(self.hidden_size, dh, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, dh))
        torch.nn.init.normal_(self.mask_token, std=.02)
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, dh), requires_grad=False)
        pe = get_2d_sincos_pos_embed(dh, int(self.num_patches ** .5), True)
        self.decoder_pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        self.decoder = Transformer(dh, config["mim_decoder_num_layers"] + 1, config["mim_decoder_num_heads"])
        self.decoder_norm = LayerNorm(dh)
        self.decoder_pred = nn.Linear(dh, self.patch_size ** 2 * 3, bias=True)

    def forward(self, x, ids_restore, keep_cls=False):
        B, Lk, _ = x.shape
        x = ops.linear(x, self.decoder_embed.weight, self.decoder_embed.bias)
        dh = x.shape[-1]
        L = ids_restore.shape[1]
        n_mask = L + 1 - Lk
        # rows of [x[:, 1:], mask tokens] un-shuffled by ids_restore (prediction_heads.py:65-68), then cls prepended
        src = torch.cat([x[:, 1:, :], self.mask_token.to(x.dtype).expand(B, n_mask, dh)], dim=1)  # [B, L, dh]
        flat_idx = (ids_restore + torch.arange(B, device=x.device).view(B, 1) * L).reshape(-1)
        x_ = ops.gather_rows(src, flat_idx).view(B, L, dh)
        x = torch.cat([x[:, :1, :], x_], dim=1) + self.decoder_pos_embed.to(x.dtype)
        x = self.decoder(x.contiguous())
        x = ops.layer_norm(x, self.decoder_norm.weight, self.decoder_norm.bias, self.decoder_norm.eps)
        x = ops.linear(x, self.decoder_pred.weight, self.decoder_pred.bias)
        return x if keep_cls else x[:, 1:, :]   # prediction_heads.py:86 drops the class row; the loss kernel skips it itself

    def weight_units(self):
        return [self.decoder_embed.weight, self.decoder_pred.weight] + self.decoder.weight_units()


class ITMHead(nn.Module):
    """prediction_heads.py:89-96."""

    def __init__(self, hidden_size):
        super().__init__()
        self.fc = nn.Linear(hidden_size, 2)

    def forward(self, x):
        return ops.linear(x, self.fc.weight, self.fc.bias)

    def weight_units(self):
        return [self.fc.weight]
