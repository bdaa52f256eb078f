"""BertCrossLayer / BertSelfLayer on the MI355X kernels -- same module tree and parameter names as the reference's
m3ae/modules/language_encoders/bert_model.py:211-546 (and HF RobertaLayer, whose math and names are identical).

Each block is three fused pieces instead of the reference's ~20 ATen calls:
  packed Q|K|V (or Q and K|V) projection GEMM -> flash attention -> output GEMM with the residual add in its
  epilogue -> LayerNorm;   FFN = GEMM(+bias+erf-GELU, pre-activation kept) -> GEMM(+bias+residual) -> LayerNorm.
Attention probabilities are never materialised (the reference hard-wires output_attentions=True,
m3ae_module.py:276-277).  Dropout: see DESIGN.md (parity runs are eval-mode).
"""
from types import SimpleNamespace as NS

import torch
import torch.nn as nn

from .. import ops
from ..param_store import PackedParam


class BertSelfAttention(nn.Module):
    """bert_model.py:211-350.  `query/key/value` stay separate nn.Linear parameters (state_dict names); the
    kernels read them as one packed projection through PackedParam views of the flat parameter buffer."""

    def __init__(self, hidden, heads):
        super().__init__()
        self.num_attention_heads = heads
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(hidden, hidden)
        self.value = nn.Linear(hidden, hidden)
        self._packs = {}

    def pack(self, kind):
        if kind not in self._packs:
            mods = {"qkv": (self.query, self.key, self.value), "kv": (self.key, self.value)}[kind]
            self._packs[kind] = (PackedParam([m.weight for m in mods]), PackedParam([m.bias for m in mods]))
        return self._packs[kind]

    def weight_units(self, cross):
        if cross:
            return [self.query.weight, self.pack("kv")[0]]
        return [self.pack("qkv")[0]]


class BertSelfOutput(nn.Module):
    """bert_model.py:353-364."""

    def __init__(self, hidden, eps):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)


class BertAttention(nn.Module):
    """bert_model.py:367-413: self- or cross-attention + output dense + residual + LayerNorm."""

    def __init__(self, hidden, heads, eps, cross=False):
        super().__init__()
        self.self = BertSelfAttention(hidden, heads)
        self.output = BertSelfOutput(hidden, eps)
        self.cross = cross

    def forward(self, h, key_mask=None, other=None, other_mask=None, pdrop=0.0):
        """Op-level composition (A/B reference of the fused blocks).  Draws its dropout seeds in the same order as
        ops._attn_sub_fwd, so fused and unfused runs started from one `ops.set_dropout_seed` use identical masks."""
        heads = self.self.num_attention_heads
        if other is not None and not torch.is_grad_enabled():
            # inference: the fused sub-block (csrc/xattn.hip) when the shapes are covered
            B, L, D = h.shape
            Lo = other.shape[1]
            P = self.block_params()
            h2, o2 = h.contiguous().view(B * L, D), other.contiguous().view(B * Lo, other.shape[2])
            if ops.xattn_supported(h2, L, o2, Lo, other_mask, P):
                return ops.xattn_fwd(h2, B, L, o2, Lo, other_mask, P, pdrop, need_bwd=False)[0].view(B, L, D)
        da = (pdrop, ops.next_dropout_seed()) if pdrop > 0 else None
        dh = (pdrop, ops.next_dropout_seed()) if pdrop > 0 else None
        if other is None:
            w, b = self.self.pack("qkv")
            qkv = ops.linear(h, w, b)
            ctx = ops.self_attention(qkv, key_mask, heads, da)
        else:
            q = ops.linear(h, self.self.query.weight, self.self.query.bias)
            w, b = self.self.pack("kv")
            kv = ops.linear(other, w, b)
            ctx = ops.cross_attention(q, kv, other_mask, heads, da)
        if dh is None:
            s = ops.linear(ctx, self.output.dense.weight, self.output.dense.bias, residual=h)
        else:
            s = ops.DropoutFn.apply(ops.linear(ctx, self.output.dense.weight, self.output.dense.bias), *dh) + h
        ln = self.output.LayerNorm
        return ops.layer_norm(s, ln.weight, ln.bias, ln.eps)

    def weight_units(self):
        return self.self.weight_units(self.cross) + [self.output.dense.weight]

    def block_params(self):
        """Parameter references in the form the fused block functions (ops.BertCrossLayerFn) consume."""
        sa, out = self.self, self.output
        if self.cross:
            w_kv, b_kv = sa.pack("kv")
            return NS(heads=sa.num_attention_heads, w_q=sa.query.weight, b_q=sa.query.bias, w_kv=w_kv, b_kv=b_kv,
                      w_o=out.dense.weight, b_o=out.dense.bias, ln=out.LayerNorm)
        w_qkv, b_qkv = sa.pack("qkv")
        return NS(heads=sa.num_attention_heads, w_qkv=w_qkv, b_qkv=b_qkv, w_o=out.dense.weight, b_o=out.dense.bias,
                  ln=out.LayerNorm)


def _ffn_params(layer):
    return NS(w1=layer.intermediate.dense.weight, b1=layer.intermediate.dense.bias, w2=layer.output.dense.weight,
              b2=layer.output.dense.bias, ln=layer.output.LayerNorm)


class BertIntermediate(nn.Module):
    def __init__(self, hidden, inter):
        super().__init__()
        self.dense = nn.Linear(hidden, inter)


class BertOutput(nn.Module):
    def __init__(self, hidden, inter, eps):
        super().__init__()
        self.dense = nn.Linear(inter, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)


def _ffn(layer, h, pdrop=0.0):
    """feed_forward_chunk (bert_model.py:500-503)."""
    if pdrop > 0:
        seed = ops.next_dropout_seed()
        s = ops.mlp(h, layer.intermediate.dense.weight, layer.intermediate.dense.bias, layer.output.dense.weight,
                    layer.output.dense.bias, ops.ACT_GELU)
        s = ops.DropoutFn.apply(s, pdrop, seed) + h
    else:
        s = ops.mlp(h, layer.intermediate.dense.weight, layer.intermediate.dense.bias, layer.output.dense.weight,
                    layer.output.dense.bias, ops.ACT_GELU, residual=h)
    ln = layer.output.LayerNorm
    return ops.layer_norm(s, ln.weight, ln.bias, ln.eps)


class BertCrossLayer(nn.Module):
    """bert_model.py:445-503: self-attention -> cross-attention (residual = self-attention output) -> FFN."""

    def __init__(self, hidden, heads, inter, eps=1e-12, drop_rate=0.0):
        super().__init__()
        self.drop_rate = drop_rate  # hidden_dropout_prob = attention_probs_dropout_prob (m3ae_module.py:31-32)
        self.attention = BertAttention(hidden, heads, eps)
        self.crossattention = BertAttention(hidden, heads, eps, cross=True)
        self.intermediate = BertIntermediate(hidden, inter)
        self.output = BertOutput(hidden, inter, eps)
        self._bp = None

    def forward(self, hidden_states, encoder_hidden_states, attention_mask=None, encoder_attention_mask=None):
        if self._bp is None:
            self._bp = NS(attn=self.attention.block_params(), cross=self.crossattention.block_params(),
                          ffn=_ffn_params(self))
            self._anchors = tuple(self.parameters())
        self._bp.pdrop = self.drop_rate if self.training else 0.0
        # forward-only calls always take the fused cross-attention sub-block; training takes it with its fused backward
        self._bp.fused_cross = (not torch.is_grad_enabled()) or (ops.XATTN_TRAIN != "off" and
                                                                 hidden_states.shape[0] >= ops.XATTN_TRAIN_MIN_BATCH)
        self._bp.cross.need_bwd = torch.is_grad_enabled()
        return ops.BertCrossLayerFn.apply(hidden_states, encoder_hidden_states, attention_mask, encoder_attention_mask,
                                          self._bp, *self._anchors)

    def forward_unfused(self, hidden_states, encoder_hidden_states, attention_mask=None, encoder_attention_mask=None):
        """Op-level composition (one autograd node per kernel group); kept for A/B checks against the fused node."""
        pd = self.drop_rate if self.training else 0.0
        a = self.attention(hidden_states, attention_mask, pdrop=pd)
        c = self.crossattention(a, None, encoder_hidden_states, encoder_attention_mask, pdrop=pd)
        return _ffn(self, c, pd)

    def weight_units(self):
        return (self.attention.weight_units() + self.crossattention.weight_units()
                + [self.intermediate.dense.weight, self.output.dense.weight])


class BertSelfLayer(nn.Module):
    """bert_model.py:506-546 == HF RobertaLayer (called at m3ae_module.py:233-234)."""

    def __init__(self, hidden, heads, inter, eps=1e-5, drop_rate=0.0):
        super().__init__()
        self.drop_rate = drop_rate
        self.attention = BertAttention(hidden, heads, eps)
        self.intermediate = BertIntermediate(hidden, inter)
        self.output = BertOutput(hidden, inter, eps)
        self._bp = None

    def forward(self, hidden_states, attention_mask=None):
        if self._bp is None:
            self._bp = NS(attn=self.attention.block_params(), ffn=_ffn_params(self))
            self._anchors = tuple(self.parameters())
        self._bp.pdrop = self.drop_rate if self.training else 0.0
        return ops.BertSelfLayerFn.apply(hidden_states, attention_mask, self._bp, *self._anchors)

    def forward_unfused(self, hidden_states, attention_mask=None):
        pd = self.drop_rate if self.training else 0.0
        return _ffn(self, self.attention(hidden_states, attention_mask, pdrop=pd), pd)

    def weight_units(self):
        return self.attention.weight_units() + [self.intermediate.dense.weight, self.output.dense.weight]


class RobertaEmbeddings(nn.Module):
    """HF RobertaEmbeddings (third party; m3ae_module.py:230)."""

    def __init__(self, vocab, hidden, max_pos, type_vocab=1, pad_id=1, eps=1e-5, drop_rate=0.0):
        super().__init__()
        self.drop_rate = drop_rate
        self.word_embeddings = nn.Embedding(vocab, hidden, padding_idx=pad_id)
        self.position_embeddings = nn.Embedding(max_pos, hidden, padding_idx=pad_id)
        self.token_type_embeddings = nn.Embedding(type_vocab, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)
        self.padding_idx = pad_id

    def forward(self, input_ids, dtype):
        e = ops.roberta_embed(input_ids, self.word_embeddings.weight, self.position_embeddings.weight,
                              self.token_type_embeddings.weight, self.padding_idx, dtype)
        y = ops.layer_norm(e, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)
        return ops.dropout(y, self.drop_rate, self.training)


class RobertaEncoder(nn.Module):
    def __init__(self, layers, hidden, heads, inter, drop_rate=0.0):
        super().__init__()
        self.layer = nn.ModuleList([BertSelfLayer(hidden, heads, inter, eps=1e-5, drop_rate=drop_rate)
                                    for _ in range(layers)])


class RobertaPooler(nn.Module):
    """Present in the reference's state_dict, never used by infer."""

    def __init__(self, hidden):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)


class RobertaModel(nn.Module):
    """Parameter-name-compatible stand-in for transformers' RobertaModel as the reference uses it
    (m3ae_module.py:66,230-234): embeddings + encoder.layer[*]; `pooler` kept for key compatibility."""

    def __init__(self, vocab, hidden, layers, heads, inter, max_pos=514, drop_rate=0.1):
        super().__init__()
        # roberta-base ships hidden_dropout_prob = attention_probs_dropout_prob = 0.1 (RobertaModel.from_pretrained)
        self.embeddings = RobertaEmbeddings(vocab, hidden, max_pos, drop_rate=drop_rate)
        self.encoder = RobertaEncoder(layers, hidden, heads, inter, drop_rate=drop_rate)
        self.pooler = RobertaPooler(hidden)

    @staticmethod
    def get_extended_attention_mask(mask):
        """transformers-4.6.0 semantics (m3ae_module.py:232): (1 - mask) * -10000.0, here kept as [B, L] fp32 --
        the kernels broadcast it over heads and queries."""
        return (1.0 - mask.to(torch.float32)) * -10000.0

    def weight_units(self):
        u = []
        for l in self.encoder.layer:
            u += l.weight_units()
        return u
