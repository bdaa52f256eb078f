"""T5 encoder-decoder for the generative answer head on the MI355X kernels.

The reference uses HF `T5ForConditionalGeneration` (third party, transformers==4.6.0) through
m3ae/modules/m3ae_t5_mm_encoder_input.py:26-27,202,244.  This module keeps HF's parameter names
(`shared.weight`, `encoder.block.{i}.layer.0.SelfAttention.{q,k,v,o}.weight`, `...relative_attention_bias.weight`,
`layer.{j}.layer_norm.weight`, `DenseReluDense.{wi,wo}.weight`, `decoder.block.{i}.layer.1.EncDecAttention...`,
`final_layer_norm.weight`) so `t5.*` checkpoints load by name, and runs each pre-norm block as one fused autograd
node (ops.T5EncBlockFn / ops.T5DecBlockFn): RMSNorm -> packed QKV GEMM -> attention with the additive relative
position bias (no 1/sqrt(d) scaling) -> output GEMM (+residual) -> RMSNorm -> GEMM(+ReLU) -> GEMM(+residual).
The LM head is tied to `shared` and the decoder output is scaled by d_model^-0.5 (T5 v1.0 / t5-small, t5-base).
"""
import math
from types import SimpleNamespace as NS

import torch
import torch.nn as nn

from .. import ops
from ..param_store import PackedParam

T5_ARCH = {
    "t5-small": dict(d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_decoder_layers=6, num_heads=8),
    "t5-base": dict(d_model=768, d_kv=64, d_ff=3072, num_layers=12, num_decoder_layers=12, num_heads=12),
    "t5-large": dict(d_model=1024, d_kv=64, d_ff=4096, num_layers=24, num_decoder_layers=24, num_heads=16),
}


class T5LayerNorm(nn.Module):
    def __init__(self, d, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = None
        self.eps = eps


class T5Attention(nn.Module):
    def __init__(self, d_model, heads, d_kv, has_bias, is_decoder, buckets=32, max_distance=128):
        super().__init__()
        inner = heads * d_kv
        self.heads, self.is_decoder, self.buckets, self.max_distance = heads, is_decoder, buckets, max_distance
        self.q = nn.Linear(d_model, inner, bias=False)
        self.k = nn.Linear(d_model, inner, bias=False)
        self.v = nn.Linear(d_model, inner, bias=False)
        self.o = nn.Linear(inner, d_model, bias=False)
        if has_bias:
            self.relative_attention_bias = nn.Embedding(buckets, heads)
        self._packs = {}

    def pack(self, kind):
        if kind not in self._packs:
            mods = {"qkv": (self.q, self.k, self.v), "kv": (self.k, self.v)}[kind]
            self._packs[kind] = PackedParam([m.weight for m in mods])
        return self._packs[kind]

    def bucket(self, Lq, Lk, device):
        """T5Attention._relative_position_bucket (integer glue, host-side shapes)."""
        ctx = torch.arange(Lq, device=device)[:, None]
        mem = torch.arange(Lk, device=device)[None, :]
        rel = mem - ctx
        nb = self.buckets
        ret = torch.zeros_like(rel)
        if not self.is_decoder:
            nb //= 2
            ret = ret + (rel > 0).long() * nb
            rel = rel.abs()
        else:
            rel = -torch.min(rel, torch.zeros_like(rel))
        max_exact = nb // 2
        is_small = rel < max_exact
        large = max_exact + (torch.log(rel.float() / max_exact) / math.log(self.max_distance / max_exact)
                             * (nb - max_exact)).long()
        large = torch.min(large, torch.full_like(large, nb - 1))
        return ret + torch.where(is_small, rel, large)

    def position_bias(self, Lq, Lk):
        """compute_bias: fp32 [H, Lq, Lk]; differentiable in the (tiny) bucket table through autograd glue."""
        w = self.relative_attention_bias.weight
        return w[self.bucket(Lq, Lk, w.device)].permute(2, 0, 1).contiguous().float()


class _SelfAttnLayer(nn.Module):
    def __init__(self, d_model, heads, d_kv, has_bias, is_decoder):
        super().__init__()
        self.SelfAttention = T5Attention(d_model, heads, d_kv, has_bias, is_decoder)
        self.layer_norm = T5LayerNorm(d_model)


class _CrossAttnLayer(nn.Module):
    def __init__(self, d_model, heads, d_kv):
        super().__init__()
        self.EncDecAttention = T5Attention(d_model, heads, d_kv, False, True)
        self.layer_norm = T5LayerNorm(d_model)


class _DenseReluDense(nn.Module):
    def __init__(self, d_model, d_ff):
        super().__init__()
        self.wi = nn.Linear(d_model, d_ff, bias=False)
        self.wo = nn.Linear(d_ff, d_model, bias=False)


class _FFLayer(nn.Module):
    def __init__(self, d_model, d_ff):
        super().__init__()
        self.DenseReluDense = _DenseReluDense(d_model, d_ff)
        self.layer_norm = T5LayerNorm(d_model)


def _attn_params(att, ln, cross):
    if cross:
        return NS(heads=att.heads, w_q=att.q.weight, w_kv=att.pack("kv"), w_o=att.o.weight, ln=ln)
    return NS(heads=att.heads, w_qkv=att.pack("qkv"), w_o=att.o.weight, ln=ln)


class T5Block(nn.Module):
    def __init__(self, d_model, heads, d_kv, d_ff, has_bias, is_decoder, dropout_rate=0.1):
        super().__init__()
        self.is_decoder = is_decoder
        self.dropout_rate = dropout_rate  # T5Config.dropout_rate (0.1 for t5-small / t5-base)
        layers = [_SelfAttnLayer(d_model, heads, d_kv, has_bias, is_decoder)]
        if is_decoder:
            layers.append(_CrossAttnLayer(d_model, heads, d_kv))
        layers.append(_FFLayer(d_model, d_ff))
        self.layer = nn.ModuleList(layers)
        self._bp = None

    def params(self):
        if self._bp is None:
            sa = self.layer[0]
            ff = self.layer[-1]
            self._bp = NS(attn=_attn_params(sa.SelfAttention, sa.layer_norm, False),
                          ffn=NS(w1=ff.DenseReluDense.wi.weight, w2=ff.DenseReluDense.wo.weight, ln=ff.layer_norm))
            if self.is_decoder:
                ca = self.layer[1]
                self._bp.cross = _attn_params(ca.EncDecAttention, ca.layer_norm, True)
            self._anchors = tuple(self.parameters())
        return self._bp

    def forward(self, h, pos_bias, enc=None):
        P = self.params()
        P.pdrop = self.dropout_rate if self.training else 0.0
        if self.is_decoder:
            return ops.T5DecBlockFn.apply(h, enc, pos_bias, P, *self._anchors)
        return ops.T5EncBlockFn.apply(h, pos_bias, P, *self._anchors)

    def weight_units(self):
        sa = self.layer[0].SelfAttention
        ff = self.layer[-1].DenseReluDense
        u = [sa.pack("qkv"), sa.o.weight, ff.wi.weight, ff.wo.weight]
        if self.is_decoder:
            ca = self.layer[1].EncDecAttention
            u += [ca.q.weight, ca.pack("kv"), ca.o.weight]
        return u


class T5Stack(nn.Module):
    def __init__(self, n_layers, d_model, heads, d_kv, d_ff, is_decoder, dropout_rate=0.1):
        super().__init__()
        self.is_decoder = is_decoder
        self.dropout_rate = dropout_rate
        self.block = nn.ModuleList([T5Block(d_model, heads, d_kv, d_ff, i == 0, is_decoder, dropout_rate)
                                    for i in range(n_layers)])
        self.final_layer_norm = T5LayerNorm(d_model)

    def forward(self, h, enc=None):
        """HF T5Stack.forward: dropout(inputs_embeds) -> blocks -> dropout(final_layer_norm(h)); every dropout site is
        active in train() mode (frozen blocks included, as in the reference: freezing only clears requires_grad)."""
        L = h.shape[1]
        h = ops.dropout(h, self.dropout_rate, self.training)
        bias = self.block[0].layer[0].SelfAttention.position_bias(L, L)
        for blk in self.block:
            h = blk(h, bias, enc)
        ln = self.final_layer_norm
        return ops.dropout(ops.layer_norm(h, ln.weight, None, ln.eps, rms=True), self.dropout_rate, self.training)

    @torch.no_grad()
    def cross_kv(self, enc):
        """Decoder only: keys / values of the encoder output for every block's cross-attention ([B * Ls, 2 inner] each),
        computed ONCE per generation instead of at every decoding step (the reference's third-party `generate` keeps them
        in `past_key_values`)."""
        B, Ls, De = enc.shape
        enc2 = enc.contiguous().view(B * Ls, De)
        return [ops.mm_nt(enc2, De, B * Ls, ops.compute_weight(blk.params().cross.w_kv))[0] for blk in self.block]

    @torch.no_grad()
    def forward_cached(self, h, cross_kv, Ls):
        """Decoder only, inference: the blocks' own kernels in the blocks' own order (ops.T5DecBlockFn.forward), with the
        cross-attention keys / values taken from `cross_kv`."""
        B, T, D = h.shape
        pd = self.dropout_rate if self.training else 0.0
        h = ops.dropout(h, self.dropout_rate, self.training)
        bias = self.block[0].layer[0].SelfAttention.position_bias(T, T).detach()
        h2 = h.contiguous().view(B * T, D)
        for blk, kv in zip(self.block, cross_kv):
            P = blk.params()
            a, _ = ops._t5_attn_fwd(h2, B, T, None, T, P.attn, bias, True, pd)
            c, _ = ops._t5_attn_fwd(a, B, T, None, Ls, P.cross, None, False, pd, kv=kv)
            h2, _ = ops._t5_ff_fwd(c, P.ffn, pd)
        ln = self.final_layer_norm
        return ops.dropout(ops.layer_norm(h2.view(B, T, D), ln.weight, None, ln.eps, rms=True), self.dropout_rate, self.training)

    @torch.no_grad()
    def new_self_cache(self, B, max_len, dtype, device):
        """Decoder only: per block a [B, max_len, 2 inner] buffer of the prefix's self-attention keys | values
        (HF keeps them in `past_key_values`; m3ae_t5_mm_encoder_input.py:209-227 decodes through them)."""
        inner2 = 2 * self.block[0].params().attn.w_o.shape[1]
        return [torch.empty((B, max_len, inner2), dtype=dtype, device=device) for _ in self.block]

    @torch.no_grad()
    def step_cached(self, h_new, cross_kv, Ls, self_cache, t):
        """Decoder only, inference: ONE new position t (h_new [B, 1, D]) through the blocks' own kernels; its key / value row
        is appended to `self_cache`, the attention of the new query runs over the cached rows 0..t (exactly the keys the
        causal mask leaves to the last row of a full-prefix pass), the encoder side comes from `cross_kv`."""
        B, _, D = h_new.shape
        pd = self.dropout_rate if self.training else 0.0
        h = ops.dropout(h_new, self.dropout_rate, self.training)
        sa = self.block[0].layer[0].SelfAttention
        bias = sa.position_bias(t + 1, t + 1).detach()[:, t:t + 1, :].contiguous()   # the last query's row [H, 1, t + 1]
        h2 = h.contiguous().view(B, D)
        for blk, kv, cache in zip(self.block, cross_kv, self_cache):
            P = blk.params()
            a = ops.t5_self_attn_step(h2, B, P.attn, bias, cache, t, pd)
            c, _ = ops._t5_attn_fwd(a, B, 1, None, Ls, P.cross, None, False, pd, kv=kv)
            h2, _ = ops._t5_ff_fwd(c, P.ffn, pd)
        ln = self.final_layer_norm
        return ops.dropout(ops.layer_norm(h2.view(B, 1, D), ln.weight, None, ln.eps, rms=True), self.dropout_rate, self.training)

    def weight_units(self):
        u = []
        for b in self.block:
            u += b.weight_units()
        return u


class T5ForConditionalGeneration(nn.Module):
    """Encoder + teacher-forced decoder + tied LM head + cross-entropy (HF T5ForConditionalGeneration.forward with
    `encoder_outputs` / `inputs_embeds` and `labels`), and beam-search `generate`."""

    def __init__(self, name_or_dims="t5-small", vocab_size=32128):
        super().__init__()
        dims = T5_ARCH[name_or_dims] if isinstance(name_or_dims, str) else dict(name_or_dims)
        self.config = NS(hidden_size=dims["d_model"], d_model=dims["d_model"], vocab_size=vocab_size,
                         num_heads=dims["num_heads"], pad_token_id=0, decoder_start_token_id=0, eos_token_id=1)
        d = dims["d_model"]
        self.shared = nn.Embedding(vocab_size, d)
        dr = dims.get("dropout_rate", 0.1)
        self.encoder = T5Stack(dims["num_layers"], d, dims["num_heads"], dims["d_kv"], dims["d_ff"], False, dr)
        self.decoder = T5Stack(dims["num_decoder_layers"], d, dims["num_heads"], dims["d_kv"], dims["d_ff"], True, dr)

    def weight_units(self):
        return [self.shared.weight] + self.encoder.weight_units() + self.decoder.weight_units()

    def shift_right(self, labels):
        out = torch.zeros_like(labels)
        out[:, 1:] = labels[:, :-1]
        out[:, 0] = self.config.decoder_start_token_id
        return out.masked_fill(out == -100, self.config.pad_token_id)

    def embed(self, ids, dtype):
        w = self.shared.weight
        table = ops.compute_weight(w) if dtype == torch.bfloat16 else w
        B, T = ids.shape
        return ops.EmbedRowsFn.apply(ids.reshape(-1).contiguous(), w, table).view(B, T, -1)   # (a [B, 1] slice reshapes to a STRIDED view)

    @torch.no_grad()
    def next_token_logits(self, enc, prefix, cross_kv=None):
        """Decoder over the whole prefix (T <= 12 here: the self-attention keys are recomputed, the 512-token encoder
        side comes from `cross_kv` when given) -> logits of the last position, fp32.  (`generate` uses
        `next_token_logits_cached`; this full-prefix form is what it is tested against.)"""
        if cross_kv is not None:
            dec = self.decoder.forward_cached(self.embed(prefix, enc.dtype), cross_kv, enc.shape[1])
        else:
            dec = self.decoder(self.embed(prefix, enc.dtype), enc)
        last = dec[:, -1].contiguous()
        return ops.linear(last, self.shared.weight, None, alpha=self.config.d_model ** -0.5).float()

    @torch.no_grad()
    def next_token_logits_cached(self, enc, last_ids, cross_kv, self_cache, t):
        """The same logits from the NEW token alone (last_ids [B, 1] at position t): the prefix's self-attention keys /
        values come from `self_cache` (HF `past_key_values`), so a step costs one decoder row instead of t + 1."""
        dec = self.decoder.step_cached(self.embed(last_ids, enc.dtype), cross_kv, enc.shape[1], self_cache, t)
        return ops.linear(dec[:, -1].contiguous(), self.shared.weight, None, alpha=self.config.d_model ** -0.5).float()

    @torch.no_grad()
    def generate(self, enc, num_beams=4, max_length=12, eos_token_id=1, pad_token_id=0, length_penalty=1.0,
                 len_offset=0):
        """HF `generate(encoder_outputs=..., num_beams, early_stopping=True, max_length)` as the reference calls it
        (m3ae_t5_mm_encoder_input.py:209-218,252-260): transformers-4.6.0 `beam_search` + `BeamSearchScorer`
        semantics, restated in oracle/m3ae_oracle.py::t5_beam_search (pinned against the installed release's
        `generate`; `len_offset` documents the one convention that changed between the two releases).
        The model runs on the GPU; the per-step candidate bookkeeping (2 * beams scores and tokens per sample) is
        host logic on one small device->host copy per step."""
        B, nb, dev = enc.shape[0], num_beams, enc.device
        enc_r = enc.repeat_interleave(nb, dim=0).contiguous()
        # encoder-side keys / values once per call; the beams of a sample share their encoder rows, so re-ordering beams
        # never touches this cache
        Ls = enc.shape[1]
        cross_kv = [kv.view(B, Ls, -1).repeat_interleave(nb, dim=0).reshape(B * nb * Ls, -1) for kv in self.decoder.cross_kv(enc)]
        ids = torch.full((B * nb, 1), self.config.decoder_start_token_id, dtype=torch.long, device=dev)
        # the beams' own keys / values: one decoder row per step; re-ordered with the beams (HF `_reorder_cache`)
        self_cache = self.decoder.new_self_cache(B * nb, max_length, enc.dtype, dev)
        beam_scores = torch.zeros(B, nb, device=dev)
        beam_scores[:, 1:] = -1e9
        beam_scores = beam_scores.view(-1)
        hyps = [[] for _ in range(B)]
        done = [False] * B
        cur_len = 1
        while cur_len < max_length:
            logp = torch.log_softmax(self.next_token_logits_cached(enc_r, ids[:, -1:], cross_kv, self_cache, cur_len - 1), dim=-1)
            V = logp.shape[-1]
            top_s, top_i = torch.topk((logp + beam_scores[:, None]).view(B, nb * V), 2 * nb, dim=1)
            top_s, top_i = top_s.cpu(), top_i.cpu()
            ids_h = ids.cpu()
            nxt_scores = torch.zeros(B, nb)
            nxt_tokens = torch.full((B, nb), pad_token_id, dtype=torch.long)
            nxt_index = torch.zeros(B, nb, dtype=torch.long)
            for b in range(B):
                if done[b]:
                    nxt_index[b] = b * nb
                    continue
                k = 0
                for rank in range(2 * nb):
                    tok, sc = int(top_i[b, rank] % V), float(top_s[b, rank])
                    src = b * nb + int(top_i[b, rank] // V)
                    if tok == eos_token_id:
                        if rank >= nb:
                            continue
                        hyp = ids_h[src].clone()
                        hyps[b].append((sc / (hyp.shape[-1] ** length_penalty), hyp))
                        hyps[b] = sorted(hyps[b], key=lambda t: -t[0])[:nb]
                    else:
                        nxt_scores[b, k], nxt_tokens[b, k], nxt_index[b, k] = sc, tok, src
                        k += 1
                    if k == nb:
                        break
                done[b] = done[b] or len(hyps[b]) >= nb
            beam_scores = nxt_scores.view(-1).to(dev)
            order = nxt_index.view(-1).to(dev)
            ids = torch.cat([ids[order], nxt_tokens.view(-1, 1).to(dev)], dim=1)
            self_cache = [ops.gather_rows(c.view(B * nb, -1), order).view_as(c) for c in self_cache]
            cur_len += 1
            if all(done):
                break
        ids_h, scores_h = ids.cpu(), beam_scores.cpu()
        out = []
        for b in range(B):
            if not done[b]:
                for j in range(nb):
                    hyp = ids_h[b * nb + j]
                    hyps[b].append((float(scores_h[b * nb + j]) / ((hyp.shape[-1] - len_offset) ** length_penalty), hyp))
                hyps[b] = sorted(hyps[b], key=lambda t: -t[0])[:nb]
            out.append(hyps[b][0][1])
        L = min(max(len(h) for h in out) + 1, max_length)
        seq = torch.full((B, L), pad_token_id, dtype=torch.long)
        for b, h in enumerate(out):
            seq[b, : len(h)] = h
            if len(h) < max_length:
                seq[b, len(h)] = eos_token_id
        return seq.to(dev)

    def forward(self, inputs_embeds, labels):
        enc = self.encoder(inputs_embeds)
        dec_in = self.embed(self.shift_right(labels), inputs_embeds.dtype)
        dec = self.decoder(dec_in, enc)
        logits = ops.linear(dec, self.shared.weight, None, alpha=self.config.d_model ** -0.5)
        loss = ops.cross_entropy(logits, labels)
        return NS(loss=loss, logits=logits, encoder_last_hidden_state=enc)
