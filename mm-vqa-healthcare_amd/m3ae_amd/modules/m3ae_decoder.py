"""DecoderModel on MI355X: frozen M3AE -> 6-layer Transformer decoder head -> vocabulary logits -> cross-entropy, and
the greedy `search_path` (reference m3ae/modules/m3ae_decoder.py:14-443; the model
`main_decoder_m3ae.py` trains, run_scripts/finetune_m3ae_decoder.sh).

Same class names / constructor (`DecoderModel(m3ae_config)`), same state_dict names (`m3ae.*`,
`decoder.dec_layers.{i}.{mha1,mha2}.{in_proj_weight,in_proj_bias,out_proj.*}`, `.ffn.{0,2}.*`,
`.{pre_norm,layernorm1,layernorm2,layernorm3}.*`, `decoder.target_embedding.weight`,
`decoder.positional_encoding.pe`, `decoder.final_linear.*`), same `training_step -> {'loss': ...}` contract.

The reference's arithmetic is kept as it is, quirks included (each is pinned by tests/golden/tiny_decoder.npz):
  * `target_embed += positional_encoding(target_embed)` (:127): the decoder input is 2 * embedding + pe;
  * every layer is fed `target_embed` (:131-134), so only the LAST layer's output reaches `final_linear`.  Layers
    0 .. n-2 are dead compute whose parameters never get a gradient; they are kept (state_dict, checkpoints) but NOT
    executed here -- the results are identical and 5/6 of the head's FLOPs disappear;
  * loss = CrossEntropy(mean over non-[PAD] golden tokens) (:226,:366-368 reduce to exactly that);
  * decoder input = tokens[:, :-1] with [SEP] -> [PAD], golden = tokens[:, 1:] (:344-351,:362).

Arithmetic: the head follows the configured mode like the M3AE below it.  Parity mode: fp32 on the library's generic
kernels (fp32 master weights, materialised-softmax attention).  Perf mode: bf16 activations, every Linear (in / out
projections, FFN, the vocabulary projection through `ops.vocab_linear`) on the bf16 MFMA GEMMs against the weight shadows;
the head has 8 heads of 96 channels (768 / 8) and the MFMA attention kernels are specialised for 64-wide heads, so its
two attentions run the fp32 materialised-softmax kernels on fp32 copies of the bf16 projections (`ops.attn_forward`:
T <= 12 tokens against 2 (CLS) or 2 + 577 + 32 encoder tokens -- a few % of the head's work).  train() mode applies the
reference's dropout sites with the library's counter-hash masks.
Tokenisation happens outside (`batch["decoder_tokens"]`: int64 [B, T] = [CLS] answer [SEP] [PAD]...) or through a
`tokenizer` callable; string metrics (ROUGE / BLEU / exact match) stay out of scope.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..param_store import ParamStore, group_hparams_decoder, param_group_of_decoder
from .m3ae_module import M3AETransformerSS, _Base, _HParams, pl

NEG = -30000.0  # additive stand-in for the reference's -inf key padding (exp underflows to exactly 0 in fp32)


class _InProjAttention(nn.Module):
    """Parameter container with nn.MultiheadAttention's names (in_proj_weight [3D, D], in_proj_bias, out_proj)."""

    def __init__(self, d_model, num_heads):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)


class DecoderLayer(nn.Module):
    """m3ae_decoder.py:38-90: t + MHA1(pre_norm(t)) -> + MHA2(LN1(.), enc) -> + FFN(LN2(.)) -> LN3."""

    def __init__(self, d_model, num_heads, d_ff, dropout=0.1):
        super().__init__()
        self.mha1 = _InProjAttention(d_model, num_heads)
        self.mha2 = _InProjAttention(d_model, num_heads)
        self.ffn = nn.Sequential(nn.Linear(d_model, d_ff), nn.ReLU(), nn.Linear(d_ff, d_model))
        self.pre_norm = nn.LayerNorm(d_model)
        self.layernorm1 = nn.LayerNorm(d_model)
        self.layernorm2 = nn.LayerNorm(d_model)
        self.layernorm3 = nn.LayerNorm(d_model)
        self.p_drop = dropout

    @staticmethod
    def _ln(ln, x):
        return ops.layer_norm(x, ln.weight, ln.bias, ln.eps)

    def forward(self, t, enc, key_mask):
        D, H = self.mha1.embed_dim, self.mha1.num_heads
        p = self.p_drop if self.training else 0.0   # MultiheadAttention(dropout=p) on the weights + dropout1..3 (:43-53)
        att = (lambda: (p, ops.next_dropout_seed())) if p > 0 else (lambda: None)
        # masked (causal + key padding) self-attention on the pre-normed input; residual = the un-normed input
        xn = self._ln(self.pre_norm, t)
        qkv = ops.linear(xn, self.mha1.in_proj_weight, self.mha1.in_proj_bias)
        ctx = ops.self_attention(qkv, key_mask, H, dropout=att(), causal=True)
        if p > 0:
            x = ops.dropout(ops.linear(ctx, self.mha1.out_proj.weight, self.mha1.out_proj.bias), p) + t
        else:
            x = ops.linear(ctx, self.mha1.out_proj.weight, self.mha1.out_proj.bias, residual=t)
        # cross-attention to the multimodal features (no mask: the reference's enc_pad_mask is all False, :71-76)
        xn = self._ln(self.layernorm1, x)
        q = ops.linear(xn, self.mha2.in_proj_weight, self.mha2.in_proj_bias)[..., :D]
        kv = ops.linear(enc, self.mha2.in_proj_weight, self.mha2.in_proj_bias)[..., D:]
        ctx = ops.cross_attention(q, kv, None, H, att())
        if p > 0:
            x = ops.dropout(ops.linear(ctx, self.mha2.out_proj.weight, self.mha2.out_proj.bias), p) + x
        else:
            x = ops.linear(ctx, self.mha2.out_proj.weight, self.mha2.out_proj.bias, residual=x)
        xn = self._ln(self.layernorm2, x)
        if p > 0:
            x = ops.dropout(ops.mlp(xn, self.ffn[0].weight, self.ffn[0].bias, self.ffn[2].weight, self.ffn[2].bias,
                                    ops.ACT_RELU), p) + x
        else:
            x = ops.mlp(xn, self.ffn[0].weight, self.ffn[0].bias, self.ffn[2].weight, self.ffn[2].bias, ops.ACT_RELU,
                        residual=x)
        return self._ln(self.layernorm3, x)

    @torch.no_grad()
    def cross_kv(self, enc):
        """Keys | values of the multimodal features for mha2 ([B, Le, 2 D]): computed once per generation."""
        D = self.mha1.embed_dim
        return ops.linear(enc, self.mha2.in_proj_weight, self.mha2.in_proj_bias)[..., D:]

    @torch.no_grad()
    def step(self, t_new, enc_kv, cache, pos):
        """Inference: ONE new position `pos` (t_new [B, 1, D]).  Its self-attention key | value row is appended to `cache`
        [B, Tmax, 2 D]; the new query attends to rows 0..pos -- the keys the causal mask leaves to the last row of a
        full-prefix pass -- and to the cached encoder-side keys / values.  Same kernels as forward()."""
        D, H = self.mha1.embed_dim, self.mha1.num_heads
        B = t_new.shape[0]
        xn = self._ln(self.pre_norm, t_new)
        qkv = ops.linear(xn, self.mha1.in_proj_weight, self.mha1.in_proj_bias)          # [B, 1, 3 D]
        cache[:, pos].copy_(qkv[:, 0, D:])
        ctx, _ = ops.attn_forward(qkv[..., :D], cache[:, :pos + 1, :D], cache[:, :pos + 1, D:], H)
        x = ops.linear(ctx, self.mha1.out_proj.weight, self.mha1.out_proj.bias, residual=t_new)
        xn = self._ln(self.layernorm1, x)
        q = ops.linear(xn, self.mha2.in_proj_weight, self.mha2.in_proj_bias)[..., :D]
        ctx, _ = ops.attn_forward(q, enc_kv[..., :D], enc_kv[..., D:], H)
        x = ops.linear(ctx, self.mha2.out_proj.weight, self.mha2.out_proj.bias, residual=x)
        xn = self._ln(self.layernorm2, x)
        x = ops.mlp(xn, self.ffn[0].weight, self.ffn[0].bias, self.ffn[2].weight, self.ffn[2].bias, ops.ACT_RELU, residual=x)
        return self._ln(self.layernorm3, x)


class PositionalEncoding(nn.Module):
    """m3ae_decoder.py:22-36 (fixed sinusoid table, a persistent buffer)."""

    def __init__(self, d_model, max_len=1024):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))


class _DecoderEmbedFn(torch.autograd.Function):
    """2 * embedding[ids] + pe (m3ae_decoder.py:125-127), gradient 2 * d_out scatter-added into the table."""

    @staticmethod
    def forward(ctx, ids, weight, pe_rows, out_dtype=torch.float32):
        n, D = ids.numel(), weight.shape[1]
        L = ops._lib.lib()
        e = torch.empty((n, D), dtype=weight.dtype, device=weight.device)
        ops.check(L.m3ae_gather_rows(ops._p(weight.data), ops._p(ids), ops._p(e), n, D, ops._dt(e), ops._stream()),
                  "m3ae_gather_rows")
        out = torch.empty_like(e)
        ops.check(L.m3ae_add(ops._p(e), ops._p(e), ops._p(out), e.numel(), ops._dt(e), ops._stream()), "m3ae_add")
        ops.check(L.m3ae_add(ops._p(out), ops._p(pe_rows), ops._p(out), e.numel(), ops._dt(e), ops._stream()), "m3ae_add")
        ctx.save_for_backward(ids)
        ctx.weight = weight
        return out.to(out_dtype)   # gathered and doubled from the fp32 master rows, rounded once

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        w = ctx.weight
        if w.requires_grad:
            d = dout.float().contiguous()
            d2 = torch.empty_like(d)
            ops.check(ops._lib.lib().m3ae_add(ops._p(d), ops._p(d), ops._p(d2), d.numel(), ops._dt(d), ops._stream()),
                      "m3ae_add")
            ops._grad_buf(w).index_add_(0, ids, d2)
            ops._done(w)
        return None, None, None, None


class Decoder(nn.Module):
    """m3ae_decoder.py:93-182."""

    def __init__(self, num_layers, d_model, num_heads, d_ff, dropout, max_len, target_vocab_size):
        super().__init__()
        self.max_len, self.num_layers, self.num_heads = max_len, num_layers, num_heads
        self.dec_layers = nn.ModuleList([DecoderLayer(d_model, num_heads, d_ff, dropout) for _ in range(num_layers)])
        self.target_embedding = nn.Embedding(target_vocab_size, d_model)
        self.positional_encoding = PositionalEncoding(d_model)
        self.final_linear = nn.Linear(d_model, target_vocab_size)
        self.p_drop = dropout   # nn.Dropout on the (doubled) embedding (:128)
        self.head_dtype = torch.float32   # DecoderModel.finalize: the configured compute dtype

    def weight_units(self):
        """GEMM weights that need a transposed bf16 copy for dgrad (perf mode): the live layer's and the vocabulary
        projection; none in parity mode (the fp32 kernels read the master weights through strides)."""
        if self.head_dtype != torch.bfloat16:
            return []
        l = self.dec_layers[self.num_layers - 1]
        return [l.mha1.in_proj_weight, l.mha1.out_proj.weight, l.mha2.in_proj_weight, l.mha2.out_proj.weight,
                l.ffn[0].weight, l.ffn[2].weight, self.final_linear.weight]

    def forward(self, padded_targets, padding_mask, cross_attn_feats):
        """padded_targets int64 [B, T]; padding_mask bool [B, T] (True = token) or None; features [B, Le, D]."""
        B, T = padded_targets.shape
        D = self.target_embedding.weight.shape[1]
        pe_rows = self.positional_encoding.pe[0, :T].to(torch.float32).repeat(B, 1).contiguous()
        t = _DecoderEmbedFn.apply(padded_targets.reshape(-1).contiguous(), self.target_embedding.weight, pe_rows,
                                  self.head_dtype)
        t = ops.dropout(t.view(B, T, D), self.p_drop, self.training)
        key_mask = None
        if padding_mask is not None:
            key_mask = torch.where(padding_mask, 0.0, NEG).to(torch.float32).contiguous()
        # the reference runs all layers on `t` and keeps the last result (:131-134): run the last one only
        x = self.dec_layers[self.num_layers - 1](t, cross_attn_feats.to(self.head_dtype), key_mask)
        return ops.vocab_linear(x, self.final_linear.weight, self.final_linear.bias)

    @torch.no_grad()
    def step_logits(self, last_ids, pos, enc_kv, cache):
        """Logits of the next token from the NEW token alone (last_ids [B, 1] at position `pos`): the prefix's keys / values
        come from `cache`, the encoder side from `enc_kv`; equal to forward(prefix)[:, -1] (eval mode)."""
        B = last_ids.shape[0]
        pe_rows = self.positional_encoding.pe[0, pos:pos + 1].to(torch.float32).repeat(B, 1).contiguous()
        t = _DecoderEmbedFn.apply(last_ids.reshape(-1).contiguous(), self.target_embedding.weight, pe_rows,
                                  self.head_dtype)
        x = self.dec_layers[self.num_layers - 1].step(t.view(B, 1, -1), enc_kv, cache, pos)
        return ops.vocab_linear(x, self.final_linear.weight, self.final_linear.bias)[:, -1]

    @torch.no_grad()
    def search_path(self, cross_attn_feats, cls_id=101, sep_id=102, eos_id=None, pad_id=0, use_cache=True):
        """m3ae_decoder.py:141-182: greedy decoding.  The reference re-runs the whole prefix every step; here a step is one
        decoder row: the prefix's self-attention keys / values are cached, the encoder-side keys / values are projected
        once (`use_cache=False` keeps the reference's loop; both give the same tokens)."""
        B, dev = cross_attn_feats.shape[0], cross_attn_feats.device
        seq = torch.full((B, 1), cls_id, dtype=torch.long, device=dev)
        finished = torch.zeros(B, dtype=torch.bool, device=dev)
        layer = self.dec_layers[self.num_layers - 1]
        if use_cache:
            D = self.target_embedding.weight.shape[1]
            enc_kv = layer.cross_kv(cross_attn_feats.to(self.head_dtype))
            cache = torch.empty((B, self.max_len, 2 * D), dtype=self.head_dtype, device=dev)
        for step in range(self.max_len):
            if use_cache:
                nxt = self.step_logits(seq[:, -1:], step, enc_kv, cache).argmax(dim=-1)
            else:
                nxt = self.forward(seq, None, cross_attn_feats)[:, -1].argmax(dim=-1)
            hit = nxt == sep_id
            if eos_id is not None:
                hit = hit | (nxt == eos_id)
            finished |= hit
            seq = torch.cat([seq, nxt[:, None]], dim=1)
            if bool(finished.all()):
                break
        seq = seq[:, 1:]
        special = seq == sep_id
        if eos_id is not None:
            special = special | (seq == eos_id)
        after = (special.long().cumsum(1) - special.long()) > 0  # strictly after the first special token
        seq = torch.where(after, torch.full_like(seq, pad_id), seq)
        return torch.nn.functional.pad(seq, (0, self.max_len - seq.shape[1]), value=pad_id)


class DecoderModel(_Base):
    """m3ae_decoder.py:185-443."""

    def __init__(self, m3ae_config, max_answer_length=12, min_answer_length=1, freeze_m3ae=True, tokenizer=None,
                 vocab_size=30522, special_ids=(101, 102, 0)):
        super().__init__()
        if pl is None:
            self.hparams = _HParams(m3ae_config=m3ae_config)
        else:
            self.save_hyperparameters(ignore=["tokenizer"])
        self.tokenizer = tokenizer
        if tokenizer is not None:
            vocab_size = tokenizer.vocab_size
            special_ids = (tokenizer.cls_token_id, tokenizer.sep_token_id, tokenizer.pad_token_id)
        self.cls_id, self.sep_id, self.pad_id = special_ids
        self.m3ae = M3AETransformerSS(m3ae_config)
        self.decoder = Decoder(num_layers=6, d_model=768, num_heads=8, d_ff=768 * 4, dropout=0.1, max_len=128,
                               target_vocab_size=vocab_size)
        if freeze_m3ae:
            for p in self.m3ae.parameters():
                p.requires_grad = False
        self.max_answer_length, self.min_answer_length = max_answer_length, min_answer_length
        self.current_tasks = list()
        self.store = None

    def weight_units(self):
        return list(self.m3ae.weight_units()) + self.decoder.weight_units()

    def finalize(self, device="cuda", compute_dtype=None):
        cfg = self.m3ae.hparams.config
        if compute_dtype is not None:
            self.m3ae._dtype = compute_dtype
        for _, b in list(self.named_buffers()):
            b.data = b.data.to(device)
        # layers 0 .. n-2 never receive a gradient in the reference (torch's AdamW skips them: no update, no weight
        # decay): keep them out of the optimizer / all-reduce buffers
        dead = tuple(f"decoder.dec_layers.{i}." for i in range(self.decoder.num_layers - 1))
        self.decoder.head_dtype = self.m3ae._dtype   # before the store asks for the weight units
        self.store = ParamStore(self, cfg, device, self.m3ae._dtype, self.weight_units, frozen=dead,
                                group_fn=param_group_of_decoder, hparams_fn=group_hparams_decoder)
        self.m3ae.store = self.store
        return self

    def features(self, batch):
        """m3ae_decoder.py:296-314: the encoder-side sequence the decoder cross-attends to."""
        cfg = self.m3ae.hparams.config
        with torch.no_grad():
            out = self.m3ae.infer(batch, mask_text=False, mask_image=False)
        parts = []
        if cfg["mm_encoder_inputs_include_imagetext_feats"]:
            parts += [out["multi_modal_image_feats"], out["multi_modal_text_feats"]]
        if cfg["mm_encoder_inputs_include_cls_feats"]:
            parts.append(out["multi_modal_cls_feats"].view(-1, 2, 768))
        return torch.cat(parts, dim=1).to(self.decoder.head_dtype).contiguous()

    def tokens_of(self, batch):
        if "decoder_tokens" in batch:
            return batch["decoder_tokens"]
        if self.tokenizer is None:
            raise ValueError("pass batch['decoder_tokens'] or construct with a tokenizer")
        flat = [a[0] for a in batch["vqa_answer"]]
        return self.tokenizer(flat, padding=True, truncation=True, return_tensors="pt",
                              max_length=self.max_answer_length).input_ids.to(batch["text_ids"].device)

    def forward(self, batch, test=False):
        if self.store is None:
            raise RuntimeError("call finalize(device) before the first forward")
        enc = self.features(batch)
        if len(self.current_tasks) == 0 or test:
            return {"vqa_loss": 0, "generated_ids": self.decoder.search_path(enc, self.cls_id, self.sep_id, None,
                                                                             self.pad_id)}
        tokens = self.tokens_of(batch)
        tin = tokens[:, :-1].clone()
        tin[tin == self.sep_id] = self.pad_id                   # :344-348
        mask = tin != self.pad_id                               # :351
        logits = self.decoder(tin, mask, enc)
        gold = tokens[:, 1:]
        labels = torch.where(gold == self.pad_id, torch.full_like(gold, -100), gold)  # ignore_index = [PAD] (:226)
        loss = ops.cross_entropy(logits, labels.contiguous())
        return {"vqa_loss": loss, "vqa_logits": logits}

    def training_step(self, batch, batch_idx=0):
        """m3ae_decoder.py:390-396."""
        names = self.m3ae.hparams.config["loss_names"]
        self.current_tasks = [k for k, v in names.items() if v > 0]
        output = self(batch)
        total = sum(v * names[k.replace("_loss", "")] for k, v in output.items() if k.endswith("_loss"))
        return {"loss": total}

    def configure_optimizers(self):
        """m3ae_t5_utils.set_schedule_decoder (:290-375): two groups by substring, one lr -- ParamStore.adamw_step."""
        tr = getattr(self, "trainer_ref", None)
        return self.store.make_optimizer(getattr(tr, "max_steps", None) if tr is not None else None)
