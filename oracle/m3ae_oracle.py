"""ORACLE -- CPU restatement of the reference's M3AE Med-VQA hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; it is the
checker, never the product.  Plain PyTorch-CPU fp32 tensor algebra (matmul / softmax / mean / erf),
written as pure functions over a state_dict whose key names and shapes are the reference's
(SURVEY.md 8b "Weights").  Gradients come from torch autograd over these functions.

Parity pin: this restatement is checked against fixtures produced by importing and running the
reference's own modules in the build container (oracle/make_golden.py -> tests/golden/*.npz;
tests/test_oracle_golden.py).  Third-party arithmetic the reference calls into (transformers==4.6.0
RobertaModel, torch 1.9 nn.MultiheadAttention) is pinned by the container's transformers 5.15 /
torch 2.10 implementations of the same math.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------------
def linear(sd, prefix, x):
    return x @ sd[prefix + ".weight"].t() + sd[prefix + ".bias"]


def layer_norm(sd, prefix, x, eps):
    """nn.LayerNorm (biased variance, eps inside the sqrt).  clip_model.py:27-33 computes it in
    fp32 and casts back; in this all-fp32 oracle that is the identity."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * sd[prefix + ".weight"] + sd[prefix + ".bias"]


def gelu_erf(x):
    """ACT2FN["gelu"] (bert_model.py:421) and nn.GELU() (m3ae_module.py:123): exact erf form."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def quick_gelu(x):
    """clip_model.py:36-38."""
    return x * torch.sigmoid(1.702 * x)


def extended_mask(mask):
    """transformers 4.6.0 get_extended_attention_mask as called at m3ae_module.py:232,255:
    (1 - mask)[:, None, None, :] * -10000.0 (additive, NOT -inf)."""
    return (1.0 - mask[:, None, None, :].to(torch.float32)) * -10000.0


def split_heads(x, heads):
    B, L, D = x.shape
    return x.view(B, L, heads, D // heads).permute(0, 2, 1, 3)


def merge_heads(x):
    B, H, L, dh = x.shape
    return x.permute(0, 2, 1, 3).reshape(B, L, H * dh)


def sdpa(q, k, v, add_mask):
    """bert_model.py:301-340: scores = QK^T / sqrt(dh) + mask; softmax; PV."""
    dh = q.shape[-1]
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if add_mask is not None:
        s = s + add_mask
    p = torch.softmax(s, dim=-1)
    return p @ v


# ------------------------------------------------------------------------------------------------
# BERT / RoBERTa blocks (bert_model.py:211-503; identical math in HF RobertaLayer)
# ------------------------------------------------------------------------------------------------
def bert_attention(sd, prefix, h, kv, add_mask, heads, eps):
    """BertAttention (bert_model.py:367-413) = BertSelfAttention (:253-350) + BertSelfOutput (:360-364).
    `kv` is `h` for self-attention, the other stream for cross-attention (:275-278); the residual is `h`."""
    q = split_heads(linear(sd, prefix + ".self.query", h), heads)
    k = split_heads(linear(sd, prefix + ".self.key", kv), heads)
    v = split_heads(linear(sd, prefix + ".self.value", kv), heads)
    ctx = merge_heads(sdpa(q, k, v, add_mask))
    out = linear(sd, prefix + ".output.dense", ctx)
    return layer_norm(sd, prefix + ".output.LayerNorm", out + h, eps)


def bert_ffn(sd, prefix, h, eps):
    """BertIntermediate (:416-428) + BertOutput (:431-442), feed_forward_chunk (:500-503)."""
    u = gelu_erf(linear(sd, prefix + ".intermediate.dense", h))
    out = linear(sd, prefix + ".output.dense", u)
    return layer_norm(sd, prefix + ".output.LayerNorm", out + h, eps)


def bert_cross_layer(sd, prefix, h, other, mask_self, mask_other, heads, eps=1e-12):
    """BertCrossLayer.forward (bert_model.py:457-498): self-attn -> cross-attn (residual = self-attn
    output) -> FFN.  eps = RobertaConfig default 1e-12 (m3ae_module.py:24-33)."""
    a = bert_attention(sd, prefix + ".attention", h, h, mask_self, heads, eps)
    c = bert_attention(sd, prefix + ".crossattention", a, other, mask_other, heads, eps)
    return bert_ffn(sd, prefix, c, eps)


def roberta_layer(sd, prefix, h, add_mask, heads, eps=1e-5):
    """HF RobertaLayer (third party, called at m3ae_module.py:233-234): post-LN self-attn + FFN."""
    a = bert_attention(sd, prefix + ".attention", h, h, add_mask, heads, eps)
    return bert_ffn(sd, prefix, a, eps)


def roberta_embeddings(sd, prefix, ids, pad_id=1, eps=1e-5):
    """HF RobertaEmbeddings (third party; m3ae_module.py:230): position ids =
    cumsum(ids != pad) * (ids != pad) + pad; word + type[0] + pos; LayerNorm(1e-5)."""
    ne = (ids != pad_id).long()
    pos = torch.cumsum(ne, dim=1) * ne + pad_id
    e = (sd[prefix + ".word_embeddings.weight"][ids]
         + sd[prefix + ".token_type_embeddings.weight"][torch.zeros_like(ids)]
         + sd[prefix + ".position_embeddings.weight"][pos])
    return layer_norm(sd, prefix + ".LayerNorm", e, eps)


# ------------------------------------------------------------------------------------------------
# CLIP ViT (clip_model.py:41-128)
# ------------------------------------------------------------------------------------------------
def clip_block(sd, prefix, x, heads):
    """ResidualAttentionBlock.forward (clip_model.py:60-63), nn.MultiheadAttention with packed
    in_proj_weight [3d, d] (rows Q, K, V), q scaled by 1/sqrt(dh) before QK^T (torch 1.9), no mask."""
    d = x.shape[-1]
    h = layer_norm(sd, prefix + ".ln_1", x, 1e-5)
    qkv = h @ sd[prefix + ".attn.in_proj_weight"].t() + sd[prefix + ".attn.in_proj_bias"]
    q, k, v = qkv.split(d, dim=-1)
    ctx = merge_heads(sdpa(split_heads(q, heads), split_heads(k, heads), split_heads(v, heads), None))
    x = x + linear(sd, prefix + ".attn.out_proj", ctx)
    h = layer_norm(sd, prefix + ".ln_2", x, 1e-5)
    h = quick_gelu(linear(sd, prefix + ".mlp.c_fc", h))
    return x + linear(sd, prefix + ".mlp.c_proj", h)


def clip_patch_embed(sd, prefix, img):
    """VisualTransformer.forward_patch_embed (clip_model.py:110-116): conv k=s=patch, no bias ->
    [B, grid^2, width] row-major (h then w) -> prepend class_embedding."""
    w = sd[prefix + ".conv1.weight"]
    x = F.conv2d(img, w, stride=w.shape[-1])
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd[prefix + ".class_embedding"].view(1, 1, -1).expand(x.shape[0], 1, -1)
    return torch.cat([cls, x], dim=1)


def clip_trans(sd, prefix, x, heads):
    """forward_trans (clip_model.py:122-128): ln_pre -> (layers-1) blocks (:71) -> ln_post; no proj."""
    x = layer_norm(sd, prefix + ".ln_pre", x, 1e-5)
    i = 0
    while f"{prefix}.transformer.resblocks.{i}.ln_1.weight" in sd:
        x = clip_block(sd, f"{prefix}.transformer.resblocks.{i}", x, heads)
        i += 1
    return layer_norm(sd, prefix + ".ln_post", x, 1e-5)


def clip_visual(sd, prefix, img, heads):
    """VisualTransformer.forward (clip_model.py:93-108)."""
    x = clip_patch_embed(sd, prefix, img)
    x = x + sd[prefix + ".positional_embedding"]
    return clip_trans(sd, prefix, x, heads)


# ------------------------------------------------------------------------------------------------
# MIM helpers (m3ae_module.py:153-192)
# ------------------------------------------------------------------------------------------------
def random_masking(sd, x, mask_ratio, noise):
    """m3ae_module.py:153-183 with the U(0,1) `noise` [B, L] supplied by the caller."""
    x_, x = x[:, :1], x[:, 1:]
    pos = sd["vision_encoder.visual.positional_embedding"].unsqueeze(0)
    N, L, D = x.shape
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :len_keep]
    x = x + pos[:, 1:]
    x_masked = torch.gather(x, 1, ids_keep.unsqueeze(-1).repeat(1, 1, D))
    mask = torch.ones(N, L)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    x_masked = torch.cat((x_ + pos[:, :1], x_masked), dim=1)
    return x_masked, mask, ids_restore


def patchify(imgs, p):
    """m3ae_module.py:185-192: nchpwq -> nhwpqc."""
    h = w = imgs.shape[2] // p
    x = imgs.reshape(imgs.shape[0], 3, h, p, w, p)
    x = torch.einsum("nchpwq->nhwpqc", x)
    return x.reshape(imgs.shape[0], h * w, p * p * 3)


# ------------------------------------------------------------------------------------------------
# M3AETransformerSS.infer (m3ae_module.py:203-312) and heads
# ------------------------------------------------------------------------------------------------
def pooler(sd, prefix, h):
    """prediction_heads.py:9-19."""
    return torch.tanh(linear(sd, prefix + ".dense", h[:, 0]))


def infer(sd, cfg, img, text_ids, text_masks, mim_noise=None, trail=None):
    """M3AETransformerSS.infer (m3ae_module.py:203-312), eval mode (no dropout).
    cfg: dict(num_heads, vit_heads, text_heads, num_top_layer, patch_size, mim_prob, mim_layer).
    If `mim_noise` is given, runs the mask_image=True branch (:239-249)."""
    ret = {}
    H, Hv, Ht = cfg["num_heads"], cfg["vit_heads"], cfg["text_heads"]
    # text (m3ae_module.py:230-235)
    t = roberta_embeddings(sd, "language_encoder.embeddings", text_ids)
    mt = extended_mask(text_masks)
    i = 0
    while f"language_encoder.encoder.layer.{i}.attention.self.query.weight" in sd:
        t = roberta_layer(sd, f"language_encoder.encoder.layer.{i}", t, mt, Ht)
        i += 1
    if trail is not None:
        trail["text_enc"] = t
    t = linear(sd, "multi_modal_language_proj", t)
    # image (m3ae_module.py:239-256)
    if mim_noise is not None:
        v = clip_patch_embed(sd, "vision_encoder.visual", img)
        v, mim_masks, ids_restore = random_masking(sd, v, cfg["mim_prob"], mim_noise)
        v = clip_trans(sd, "vision_encoder.visual", v, Hv)
        ret["mim_masks"], ret["mim_ids_restore"] = mim_masks, ids_restore
    else:
        v = clip_visual(sd, "vision_encoder.visual", img, Hv)
    if trail is not None:
        trail["image_enc"] = v
    v = linear(sd, "multi_modal_vision_proj", v)
    mv = torch.zeros(v.shape[0], 1, 1, v.shape[1])  # all-ones mask -> additive zeros (:253-256)
    # type embeddings (m3ae_module.py:260-263)
    t = t + sd["modality_type_embeddings.weight"][0]
    v = v + sd["modality_type_embeddings.weight"][1]
    # fusion (m3ae_module.py:269-278): both streams read the PRE-update x, y
    x, y = t, v
    for l in range(cfg["num_top_layer"]):
        if mim_noise is not None and cfg.get("mim_layer", -1) == l:
            ret[f"multi_modal_text_feats_{l}"], ret[f"multi_modal_image_feats_{l}"] = x, y
        x1 = bert_cross_layer(sd, f"multi_modal_language_layers.{l}", x, y, mt, mv, H)
        y1 = bert_cross_layer(sd, f"multi_modal_vision_layers.{l}", y, x, mv, mt, H)
        x, y = x1, y1
        if trail is not None:
            trail[f"fusion_text_{l}"], trail[f"fusion_image_{l}"] = x, y
    # pool (m3ae_module.py:288-296)
    cls = torch.cat([pooler(sd, "multi_modal_language_pooler", x), pooler(sd, "multi_modal_vision_pooler", y)], -1)
    ret.update(multi_modal_text_feats=x, multi_modal_image_feats=y, multi_modal_cls_feats=cls,
               extended_text_masks=mt, extended_image_masks=mv)
    return ret


def vqa_head(sd, cls):
    """m3ae_module.py:118-126: Linear -> LayerNorm(1e-5) -> GELU(erf) -> Linear."""
    h = linear(sd, "vqa_head.0", cls)
    h = gelu_erf(layer_norm(sd, "vqa_head.1", h, 1e-5))
    return linear(sd, "vqa_head.3", h)


def vqa_targets(vqa_labels, vqa_scores, label_size):
    """objectives.py:188-197."""
    t = torch.zeros(len(vqa_labels), label_size)
    for i, (ls, ss) in enumerate(zip(vqa_labels, vqa_scores)):
        for l, s in zip(ls, ss):
            t[i, l] = s
    return t


def vqa_loss(logits, targets):
    """objectives.py:201: BCEWithLogits(mean over all elements) * num_labels."""
    x, z = logits, targets
    per = torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs()))
    return per.mean() * targets.shape[1]


def training_loss(sd, cfg, batch):
    """training_step for loss_names = {vqa: 1} (m3ae_module.py:347-353 -> objectives.py:176-201)."""
    out = infer(sd, cfg, batch["image"][0], batch["text_ids"], batch["text_masks"])
    logits = vqa_head(sd, out["multi_modal_cls_feats"])
    tgt = vqa_targets(batch["vqa_labels"], batch["vqa_scores"], logits.shape[1])
    return vqa_loss(logits, tgt), logits, out


# ------------------------------------------------------------------------------------------------
# pretraining heads / losses (prediction_heads.py:22-96; objectives.py:14-119)
# ------------------------------------------------------------------------------------------------
def mlm_head(sd, x, eps=1e-12):
    """MLMHead (prediction_heads.py:22-34): BertPredictionHeadTransform (dense + gelu + LN) ->
    decoder (no bias) + bias."""
    h = gelu_erf(linear(sd, "mlm_head.transform.dense", x))
    h = layer_norm(sd, "mlm_head.transform.LayerNorm", h, eps)
    return h @ sd["mlm_head.decoder.weight"].t() + sd["mlm_head.bias"]


def mlm_loss(logits, labels):
    """objectives.py:19-23: CE with ignore_index=-100, mean over non-ignored."""
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=-100)


def mim_head(sd, x, ids_restore, heads):
    """MIMHead.forward (prediction_heads.py:60-86); decoder = Transformer(layers+1) -> layers blocks."""
    x = linear(sd, "mim_head.decoder_embed", x)
    n_mask = ids_restore.shape[1] + 1 - x.shape[1]
    mt = sd["mim_head.mask_token"].repeat(x.shape[0], n_mask, 1)
    x_ = torch.cat([x[:, 1:], mt], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))
    x = torch.cat([x[:, :1], x_], dim=1)
    x = x + sd["mim_head.decoder_pos_embed"]
    i = 0
    while f"mim_head.decoder.resblocks.{i}.ln_1.weight" in sd:
        x = clip_block(sd, f"mim_head.decoder.resblocks.{i}", x, heads)
        i += 1
    x = layer_norm(sd, "mim_head.decoder_norm", x, 1e-5)
    x = linear(sd, "mim_head.decoder_pred", x)
    return x[:, 1:]


def mim_loss(pred, img, mask, patch, norm_pix=True):
    """objectives.py:52-62: per-patch normalised target (UNBIASED var + 1e-6), MSE on masked patches."""
    target = patchify(img, patch)
    if norm_pix:
        mean = target.mean(-1, keepdim=True)
        var = target.var(-1, keepdim=True)
        target = (target - mean) / (var + 1e-6) ** 0.5
    l = ((pred - target) ** 2).mean(-1)
    return (l * mask).sum() / mask.sum()


def itm_head(sd, cls):
    """ITMHead (prediction_heads.py:89-96)."""
    return linear(sd, "itm_head.fc", cls)


# ------------------------------------------------------------------------------------------------
# optimizer / schedule (m3ae_utils.py:112-242)
# ------------------------------------------------------------------------------------------------
NO_DECAY = ["bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight",
            "norm1.bias", "norm1.weight", "norm2.bias", "norm2.weight"]
HEAD_NAMES = ["mlm_head", "mim_head", "itm_head", "vqa_head", "cls_head", "irtr_head"]


def param_group_of(name):
    """Index 0..5 of the six groups of m3ae_utils.py:135-204 (substring matches on the name)."""
    nd = any(k in name for k in NO_DECAY)
    hd = any(k in name for k in HEAD_NAMES)
    mm = "multi_modal" in name
    if not hd and not mm:
        return 1 if nd else 0
    if hd and not mm:
        return 3 if nd else 2
    if mm and not hd:
        return 5 if nd else 4
    return -1  # head AND multi_modal: in no group (the reference silently drops such params)


def poly_lr_factor(step, warmup_steps, max_steps, lr_init, lr_end=0.0, power=1.0):
    """transformers get_polynomial_decay_schedule_with_warmup (called at m3ae_utils.py:232-238)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    if step > max_steps:
        return lr_end / lr_init
    lr_range = lr_init - lr_end
    decay_steps = max_steps - warmup_steps
    pct_remaining = 1 - (step - warmup_steps) / decay_steps
    return (lr_range * pct_remaining ** power + lr_end) / lr_init


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.98, eps=1e-8):
    """transformers 4.6.0 AdamW.step (m3ae_utils.py:206): bias-corrected Adam update, then decoupled
    weight decay p -= lr * wd * p applied AFTER the update.  `step` counts from 1.  In place."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if wd > 0:
        p.add_(p, alpha=-lr * wd)
    return p


# ------------------------------------------------------------------------------------------------
# T5 generative head (m3ae_t5_mm_encoder_input.py:100-295; T5 arithmetic = third-party HF
# T5ForConditionalGeneration, transformers==4.6.0, restated from the container's implementation)
# ------------------------------------------------------------------------------------------------
def t5_rmsnorm(sd, key, x, eps=1e-6):
    """T5LayerNorm: x * rsqrt(mean(x^2) + eps) * weight (no mean subtraction, no bias)."""
    var = x.pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * sd[key]


def t5_rel_bucket(rel, bidirectional, num_buckets=32, max_distance=128):
    """T5Attention._relative_position_bucket."""
    ret = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        ret = ret + (rel > 0).long() * num_buckets
        rel = rel.abs()
    else:
        rel = -torch.min(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).long()
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return ret + torch.where(is_small, rel, large)


def t5_position_bias(table, Lq, Lk, bidirectional):
    """T5Attention.compute_bias -> [H, Lq, Lk]."""
    ctx = torch.arange(Lq)[:, None]
    mem = torch.arange(Lk)[None, :]
    bucket = t5_rel_bucket(mem - ctx, bidirectional)
    return table[bucket].permute(2, 0, 1)


def t5_attention(sd, prefix, x, kv, bias, heads):
    """T5Attention.forward: no 1/sqrt(d) scaling, additive position bias (+ mask), no linear biases."""
    q = split_heads(x @ sd[prefix + ".q.weight"].t(), heads)
    k = split_heads(kv @ sd[prefix + ".k.weight"].t(), heads)
    v = split_heads(kv @ sd[prefix + ".v.weight"].t(), heads)
    s = q @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias
    p = torch.softmax(s.float(), dim=-1)
    return merge_heads(p @ v) @ sd[prefix + ".o.weight"].t()


def t5_ff(sd, prefix, x):
    """T5DenseActDense (feed_forward_proj = relu): wo(relu(wi(x)))."""
    return torch.relu(x @ sd[prefix + ".wi.weight"].t()) @ sd[prefix + ".wo.weight"].t()


def t5_encoder(sd, embeds, heads, prefix="t5.encoder"):
    """T5Stack (encoder), all-ones attention mask (m3ae_t5_mm_encoder_input.py:174-178,202)."""
    L = embeds.shape[1]
    bias = t5_position_bias(sd[f"{prefix}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], L, L, True)
    h = embeds
    i = 0
    while f"{prefix}.block.{i}.layer.0.SelfAttention.q.weight" in sd:
        b = f"{prefix}.block.{i}"
        h = h + t5_attention(sd, b + ".layer.0.SelfAttention", t5_rmsnorm(sd, b + ".layer.0.layer_norm.weight", h),
                             t5_rmsnorm(sd, b + ".layer.0.layer_norm.weight", h), bias[None], heads)
        h = h + t5_ff(sd, b + ".layer.1.DenseReluDense", t5_rmsnorm(sd, b + ".layer.1.layer_norm.weight", h))
        i += 1
    return t5_rmsnorm(sd, prefix + ".final_layer_norm.weight", h)


def t5_decoder(sd, dec_ids, enc_out, heads, prefix="t5.decoder"):
    """T5Stack (decoder), teacher forced: causal self-attention with unidirectional relative bias, cross-attention
    over the encoder output (zero position bias, all-ones encoder mask)."""
    T = dec_ids.shape[1]
    h = sd["t5.shared.weight"][dec_ids]
    bias = t5_position_bias(sd[f"{prefix}.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], T, T, False)
    causal = torch.full((T, T), float("-inf")).triu(1)
    sbias = (bias + causal)[None]
    i = 0
    while f"{prefix}.block.{i}.layer.0.SelfAttention.q.weight" in sd:
        b = f"{prefix}.block.{i}"
        n = t5_rmsnorm(sd, b + ".layer.0.layer_norm.weight", h)
        h = h + t5_attention(sd, b + ".layer.0.SelfAttention", n, n, sbias, heads)
        n = t5_rmsnorm(sd, b + ".layer.1.layer_norm.weight", h)
        h = h + t5_attention(sd, b + ".layer.1.EncDecAttention", n, enc_out, None, heads)
        h = h + t5_ff(sd, b + ".layer.2.DenseReluDense", t5_rmsnorm(sd, b + ".layer.2.layer_norm.weight", h))
        i += 1
    return t5_rmsnorm(sd, prefix + ".final_layer_norm.weight", h)


def t5_shift_right(labels, start_id=0, pad_id=0):
    """T5ForConditionalGeneration._shift_right."""
    out = torch.zeros_like(labels)
    out[:, 1:] = labels[:, :-1]
    out[:, 0] = start_id
    return out.masked_fill(out == -100, pad_id)


def t5_head_inputs(sd, cls_feats, prefix_ids, proj_w, proj_b, max_len=512):
    """prepare_inputs (m3ae_t5_mm_encoder_input.py:100-190) with include_cls_feats=True, include_imagetext_feats=False:
    per sample [shared[prefix_ids] ; Linear(1536->512)(cls)] zero-padded to 512 rows.  The reference draws a FRESH
    random nn.Linear per sample per call (:75-77,128-129); here it is an explicit (proj_w, proj_b) -- documented
    deviation, see DESIGN.md."""
    B = cls_feats.shape[0]
    d = sd["t5.shared.weight"].shape[1]
    pre = sd["t5.shared.weight"][prefix_ids]                    # [P, d]
    proj = cls_feats @ proj_w.t() + proj_b                      # [B, d]
    x = torch.zeros(B, max_len, d)
    P = pre.shape[0]
    x[:, :P] = pre
    x[:, P] = proj
    return x


def t5_loss(sd, enc_in, labels, heads):
    """forward (m3ae_t5_mm_encoder_input.py:202,244): encoder -> teacher-forced decoder -> tied LM head (decoder
    output scaled by d_model^-0.5) -> CrossEntropy(ignore_index=-100) over ALL label positions (pad id 0 included,
    as the reference passes tokenizer output unmasked)."""
    enc = t5_encoder(sd, enc_in, heads)
    dec = t5_decoder(sd, t5_shift_right(labels), enc, heads)
    d = dec.shape[-1]
    logits = (dec * d ** -0.5) @ sd["t5.shared.weight"].t()
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), ignore_index=-100)
    return loss, logits


# ------------------------------------------------------------------------------------------------
# Decoder-only generative head (m3ae/modules/m3ae_decoder.py) -- SURVEY 8f-3.
# The reference's quirks are part of the contract and are restated as they are:
#   * `target_embed += positional_encoding(target_embed)` (:127) makes the input 2 * emb + pe;
#   * every layer is fed `target_embed`, not the previous layer's output (:131-134): only the LAST layer's
#     output reaches `final_linear`, layers 0..4 are dead compute and their parameters never get a gradient;
#   * criterion(mean over non-pad golden tokens) * padding_mask, summed / padding_mask.sum() (:366-368) is that
#     same mean again;
#   * the [SEP] at the end of every target is replaced by [PAD] in the decoder INPUT (:345-348), and the padding mask
#     is taken from that input.
# ------------------------------------------------------------------------------------------------
def decoder_pe(T, d):
    """PositionalEncoding (m3ae_decoder.py:24-33)."""
    pe = torch.zeros(T, d)
    pos = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def torch_mha(sd, prefix, xq, xkv, heads, attn_mask=None, key_padding_mask=None):
    """nn.MultiheadAttention(batch_first=True) forward in eval mode (third party: torch; called at
    m3ae_decoder.py:61-62,79-80): packed in_proj (q | k | v rows), scores / sqrt(dh), boolean masks -> -inf."""
    D = xq.shape[-1]
    W, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = xq @ W[:D].t() + b[:D]
    k = xkv @ W[D:2 * D].t() + b[D:2 * D]
    v = xkv @ W[2 * D:].t() + b[2 * D:]
    qh, kh, vh = split_heads(q, heads), split_heads(k, heads), split_heads(v, heads)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(D // heads)
    if attn_mask is not None:
        s = s.masked_fill(attn_mask[None, None], float("-inf"))
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    o = merge_heads(torch.softmax(s, dim=-1) @ vh)
    return o @ sd[prefix + ".out_proj.weight"].t() + sd[prefix + ".out_proj.bias"]


def decoder_layer(sd, p, t, enc, pad_mask_dec, causal, heads=8):
    """DecoderLayer.forward (m3ae_decoder.py:55-90), eval mode."""
    x = t + torch_mha(sd, p + ".mha1", layer_norm(sd, p + ".pre_norm", t, 1e-5), layer_norm(sd, p + ".pre_norm", t, 1e-5),
                      heads, causal, pad_mask_dec)
    xn = layer_norm(sd, p + ".layernorm1", x, 1e-5)
    x = x + torch_mha(sd, p + ".mha2", xn, enc, heads)
    xn = layer_norm(sd, p + ".layernorm2", x, 1e-5)
    f = torch.relu(xn @ sd[p + ".ffn.0.weight"].t() + sd[p + ".ffn.0.bias"]) @ sd[p + ".ffn.2.weight"].t() + sd[p + ".ffn.2.bias"]
    return layer_norm(sd, p + ".layernorm3", x + f, 1e-5)


def decoder_forward(sd, target_in, padding_mask, enc, num_layers=6, heads=8, prefix="decoder"):
    """Decoder.forward (m3ae_decoder.py:118-138).  `padding_mask`: True = real token, or None (generation)."""
    T = target_in.shape[1]
    emb = sd[prefix + ".target_embedding.weight"][target_in]
    t = emb + (emb + decoder_pe(T, emb.shape[-1]))          # `+=` of the positional-encoding OUTPUT (:127)
    causal = ~torch.tril(torch.ones(T, T, dtype=torch.bool))
    pad = None if padding_mask is None else ~padding_mask
    x = None
    for i in range(num_layers):                              # every layer reads `t` (:131-134)
        x = decoder_layer(sd, f"{prefix}.dec_layers.{i}", t, enc, pad, causal, heads)
    return x @ sd[prefix + ".final_linear.weight"].t() + sd[prefix + ".final_linear.bias"]


def decoder_inputs(target_tokens, sep_id, pad_id):
    """m3ae_decoder.py:344-351,362: input = tokens[:, :-1] with [SEP] -> [PAD]; golden = tokens[:, 1:]."""
    tin = target_tokens[:, :-1].clone()
    tin[tin == sep_id] = pad_id
    return tin, tin != pad_id, target_tokens[:, 1:]


def decoder_loss(sd, target_tokens, enc, sep_id=102, pad_id=0, **kw):
    tin, mask, gold = decoder_inputs(target_tokens, sep_id, pad_id)
    logits = decoder_forward(sd, tin, mask, enc, **kw)
    ce = F.cross_entropy(logits.transpose(1, 2), gold, ignore_index=pad_id)   # criterion (:226, :366)
    loss = (ce * mask).sum() / mask.sum()                                     # (:366-367)
    return loss, logits


def decoder_search(sd, enc, cls_id=101, sep_id=102, eos_id=None, pad_id=0, max_len=128, **kw):
    """Decoder.search_path (m3ae_decoder.py:141-182): greedy, the whole prefix is re-run every step (no cache)."""
    B = enc.shape[0]
    seq = torch.full((B, 1), cls_id, dtype=torch.long)
    finished = torch.zeros(B, dtype=torch.bool)
    for _ in range(max_len):
        nxt = decoder_forward(sd, seq, None, enc, **kw)[:, -1].argmax(-1)
        finished |= (nxt == sep_id) | ((nxt == eos_id) if eos_id is not None else torch.zeros(B, dtype=torch.bool))
        seq = torch.cat([seq, nxt[:, None]], dim=1)
        if finished.all():
            break
    seq = seq[:, 1:]
    for i in range(B):
        hit = torch.where((seq[i] == sep_id) | ((seq[i] == eos_id) if eos_id is not None else torch.zeros_like(seq[i], dtype=torch.bool)))[0]
        if len(hit) > 0:
            seq[i, hit[0] + 1:] = pad_id
    return F.pad(seq, (0, max_len - seq.shape[1]), value=pad_id)


# ------------------------------------------------------------------------------------------------
# T5 beam-search generation.  The reference calls HF `generate(num_beams=4, early_stopping=True,
# max_length=t5_max_length)` (m3ae_t5_mm_encoder_input.py:209-218,252-260; third party: transformers==4.6.0
# `GenerationMixin.beam_search` + `BeamSearchScorer`).  Restated from the 4.6.0 algorithm: log-softmax scores added to the
# running beam scores, top 2*beams candidates per sample, EOS candidates ranked inside the first `beams` become
# finished hypotheses scored sum_logprobs / len(prefix) ** length_penalty (prefix = tokens so far INCLUDING the decoder
# start token, EXCLUDING the EOS), a sample is done as soon as it holds `beams` hypotheses (early_stopping=True), open
# beams are added at the end, the best hypothesis per sample is returned (+ EOS when shorter than max_length), padded.
# The decoder re-runs the whole prefix every step here (no KV cache) -- same numbers, this is the checker.
# ------------------------------------------------------------------------------------------------
def t5_next_token_logits(sd, enc, prefix, heads):
    dec = t5_decoder(sd, prefix, enc, heads)
    return (dec[:, -1] * dec.shape[-1] ** -0.5) @ sd["t5.shared.weight"].t()


def t5_beam_search(sd, enc, heads, num_beams=4, max_length=12, eos_id=1, pad_id=0, start_id=0, length_penalty=1.0,
                   len_offset=0):
    """len_offset = 0: transformers 4.6.0 (the reference's pin) divides an OPEN beam's log-probability at max_length by its
    token count including the decoder start token; later releases (the one installed in the build container) exclude the
    start token there (len_offset = 1) -- used only to pin this restatement against that release's `generate`.
    EOS-terminated hypotheses are divided by the prefix length in both."""
    B = enc.shape[0]
    nb = num_beams
    enc_r = enc.repeat_interleave(nb, dim=0)
    ids = torch.full((B * nb, 1), start_id, dtype=torch.long)
    beam_scores = torch.zeros(B, nb)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    hyps = [[] for _ in range(B)]          # (score, tokens)
    done = [False] * B
    cur_len = 1
    while cur_len < max_length:
        logp = torch.log_softmax(t5_next_token_logits(sd, enc_r, ids, heads), dim=-1)
        V = logp.shape[-1]
        scores = (logp + beam_scores[:, None]).view(B, nb * V)
        top_s, top_i = torch.topk(scores, 2 * nb, dim=1, largest=True, sorted=True)
        nxt_scores = torch.zeros(B, nb)
        nxt_tokens = torch.full((B, nb), pad_id, dtype=torch.long)
        nxt_index = torch.zeros(B, nb, dtype=torch.long)
        for b in range(B):
            if done[b]:
                nxt_index[b] = b * nb
                continue
            k = 0
            for rank in range(2 * nb):
                tok, sc, src = int(top_i[b, rank] % V), float(top_s[b, rank]), b * nb + int(top_i[b, rank] // V)
                if tok == eos_id:
                    if rank >= nb:
                        continue
                    hyp = ids[src].clone()
                    hyps[b].append((sc / (hyp.shape[-1] ** length_penalty), hyp))
                    hyps[b] = sorted(hyps[b], key=lambda t: -t[0])[:nb]
                else:
                    nxt_scores[b, k], nxt_tokens[b, k], nxt_index[b, k] = sc, tok, src
                    k += 1
                if k == nb:
                    break
            done[b] = done[b] or len(hyps[b]) >= nb      # early_stopping=True
        beam_scores = nxt_scores.view(-1)
        ids = torch.cat([ids[nxt_index.view(-1)], nxt_tokens.view(-1, 1)], dim=1)
        cur_len += 1
        if all(done):
            break
    out = []
    for b in range(B):
        if not done[b]:
            for j in range(nb):
                hyp = ids[b * nb + j]
                hyps[b].append((float(beam_scores[b * nb + j]) / ((hyp.shape[-1] - len_offset) ** length_penalty), hyp))
            hyps[b] = sorted(hyps[b], key=lambda t: -t[0])[:nb]
        out.append(hyps[b][0][1])
    L = min(max(len(h) for h in out) + 1, max_length)
    seq = torch.full((B, L), pad_id, dtype=torch.long)
    for b, h in enumerate(out):
        seq[b, : len(h)] = h
        if len(h) < max_length:
            seq[b, len(h)] = eos_id
    return seq
