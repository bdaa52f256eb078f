"""M3AETransformerSS on MI355X -- the drop-in boundary of the hot path.

Same constructor (`M3AETransformerSS(config: dict)`), same `infer()` / `forward()` / `training_step()` contract,
same module tree and state_dict key names / shapes as the reference's m3ae/modules/m3ae_module.py:16-373, so
`load_state_dict(ckpt["state_dict"], strict=False)` of an upstream M3AE checkpoint works and the generator wrappers
(`self.m3ae.infer(batch)`, m3ae_t5_mm_encoder_input.py:102) are served unchanged.  Below the modules every FLOP
runs in hand-written gfx950 kernels (libm3ae_hip.so) -- there is no PyTorch-op fallback.

Differences that are deliberate: weights are never downloaded (architecture comes from the config);
`finalize(device)` moves the parameters into the flat MI355X layout (ParamStore) and must be called before the
first forward; attention probabilities are not materialised (`output_attentions` is rejected); dropout
(`module.training`, RoBERTa p = 0.1, fusion layers p = `drop_rate`) uses the library's counter-hash masks fused into
the kernels (seeded by `ops.set_dropout_seed`), not torch's Philox stream.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from ..config import resolve_arch
from ..param_store import ParamStore
from . import objectives, prediction_heads
from .bert_model import BertCrossLayer, RobertaModel
from .clip_model import adapt_position_encoding, build_model

class _JoinAtEndFn(torch.autograd.Function):
    """Identity.  Its backward (among the first nodes of a backward pass) queues `main.wait_stream(side)` to run when the pass ends."""

    @staticmethod
    def forward(ctx, x, main, side):
        ctx.streams = (main, side)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        main, side = ctx.streams

        def join():
            # the in-place weight gradients are invisible to autograd's own end-of-backward synchronisation: whoever reads
            # store.grad next does so on the stream that is current NOW (normally `main`; it may differ from the forward's)
            main.wait_stream(side)
            cur = torch.cuda.current_stream()
            if cur != main:
                cur.wait_stream(main)
                cur.wait_stream(side)

        torch.autograd.Variable._execution_engine.queue_callback(join)
        return g, None, None


try:  # drop-in for pl.Trainer when Lightning is installed on the user's side
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None
    _Base = nn.Module


class _HParams(dict):
    __getattr__ = dict.__getitem__


def init_weights(module):
    """m3ae_utils.py:101-109."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=0.02)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


class M3AETransformerSS(_Base):
    def __init__(self, config):
        super().__init__()
        config = resolve_arch(config)
        if pl is None:
            self.hparams = _HParams(config=config)
        else:
            self.save_hyperparameters()
        cfg = config
        hs = cfg["hidden_size"]
        self.is_clip = "swin" not in cfg["vit"]
        if not self.is_clip:
            raise NotImplementedError("swin backbones are unreachable in the reference (SURVEY 2 #14)")
        # == 1. Build Models (m3ae_module.py:21-89) ==
        self.vision_encoder = build_model(cfg["vit"], resolution_after=cfg["image_size"], vision_width=cfg["vit_width"],
                                          vision_layers=cfg["vit_layers"], patch_size=cfg["patch_size"])
        self.language_encoder = RobertaModel(cfg["vocab_size"], cfg["text_hidden"], cfg["text_layers"],
                                             cfg["text_heads"], cfg["text_inter"], cfg["text_max_pos"])
        self.multi_modal_language_proj = nn.Linear(cfg["input_text_embed_size"], hs)
        self.multi_modal_vision_proj = nn.Linear(cfg["input_image_embed_size"], hs)
        self.modality_type_embeddings = nn.Embedding(2, hs)
        inter = hs * cfg["mlp_ratio"]
        self.multi_modal_vision_layers = nn.ModuleList(
            [BertCrossLayer(hs, cfg["num_heads"], inter, drop_rate=cfg["drop_rate"]) for _ in range(cfg["num_top_layer"])])
        self.multi_modal_language_layers = nn.ModuleList(
            [BertCrossLayer(hs, cfg["num_heads"], inter, drop_rate=cfg["drop_rate"]) for _ in range(cfg["num_top_layer"])])
        self.multi_modal_vision_pooler = prediction_heads.Pooler(hs)
        self.multi_modal_language_pooler = prediction_heads.Pooler(hs)
        for m in (self.multi_modal_language_proj, self.multi_modal_vision_proj, self.modality_type_embeddings,
                  self.multi_modal_vision_layers, self.multi_modal_language_layers, self.multi_modal_vision_pooler,
                  self.multi_modal_language_pooler):
            m.apply(init_weights)
        # == 2. Pre-training heads (m3ae_module.py:91-101) ==
        if cfg["loss_names"]["mlm"] > 0:
            self.mlm_head = prediction_heads.MLMHead(hs, cfg["vocab_size"])
            self.mlm_head.apply(init_weights)
        if cfg["loss_names"]["mim"] > 0:
            self.mim_head = prediction_heads.MIMHead(cfg)
            self.mim_head.apply(init_weights)
        if cfg["loss_names"]["itm"] > 0 or cfg["loss_names"]["irtr"] > 0:
            self.itm_head = prediction_heads.ITMHead(hs * 2)
            self.itm_head.apply(init_weights)
        # == 3. Load (m3ae_module.py:103-114) ==
        if cfg["load_path"] != "" and not cfg["test_only"]:
            self._load(cfg["load_path"])
        # == 4. Downstream heads (m3ae_module.py:116-126) ==
        if cfg["loss_names"]["vqa"] > 0:
            vs = cfg["vqa_label_size"]
            self.vqa_head = nn.Sequential(nn.Linear(hs * 2, hs * 2), nn.LayerNorm(hs * 2), nn.GELU(),
                                          nn.Linear(hs * 2, vs))
            self.vqa_head.apply(init_weights)
        self.current_tasks = list()
        # == 5. Load for testing (m3ae_module.py:132-142) ==
        if cfg["load_path"] != "" and cfg["test_only"]:
            self._load(cfg["load_path"])
        self.store = None
        self._dtype = torch.bfloat16 if cfg.get("compute_dtype", "bf16") == "bf16" else torch.float32

    # ------------------------------------------------------------------------------------------------------
    def _load(self, path):
        # upstream checkpoints are Lightning pickles (callbacks / hyper_parameters hold class globals): the weights-only
        # unpickler of torch >= 2.6 rejects them, as it would for the reference's own torch.load (m3ae_module.py:105)
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
        sd = adapt_position_encoding(ckpt["state_dict"], after=self.hparams.config["image_size"],
                                     patch_size=self.hparams.config["patch_size"])
        self.load_state_dict(sd, strict=False)
        if getattr(self, "store", None) is not None:   # also reachable from __init__ (load_path), before finalize()
            self.store.sync_shadows()

    def weight_units(self):
        u = self.vision_encoder.weight_units() + self.language_encoder.weight_units()
        u += [self.multi_modal_language_proj.weight, self.multi_modal_vision_proj.weight]
        for l in list(self.multi_modal_vision_layers) + list(self.multi_modal_language_layers):
            u += l.weight_units()
        u += self.multi_modal_vision_pooler.weight_units() + self.multi_modal_language_pooler.weight_units()
        for name in ("mlm_head", "mim_head", "itm_head"):
            if hasattr(self, name):
                u += getattr(self, name).weight_units()
        if hasattr(self, "vqa_head"):
            u += [self.vqa_head[0].weight, self.vqa_head[3].weight]
        return u

    def finalize(self, device="cuda", compute_dtype=None, frozen=()):
        """Move parameters into the flat MI355X layout (ParamStore).  Call once, after loading weights."""
        if compute_dtype is not None:
            self._dtype = compute_dtype
        for b_name, b in list(self.named_buffers()):
            b.data = b.data.to(device)
        self.store = ParamStore(self, self.hparams.config, device, self._dtype, self.weight_units, frozen=frozen)
        return self

    # ------------------------------------------------------------------------------------------------------
    two_streams = os.environ.get("M3AE_TWO_STREAMS", "1") == "1"   # M3AE_TWO_STREAMS=0: everything on the caller's stream
    _side_stream = None

    def _side(self):
        if self._side_stream is None:
            # default priority (a high-priority side stream measured the same in round 3; M3AE_SIDE_STREAM_PRIORITY: A/B runs only)
            prio = os.environ.get("M3AE_SIDE_STREAM_PRIORITY")
            self._side_stream = torch.cuda.Stream() if prio is None else torch.cuda.Stream(priority=int(prio))
            # gradients of directly-used leaves (e.g. the modality type embeddings) arrive from nodes of either stream: intended
            warn_off = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if warn_off is not None:
                warn_off(False)
        return self._side_stream

    def _fusion_two_streams(self, text_tower, v, mt, mv, mask_image, ret, main, side, ev_inputs):
        """The text tower and the text half of every fusion layer on a second HIP stream.  The towers are independent until the
        fusion layers and a fusion layer's two halves read only the PREVIOUS layer's outputs, so the short text-side kernels
        (B*32 rows: half-empty grids) fill the tail rounds of the image-side launches instead of running alone.  Events order
        the hand-overs; `record_stream` keeps the caching allocator from recycling a tensor while the other stream still reads
        it.  autograd runs every node's backward on the stream of its forward and inserts the same hand-overs in reverse."""
        side.wait_event(ev_inputs)
        mt.record_stream(side)
        self.store.streams.update((main, side))
        with torch.cuda.stream(side):
            x = text_tower()
            ev_x = side.record_event()
        y = v
        ev_y = main.record_event()
        for layer_idx, (text_layer, image_layer) in enumerate(zip(self.multi_modal_language_layers,
                                                                  self.multi_modal_vision_layers)):
            if mask_image and self.hparams.config["mim_layer"] == layer_idx:
                ret[f"multi_modal_text_feats_{layer_idx}"], ret[f"multi_modal_image_feats_{layer_idx}"] = x, y
            # x and y each feed both layers of the pair: forked on the stream that produced them, so the two gradients are joined
            # there by the library's add (ops.Fork2Fn)
            with torch.cuda.stream(side):
                xa, xb = ops.fork2(x)
            ya, yb = ops.fork2(y)
            side.wait_event(ev_y)
            y.record_stream(side)
            with torch.cuda.stream(side):
                x1 = text_layer(xa, yb, mt, mv)
                ev_x1 = side.record_event()
            main.wait_event(ev_x)
            x.record_stream(main)
            y1 = image_layer(ya, xb, mv, mt)
            ev_y = main.record_event()
            x, y, ev_x = x1, y1, ev_x1
        main.wait_event(ev_x)
        x.record_stream(main)
        if torch.is_grad_enabled():
            # whichever of the two runs in backward queues the end-of-backward join: when loss.backward() returns, the caller's
            # stream has waited for the side stream (the weight gradients land in the flat buffer as side effects of the nodes,
            # not as autograd outputs, so the engine's own end-of-backward synchronisation does not cover them)
            x, y = _JoinAtEndFn.apply(x, main, side), _JoinAtEndFn.apply(y, main, side)
        return x, y

    def random_masking(self, x, mask_ratio, noise=None):
        """m3ae_module.py:153-183.  The argsort / argsort-of-argsort / gather-of-ones bookkeeping is one kernel
        (`m3ae_mask_ranks`: ranks by counting); the row gather and its gradient run in the library too."""
        B, Lp1, D = x.shape
        L = Lp1 - 1
        len_keep = int(L * (1 - mask_ratio))
        if noise is None:
            noise = torch.rand(B, L, device=x.device)
        ids_restore, keep_rows, mask = ops.mask_ranks(noise, len_keep)
        pos = self.vision_encoder.visual.positional_embedding
        xp = x + pos.to(x.dtype)  # x + pos, cls row included (m3ae_module.py:169,180); pretrain-only glue
        x_masked = ops.gather_rows(xp, keep_rows).view(B, len_keep + 1, D)
        return x_masked, mask, ids_restore

    def patchify(self, imgs):
        """m3ae_module.py:185-192 (`m3ae_mim_targets` without the standardisation)."""
        return ops.mim_targets(imgs, self.hparams.config["patch_size"], False)

    def infer(self, batch, mask_text=False, mask_image=False, image_token_type_idx=1, img=None,
              output_attentions=False, unimodal=False):
        """m3ae_module.py:203-312."""
        if self.store is None:
            raise RuntimeError("call finalize(device) before the first forward")
        if output_attentions:
            raise NotImplementedError("attention probabilities are never materialised on this path")
        ret = dict()
        if img is None:
            img_key = f"image_{image_token_type_idx - 1}" if f"image_{image_token_type_idx - 1}" in batch else "image"
            img = batch[img_key][0]
        do_mlm = "_mlm" if mask_text else ""
        text_ids = batch[f"text_ids{do_mlm}"]
        text_labels = batch[f"text_labels{do_mlm}"]
        text_masks = batch["text_masks"]
        dt = self._dtype
        H = self.hparams.config["num_heads"]
        type_emb = self.modality_type_embeddings.weight
        mt = self.language_encoder.get_extended_attention_mask(text_masks).contiguous()
        ev_inputs = None
        if self.two_streams and img.is_cuda:   # the text tower (side stream) may start as soon as the inputs are there
            ev_inputs = torch.cuda.current_stream().record_event()
        # == Image Encoding (m3ae_module.py:238-256) ==
        if mask_image:
            v = self.vision_encoder.forward_patch_embed(img, dt)
            v, mim_masks, mim_ids_restore = self.random_masking(v, self.hparams.config["mim_prob"],
                                                                batch.get("mim_noise"))
            v = self.vision_encoder.forward_trans(v)
            ret["mim_masks"], ret["mim_ids_restore"] = mim_masks, mim_ids_restore
        else:
            v = self.vision_encoder(img, dt)
        v = ops.linear(v, self.multi_modal_vision_proj.weight, self.multi_modal_vision_proj.bias,
                       extra_bias=type_emb[image_token_type_idx])
        # == Text Encoding (m3ae_module.py:229-236) ==
        # The two towers are independent until the fusion layers.  The reference runs the text tower first; here it runs SECOND, so
        # that autograd (latest-created nodes first) runs its short backward FIRST: the 154-MB word-embedding gradient is then
        # complete ~5 % into backward and its all-reduce overlaps the image tower's backward, and the last gradients of the step
        # are the image tower's first parameters -- the small tail bucket of ddp.FlatGradReducer.  Same values either way (only the
        # order in which the dropout sites draw their seeds changes).
        side = self._side() if (self.two_streams and img.is_cuda) else None
        main = torch.cuda.current_stream() if side is not None else None
        mv = None  # all-ones image mask -> additive zeros (m3ae_module.py:253-256)

        def text_tower():
            if side is not None:
                text_ids.record_stream(side)
            t = self.language_encoder.embeddings(text_ids, dt)
            for layer in self.language_encoder.encoder.layer:
                t = layer(t, mt)
            # projection + modality type embedding (m3ae_module.py:235,260-263): the type row rides in the GEMM bias
            return ops.linear(t, self.multi_modal_language_proj.weight, self.multi_modal_language_proj.bias,
                              extra_bias=type_emb[0])

        if side is None:
            t = text_tower()
            # == Multi-Modal Fusion (m3ae_module.py:266-285): both streams read the PRE-update x, y ==
            x, y = t, v
            for layer_idx, (text_layer, image_layer) in enumerate(zip(self.multi_modal_language_layers,
                                                                      self.multi_modal_vision_layers)):
                if mask_image and self.hparams.config["mim_layer"] == layer_idx:
                    ret[f"multi_modal_text_feats_{layer_idx}"], ret[f"multi_modal_image_feats_{layer_idx}"] = x, y
                (xa, xb), (ya, yb) = ops.fork2(x), ops.fork2(y)
                x1 = text_layer(xa, yb, mt, mv)
                y1 = image_layer(ya, xb, mv, mt)
                x, y = x1, y1
        else:
            x, y = self._fusion_two_streams(text_tower, v, mt, mv, mask_image, ret, main, side, ev_inputs)
        # == Output (m3ae_module.py:287-297) ==
        cls_t = self.multi_modal_language_pooler(x)
        cls_v = self.multi_modal_vision_pooler(y)
        cls = torch.cat([cls_t, cls_v], dim=-1)
        ret.update({
            "images": img,
            "text_labels": text_labels,
            "text_ids": text_ids,
            "text_masks": text_masks,
            "extended_image_masks": mv,
            "extended_text_masks": mt,
            "multi_modal_text_feats": x,
            "multi_modal_image_feats": y,
            "multi_modal_cls_feats": cls,
        })
        if mask_image:  # only MIM needs it (the reference recomputes it on every call, m3ae_module.py:301)
            ret["patched_images"] = self.patchify(img)
        ret["attentions"] = None
        return ret

    def vqa_head_forward(self, cls):
        """m3ae_module.py:120-125: Linear -> LayerNorm -> GELU (fused into the LN kernel) -> Linear."""
        h = ops.linear(cls, self.vqa_head[0].weight, self.vqa_head[0].bias)
        ln = self.vqa_head[1]
        h = ops.layer_norm(h, ln.weight, ln.bias, ln.eps, act=ops.ACT_GELU)
        # 498 answers: N % 4 != 0 / K % 64 != 0 would send forward, dgrad and wgrad to the fp32-FMA generic kernel in perf mode
        # (6 launches per step at 3 TFLOP/s); the zero-padded operand copies of ops.vocab_linear keep them on the MFMA kernels
        return ops.vocab_linear(h, self.vqa_head[3].weight, self.vqa_head[3].bias)

    def forward(self, batch, test=False):
        """m3ae_module.py:314-345."""
        ret = dict()
        if len(self.current_tasks) == 0:
            ret.update(self.infer(batch))
            return ret
        if "mlm" in self.current_tasks:
            ret.update(objectives.compute_mlm(self, batch))
        if "mim" in self.current_tasks:
            ret.update(objectives.compute_mim(self, batch))
        if "itm" in self.current_tasks:
            ret.update(objectives.compute_itm(self, batch, batch.get("itm_labels")))
        if "vqa" in self.current_tasks:
            ret.update(objectives.compute_vqa_m3ae(self, batch, test=test))
        return ret

    def set_task(self):
        """m3ae_utils.py:95-97."""
        self.current_tasks = [k for k, v in self.hparams.config["loss_names"].items() if v > 0]

    def training_step(self, batch, batch_idx=0):
        """m3ae_module.py:347-353."""
        self.set_task()
        output = self(batch)
        return sum([v * self.hparams.config["loss_names"][k.replace("_loss", "")]
                    for k, v in output.items() if k.endswith("_loss")])

    def configure_optimizers(self):
        """m3ae_module.py:372-373 -> m3ae_utils.set_schedule (:112-242): `([optimizer], [{"scheduler", "interval": "step"}])`.
        The optimizer is a torch.optim.Optimizer (six param groups) whose step() is the fused AdamW over the flat buffers;
        the scheduler is a LambdaLR with the reference's polynomial (or cosine) decay with warm-up.  max_steps comes from
        the attached trainer when there is one (m3ae_utils.py:212-219), else from the config."""
        max_steps = None
        tr = getattr(self, "_trainer", None) or getattr(self, "trainer_ref", None)
        if tr is not None and getattr(tr, "max_steps", None) not in (None, -1):
            max_steps = tr.max_steps
        return self.store.make_optimizer(max_steps)


def state_dict_spec(config):
    """{name: shape} of the module's state_dict (== the reference's key names / shapes)."""
    with torch.device("meta"):
        m = M3AETransformerSS(dict(config, load_path=""))
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}
