"""Import harness for the UNMODIFIED reference (better62/MM-VQA-Healthcare at /root/reference).

TEST INFRASTRUCTURE ONLY.  This file runs in the build container (where /root/reference is
mounted) to generate the golden fixtures under tests/golden/ (see oracle/make_golden.py).
It never runs on the GPU box (the reference cannot travel) and nothing in the product path
imports it.

The reference cannot be imported as shipped: missing third-party packages
(pytorch_lightning, torchmetrics, nltk, rouge_score, timm, sacred), transformers 4.6 -> 5.x API
drift, a missing source file (m3ae/modules/__init__.py:4) and network loaders.  Everything
below is an in-process stub; no reference file is modified or copied.
"""
import os
import sys
import types

REF_ROOT = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Install the stubs and make `import m3ae` resolve to the reference. Idempotent."""
    if "m3ae" in sys.modules and getattr(sys.modules["m3ae"], "_graft_shimmed", False):
        return
    import numpy as np
    import torch
    import torch.nn as nn
    import transformers  # must come first
    import transformers.file_utils as fu
    import transformers.modeling_utils as mu
    import transformers.optimization as topt
    import transformers.pytorch_utils as pu

    if not hasattr(np, "float"):
        np.float = float  # position_embeddings.py:57

    # --- pytorch_lightning -------------------------------------------------------------
    class _HParams(dict):
        __getattr__ = dict.__getitem__

    class LightningModule(nn.Module):
        def save_hyperparameters(self):
            import inspect
            frame = inspect.currentframe().f_back
            loc = {k: v for k, v in frame.f_locals.items() if k not in ("self", "__class__")}
            self.hparams = _HParams(loc)

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    pl = _stub("pytorch_lightning", LightningModule=LightningModule)
    pl.seed_everything = lambda s: torch.manual_seed(s)

    # --- torchmetrics ------------------------------------------------------------------
    class Metric(nn.Module):
        def __init__(self, dist_sync_on_step=False):
            super().__init__()
            self._defaults = {}

        def add_state(self, name, default, dist_reduce_fx=None):
            if isinstance(default, torch.Tensor):
                self.register_buffer(name, default.clone())
            else:
                setattr(self, name, default)
            self._defaults[name] = default

        def forward(self, *a, **k):
            try:  # string metrics are out of scope; never let them break a step
                self.update(*a, **k)
                return self.compute()
            except Exception:
                return torch.tensor(0.0)

        def reset(self):
            for k, v in self._defaults.items():
                setattr(self, k, v.clone() if isinstance(v, torch.Tensor) else v)

    _stub("torchmetrics", Metric=Metric)

    # --- string-metric deps (never exercised for parity) -------------------------------
    _stub("nltk")
    _stub("nltk.translate")
    _stub("nltk.translate.bleu_score", sentence_bleu=lambda *a, **k: 0.0,
          SmoothingFunction=lambda: types.SimpleNamespace(method1=None))

    class _Scorer:
        def __init__(self, *a, **k):
            pass

        def score(self, a, b):
            s = types.SimpleNamespace(fmeasure=0.0, recall=0.0, precision=0.0)
            return {"rouge1": s, "rouge2": s}

    _stub("rouge_score", rouge_scorer=types.SimpleNamespace(RougeScorer=_Scorer))
    _stub("rouge_score.rouge_scorer", RougeScorer=_Scorer)

    # --- timm (names only; swin path is dead) ----------------------------
    ident = lambda *a, **k: (lambda f: f)
    _stub("timm")
    _stub("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406), IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225),
          IMAGENET_INCEPTION_MEAN=(0.5,) * 3, IMAGENET_INCEPTION_STD=(0.5,) * 3)
    _stub("timm.models")
    _stub("timm.models.helpers", build_model_with_cfg=None, overlay_external_default_cfg=None,
          load_state_dict=None, adapt_input_conv=None, load_custom_pretrained=None)
    _stub("timm.models.layers", PatchEmbed=nn.Identity, Mlp=nn.Identity, DropPath=nn.Identity,
          to_2tuple=lambda x: (x, x), trunc_normal_=lambda *a, **k: None, Conv2dSame=nn.Conv2d,
          Linear=nn.Linear)
    _stub("timm.models.registry", register_model=lambda f: f)
    _stub("timm.models.vision_transformer", checkpoint_filter_fn=None, _init_vit_weights=None)
    _stub("timm.models.features", FeatureListNet=None, FeatureDictNet=None, FeatureHookNet=None)
    _stub("timm.models.hub", has_hf_hub=lambda *a, **k: False, download_cached_file=None,
          load_state_dict_from_hf=None, load_state_dict_from_url=None)

    # --- transformers 4.6 -> 5.x drift --------------------------------------------------
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer

    def _no_prune(*a, **k):
        raise NotImplementedError("head pruning is not on the hot path")

    mu.find_pruneable_heads_and_indices = _no_prune
    for n in ("add_code_sample_docstrings", "add_start_docstrings", "add_start_docstrings_to_model_forward",
              "replace_return_docstrings", "add_end_docstrings"):
        setattr(fu, n, ident)
    topt.AdamW = torch.optim.AdamW

    # --- the package itself: the reference's __init__ imports a file that does not exist
    sys.path.insert(0, REF_ROOT)
    _stub("m3ae.modules.m3ae_t5_text_encoder_input", T5VQA_TextEncoderInput=None)
    import m3ae  # noqa: F401
    sys.modules["m3ae"]._graft_shimmed = True


def build_reference_model(config, vision_layers=12, vision_width=768, text_layers=12, text_hidden=768,
                          text_heads=12, text_inter=3072, vocab=50265):
    """Construct the reference's M3AETransformerSS with the network loaders replaced by local,
    random-init constructors of the same architecture."""
    install()
    import torch
    from transformers import RobertaConfig, RobertaModel
    import m3ae.modules.m3ae_module as mm
    import m3ae.gadgets.my_metrics as my_metrics
    from m3ae.modules.vision_encoders.clip_model import CLIP

    def local_build_model(name, resolution_after=224, jit=False):
        return CLIP(embed_dim=512, image_resolution=224, vision_layers=vision_layers, vision_width=vision_width,
                    vision_patch_size=config["patch_size"], context_length=77, vocab_size=49408,
                    transformer_width=512, transformer_heads=8, transformer_layers=12,
                    resolution_after=resolution_after)

    class _LocalRoberta:
        @staticmethod
        def from_pretrained(name, *a, **k):
            cfg = RobertaConfig(vocab_size=vocab, hidden_size=text_hidden, num_hidden_layers=text_layers,
                                num_attention_heads=text_heads, intermediate_size=text_inter,
                                max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                                pad_token_id=1, bos_token_id=0, eos_token_id=2,
                                hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
            cfg._attn_implementation = "eager"
            m = RobertaModel(cfg)
            # 4.6.0 semantics at the reference's two call sites (m3ae_module.py:232,234,255)
            m.get_extended_attention_mask = (
                lambda mask, shape, device=None: (1.0 - mask[:, None, None, :].to(torch.float32)) * -10000.0)
            for layer in m.encoder.layer:
                orig = layer.forward

                def fwd(*a, __orig=orig, **k):
                    out = __orig(*a, **k)
                    return out if isinstance(out, tuple) else (out,)

                layer.forward = fwd
            return m

    mm.build_model = local_build_model
    mm.RobertaModel = _LocalRoberta

    class _Tok:
        @staticmethod
        def from_pretrained(*a, **k):
            return types.SimpleNamespace(tokenize=lambda s: s.split())

    my_metrics.BertTokenizerFast = _Tok
    return mm.M3AETransformerSS(config)


def reference_config(**over):
    """config.py:18-119 defaults + task_finetune_vqa_vqa_rad (:175-198) + clip16 (:244-251) +
    text_roberta (:255-259) + image_size=384 (run_scripts/test_m3ae.sh) as a plain dict (sacred absent)."""
    cfg = dict(
        exp_name="task_finetune_vqa_vqa_rad", seed=0, datasets=["vqa_vqa_rad"],
        loss_names={"mlm": 0, "mim": 0, "itm": 0, "vqa": 1, "cls": 0, "irtr": 0}, batch_size=64,
        image_size=384, patch_size=16, draw_false_image=0, image_only=False,
        vqa_label_size=498, mlc_label_size=14, max_text_len=32, tokenizer="roberta-base", vocab_size=50265,
        whole_word_masking=True, mlm_prob=0.15, draw_false_text=0,
        num_top_layer=6, input_image_embed_size=768, input_text_embed_size=768, vit="ViT-B/16",
        hidden_size=768, num_heads=12, num_layers=6, mlp_ratio=4, drop_rate=0.1,
        mim_prob=0.75, mim_decoder_hidden_size=384, mim_decoder_num_layers=4, mim_decoder_num_heads=6,
        norm_pix_loss=True, mim_layer=-1,
        optim_type="adamw", learning_rate=1e-5, weight_decay=0.01, decay_power=1, max_epoch=20, max_steps=1000,
        warmup_steps=0.1, end_lr=0, lr_multiplier_head=100, lr_multiplier_multi_modal=5,
        get_recall_metric=False, test_only=False, load_path="", precision=32,
    )
    cfg.update(over)
    return cfg


def chdir_ref():
    os.chdir(REF_ROOT)  # objectives.py:180 opens label2ans.json relative to cwd
