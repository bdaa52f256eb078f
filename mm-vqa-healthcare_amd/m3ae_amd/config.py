"""Config dict for the hot path: the reference's sacred defaults and named configs as plain dicts.

Mirrors reference m3ae/config.py:18-119 (defaults) and :121-282 (named configs); the model only
ever reads a plain dict (SURVEY.md 5 "Config / flags"), so sacred itself is not needed.  Named configs
compose left to right exactly like on the reference CLI (`... task_finetune_vqa_vqa_rad clip16
text_roberta image_size=384`, run_scripts/finetune_m3ae.sh:18-21); see `compose()` / `parse_cli()`.

Architecture widths that the reference derives from downloaded checkpoints (clip_model.py:277-290,
RobertaModel.from_pretrained) are looked up from the `vit` / `tokenizer` names here (`_ARCH`), and can be
overridden with explicit `vit_width/vit_layers/text_hidden/...` keys (used by the tiny test config).
"""
import ast
import copy


def _loss_names(d):
    ret = {"mlm": 0, "mim": 0, "itm": 0, "vqa": 0, "cls": 0, "irtr": 0}
    ret.update(d)
    return ret


DEFAULTS = dict(
    exp_name="meter", seed=0, datasets=["medicat", "roco"], loss_names=_loss_names({"itm": 1, "mlm": 1}),
    batch_size=4096,
    image_size=224, patch_size=32, draw_false_image=1, image_only=False,
    vqa_label_size=3129, mlc_label_size=14, max_text_len=40, tokenizer="bert-base-uncased", vocab_size=30522,
    whole_word_masking=True, mlm_prob=0.15, draw_false_text=0,
    num_top_layer=6, input_image_embed_size=768, input_text_embed_size=768, vit="ViT-B/32", hidden_size=768,
    num_heads=12, num_layers=6, mlp_ratio=4, drop_rate=0.1,
    mim_prob=0.75, mim_decoder_hidden_size=384, mim_decoder_num_layers=4, mim_decoder_num_heads=6,
    norm_pix_loss=True, mim_layer=-1,
    optim_type="adamw", learning_rate=1e-5, weight_decay=0.01, decay_power=1, max_epoch=100, max_steps=-1,
    warmup_steps=10000, end_lr=0, lr_multiplier_head=5, lr_multiplier_multi_modal=5,
    mm_encoder_inputs_include_cls_feats=True, mm_encoder_inputs_include_imagetext_feats=False,
    mm_encoder_inputs_mm_feats_width=0,
    t5_model_name="t5-small", t5_max_length=25, t5_generation=True,
    unfreeze_num_encoder_layers=2, unfreeze_num_decoder_layers=2,
    get_recall_metric=False, resume_from=None, fast_dev_run=False, val_check_interval=1.0, test_only=False,
    default_root_dir="checkpoints", data_root="", log_dir="result", per_gpu_batchsize=0, use_ddp=False,
    num_gpus=1, num_nodes=1, load_path="", decoder_load_path="", load_path_t5="", num_workers=8, precision=32,
    gpu_device_number=0, label_column_name="",
    # build extensions (not in the reference): compute mode of the HIP path
    compute_dtype="bf16",  # "bf16" (perf mode) | "fp32" (parity mode)
)

NAMED = {
    "task_pretrain_m3ae": dict(
        exp_name="task_pretrain_m3ae", datasets=["medicat", "roco"],
        loss_names=_loss_names({"itm": 1, "mlm": 1, "mim": 1}), batch_size=256, max_epoch=10, max_steps=100000,
        warmup_steps=0.1, whole_word_masking=True, vocab_size=30522, max_text_len=64, image_size=224,
        tokenizer="bert-base-uncased", learning_rate=1e-5, val_check_interval=1.0, lr_multiplier_head=5,
        lr_multiplier_multi_modal=5, num_top_layer=6, hidden_size=768, num_heads=12, precision=16, mim_layer=3),
    "task_finetune_vqa_vqa_rad": dict(
        exp_name="task_finetune_vqa_vqa_rad", datasets=["vqa_vqa_rad"], loss_names=_loss_names({"vqa": 1}),
        batch_size=64, max_epoch=20, max_steps=1000, warmup_steps=0.1, draw_false_image=0, learning_rate=1e-5,
        val_check_interval=1.0, lr_multiplier_head=100, lr_multiplier_multi_modal=5,
        tokenizer="bert-base-uncased", input_text_embed_size=768, vit="ViT-B/32", input_image_embed_size=768,
        image_size=576, vqa_label_size=498, max_text_len=32),
    "task_finetune_vqa_ehr_xqa": dict(
        exp_name="task_finetune_vqa_ehr_xqa", datasets=["vqa_ehr_xqa"], loss_names=_loss_names({"vqa": 1}),
        batch_size=64, max_epoch=50, max_steps=1000, warmup_steps=0.1, draw_false_image=0, learning_rate=5e-6,
        val_check_interval=1.0, lr_multiplier_head=100, lr_multiplier_multi_modal=5,
        tokenizer="bert-base-uncased", input_text_embed_size=768, vit="ViT-B/32", input_image_embed_size=768,
        image_size=576, vqa_label_size=498, max_text_len=32),
    "clip32": dict(vit="ViT-B/32", image_size=224, patch_size=32, input_image_embed_size=768),
    "clip16": dict(vit="ViT-B/16", image_size=224, patch_size=16, input_image_embed_size=768),
    "clip16_large": dict(vit="ViT-L/16", image_size=224, patch_size=16, input_image_embed_size=1024),  # extension
    "text_roberta": dict(tokenizer="roberta-base", vocab_size=50265, input_text_embed_size=768),
    "text_roberta_large": dict(tokenizer="roberta-large", vocab_size=50265, input_text_embed_size=1024),
}

# widths the reference reads out of downloaded checkpoints
_ARCH_VIT = {
    "ViT-B/32": dict(vit_width=768, vit_layers=12),
    "ViT-B/16": dict(vit_width=768, vit_layers=12),
    "ViT-L/14": dict(vit_width=1024, vit_layers=24),
    "ViT-L/16": dict(vit_width=1024, vit_layers=24),
}
_ARCH_TEXT = {
    "base": dict(text_hidden=768, text_layers=12, text_heads=12, text_inter=3072),
    "large": dict(text_hidden=1024, text_layers=24, text_heads=16, text_inter=4096),
}


def resolve_arch(cfg):
    """Fill vit_width/vit_layers/text_* from the `vit` / `tokenizer` names unless given explicitly."""
    cfg = dict(cfg)
    vit = _ARCH_VIT.get(cfg["vit"], _ARCH_VIT["ViT-B/16"])
    for k, v in vit.items():
        cfg.setdefault(k, v)
    text = _ARCH_TEXT["large" if "large" in cfg["tokenizer"] else "base"]
    for k, v in text.items():
        cfg.setdefault(k, v)
    cfg.setdefault("text_max_pos", 514 if "roberta" in cfg["tokenizer"] else 512)
    return cfg


def compose(*parts, **overrides):
    """compose("task_finetune_vqa_vqa_rad", "clip16", "text_roberta", image_size=384)."""
    cfg = copy.deepcopy(DEFAULTS)
    for p in parts:
        cfg.update(copy.deepcopy(NAMED[p]) if isinstance(p, str) else p)
    cfg.update(overrides)
    return resolve_arch(cfg)


def parse_cli(argv):
    """Parse the sacred grammar of run_scripts/*.sh: `with k=v ... named_config ... k=v`.  As in sacred, named configs
    apply first (left to right) and explicit `k=v` updates win over them wherever they stand on the command line
    (finetune_m3ae.sh passes `max_epoch=70` BEFORE `task_finetune_vqa_vqa_rad`, whose own max_epoch is 20)."""
    args = [a for a in argv if a != "with"]
    cfg = copy.deepcopy(DEFAULTS)
    for a in args:
        if "=" not in a:
            if a not in NAMED:
                raise KeyError(f"unknown named config {a!r}")
            cfg.update(copy.deepcopy(NAMED[a]))
    for a in args:
        if "=" in a:
            k, v = a.split("=", 1)
            try:
                v = ast.literal_eval(v)
            except (ValueError, SyntaxError):
                pass
            cfg[k] = v
    return resolve_arch(cfg)


def finetune_vqa_rad_config(**over):
    """The effective config of run_scripts/test_m3ae.sh / finetune_m3ae.sh (BASELINE configs[0..2])."""
    return compose("task_finetune_vqa_vqa_rad", "clip16", "text_roberta", image_size=384, **over)


def tiny_config(**over):
    """Reduced-size model of the same architecture used by the golden fixtures (oracle/make_golden.py TINY)."""
    base = dict(image_size=64, hidden_size=128, num_heads=2, num_top_layer=2, input_image_embed_size=128,
                input_text_embed_size=128, vocab_size=1000, vit_width=128, vit_layers=3, text_hidden=128,
                text_layers=2, text_heads=2, text_inter=512)
    base.update(over)
    return compose("task_finetune_vqa_vqa_rad", "clip16", "text_roberta", **base)
