"""Helpers shared by the parity tests: build the oracle's state_dict / config for the golden configurations."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mm-vqa-healthcare_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from m3ae_amd import synth  # noqa: E402
from m3ae_amd.config import tiny_config, finetune_vqa_rad_config  # noqa: E402
from m3ae_amd.modules.m3ae_module import state_dict_spec  # noqa: E402


def oracle_cfg(cfg):
    return dict(num_heads=cfg["num_heads"], vit_heads=cfg["vit_width"] // 64, text_heads=cfg["text_heads"],
                num_top_layer=cfg["num_top_layer"], patch_size=cfg["patch_size"], mim_prob=cfg["mim_prob"],
                mim_layer=cfg["mim_layer"])


def make_sd(cfg, requires_grad=False):
    """Reference-named fp32 CPU state_dict filled by the deterministic generator."""
    sd = {}
    for name, shape in state_dict_spec(cfg).items():
        sd[name] = torch.empty(shape, dtype=torch.float32)
    synth.fill_deterministic(sd)
    if requires_grad:
        for t in sd.values():
            t.requires_grad_(True)
    return sd


def tiny_batch(pretrain=False):
    return synth.synthetic_batch(2, text_len=32, image_size=64, vocab_size=1000, rank=0, pretrain=pretrain)


def full_batch(B=2):
    return synth.synthetic_batch(B, text_len=32, image_size=384, vocab_size=50265, rank=0)


def large1_config(**over):
    """configs[4] tower dimensions at reduced depth (oracle/make_golden.py LARGE1): one ViT-L/16 block (width 1024, 16 heads,
    512 x 512 -> 1025 tokens), one RoBERTa-large layer, one co-attention layer pair."""
    from m3ae_amd.config import compose
    base = dict(image_size=512, num_top_layer=1, input_image_embed_size=1024, input_text_embed_size=1024, vocab_size=1000,
                vit="ViT-L/16", tokenizer="roberta-large", vit_width=1024, vit_layers=2, text_hidden=1024, text_layers=1,
                text_heads=16, text_inter=4096)
    base.update(over)
    return compose("task_finetune_vqa_vqa_rad", "clip16", "text_roberta", **base)


def large1_batch():
    return synth.synthetic_batch(2, text_len=32, image_size=512, vocab_size=1000, rank=0)


def load_golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False)


def gen_t5_weights(sd):
    """Same recipe as oracle/make_golden.py::gen_t5_weights (kept in sync by the fixture test)."""
    from m3ae_amd import synth
    synth.fill_deterministic(sd)
    for k in sd:
        if any(t in k for t in (".q.weight", ".k.weight", ".v.weight", ".o.weight", ".wi.weight", ".wo.weight")):
            sd[k].mul_(8.0)
    sd["t5.shared.weight"].copy_(synth.det_normal("t5.shared.weight", sd["t5.shared.weight"].shape, std=0.3))
    return sd


def canon_generated(seq, eos):
    """Sequences up to and including the first EOS after the start token (what follows is padding, whose id differs
    between transformers releases)."""
    out = []
    for r in seq:
        r = [int(t) for t in r]
        if eos in r[1:]:
            r = r[: r.index(eos, 1) + 1]
        out.append(r)
    return out
