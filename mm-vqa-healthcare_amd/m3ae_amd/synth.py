"""Deterministic synthetic weights and batches (SURVEY.md 8c "build-owned deterministic generator", 8d
"Synthetic inputs").

There is no network, no checkpoint and no dataset on either box, so weights and inputs come from a
counter-based generator (numpy Philox keyed by a hash of the tensor NAME).  The same call produces
the same bits in the build container (where it fills the reference's modules for the golden
fixtures) and on the GPU box (where it fills this package's modules), independent of construction
order and of torch's RNG.
"""
import hashlib

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)  # reference m3ae/transforms/transform.py:66
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _philox(name: str, salt: int = 0) -> np.random.Generator:
    h = hashlib.sha256(f"{salt}:{name}".encode()).digest()
    key = int.from_bytes(h[:16], "little")
    return np.random.Generator(np.random.Philox(key=key))


def det_normal(name: str, shape, std: float = 1.0, mean: float = 0.0, salt: int = 0) -> torch.Tensor:
    g = _philox(name, salt)
    a = g.standard_normal(size=tuple(shape), dtype=np.float32)
    if std != 1.0:
        a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return torch.from_numpy(a)


def det_uniform(name: str, shape, salt: int = 0) -> torch.Tensor:
    g = _philox(name, salt)
    return torch.from_numpy(g.random(size=tuple(shape), dtype=np.float32))


def det_randint(name: str, lo: int, hi: int, shape, salt: int = 0) -> torch.Tensor:
    """Integers in [lo, hi)."""
    g = _philox(name, salt)
    return torch.from_numpy(g.integers(lo, hi, size=tuple(shape), dtype=np.int64))


def _std_for(name: str, shape) -> tuple:
    """(mean, std) per tensor name: reference inits (m3ae_utils.py:101-109 N(0,0.02);
    clip_model.py:86-88,169-180) with non-trivial LayerNorm scales/biases so parity tests exercise them."""
    last = name.rsplit(".", 1)[-1]
    if len(shape) == 1 and last == "weight":  # LayerNorm / norm scales
        return 1.0, 0.1
    if last == "bias" or name.endswith("in_proj_bias") or name == "mlm_head.bias":
        return 0.0, 0.02
    if name.endswith("class_embedding") or name.endswith("visual.positional_embedding"):
        return 0.0, float(shape[-1]) ** -0.5
    if name.endswith("in_proj_weight"):
        return 0.0, float(shape[-1]) ** -0.5
    if name.endswith("attn.out_proj.weight") or name.endswith("mlp.c_proj.weight"):
        return 0.0, float(shape[0]) ** -0.5 * (2 * 12) ** -0.5
    if name.endswith("mlp.c_fc.weight"):
        return 0.0, (2.0 * float(shape[-1])) ** -0.5
    return 0.0, 0.02


@torch.no_grad()
def fill_deterministic(module_or_sd, salt: int = 0, skip=()):
    """Overwrite every parameter (and floating buffer that is part of the state_dict) by name."""
    if isinstance(module_or_sd, dict):
        items = list(module_or_sd.items())
    else:
        items = list(module_or_sd.state_dict().items())
    for name, t in items:
        if not torch.is_floating_point(t) or any(s in name for s in skip):
            continue
        if name.endswith("decoder_pos_embed"):  # fixed sin-cos table (prediction_heads.py:52-55)
            continue
        if name.endswith("positional_encoding.pe"):  # fixed sinusoid buffer of the decoder head (m3ae_decoder.py:24-33)
            continue
        mean, std = _std_for(name, t.shape)
        t.copy_(det_normal(name, t.shape, std=std, mean=mean, salt=salt).to(t.dtype))
    return module_or_sd


def synthetic_batch(batch_size: int, text_len: int = 32, image_size: int = 384, vocab_size: int = 50265,
                    label_size: int = 498, rank: int = 0, device="cpu", pretrain: bool = False):
    """CLIP-normalised noise images, RoBERTa-style ids with random lengths and pad id 1,
    one answer label per sample.  Schema = reference base_dataset.py:165-228 collate output."""
    B, S, R = batch_size, text_len, image_size
    u = det_uniform("image", (B, 3, R, R), salt=1234 + rank)
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    image = (u - mean) / std
    lens = det_randint("lens", min(6, S), S + 1, (B,), salt=4321 + rank)
    ids = det_randint("ids", 3, vocab_size - 1, (B, S), salt=4321 + rank)
    pos = torch.arange(S).view(1, S)
    ids = torch.where(pos == 0, torch.zeros_like(ids), ids)
    ids = torch.where(pos == (lens.view(B, 1) - 1), torch.full_like(ids, 2), ids)
    ids = torch.where(pos >= lens.view(B, 1), torch.ones_like(ids), ids)
    masks = (ids != 1).long()
    labels = det_randint("vqa", 0, label_size, (B,), salt=777 + rank)
    types = det_randint("atype", 0, 2, (B,), salt=778 + rank)
    batch = {
        "image": [image.to(device)],
        "text_ids": ids.to(device),
        "text_labels": torch.full((B, S), -100, dtype=torch.long, device=device),
        "text_masks": masks.to(device),
        "vqa_labels": [[int(l)] for l in labels],
        "vqa_scores": [[1.0] for _ in range(B)],
        "vqa_answer": [["a"] for _ in range(B)],
        "answer_types": [int(t) for t in types],
        "text": ["q"] * B,
    }
    if pretrain:
        u2 = det_uniform("false_image", (B, 3, R, R), salt=2234 + rank)
        batch["false_image_0"] = [((u2 - mean) / std).to(device)]
        special = (ids <= 2)
        pick = det_uniform("mlm_pick", (B, S), salt=55 + rank) < 0.15
        pick = pick & ~special
        ids_mlm = torch.where(pick, torch.full_like(ids, vocab_size - 1), ids)
        labels_mlm = torch.where(pick, ids, torch.full_like(ids, -100))
        batch["text_ids_mlm"] = ids_mlm.to(device)
        batch["text_labels_mlm"] = labels_mlm.to(device)
        batch["mim_noise"] = det_uniform("mim_noise", (B, (R // 16) ** 2), salt=99 + rank).to(device)
    return batch
