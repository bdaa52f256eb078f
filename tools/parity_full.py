"""Prints the observed full-size bf16 (and fp32) errors against tests/golden/full_vqa.npz:  python tools/parity_full.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mm-vqa-healthcare_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from m3ae_amd import ops, synth  # noqa: E402
from m3ae_amd.modules import M3AETransformerSS  # noqa: E402
from m3ae_amd.parity import parity_report  # noqa: E402
from oracle_util import finetune_vqa_rad_config, full_batch, load_golden  # noqa: E402


def to_dev(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if isinstance(v, list) and v and isinstance(v[0], torch.Tensor) else v))
            for k, v in batch.items()}


for mode, dtype in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
    for rule in ((0, 96) if mode == "bf16" else (96,)):
        ops.XATTN_TRAIN_MIN_BATCH = rule
        m = M3AETransformerSS(finetune_vqa_rad_config(compute_dtype=mode))
        synth.fill_deterministic(m)
        m.finalize("cuda", dtype)
        m.eval()
        rep = parity_report(m, load_golden("full_vqa.npz"), to_dev(full_batch()))
        print(mode, "xattn_train_min_batch", rule, json.dumps(rep), flush=True)
        del m
        torch.cuda.empty_cache()
