"""cProfile of the host side of the training step at a small per-GPU batch (where the step is bound by the CPU's enqueue rate):
   B=32 python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from m3ae_amd import ops, synth  # noqa: E402
from m3ae_amd.config import finetune_vqa_rad_config  # noqa: E402
from m3ae_amd.modules import M3AETransformerSS  # noqa: E402

B = int(os.environ.get("B", 32))
cfg = finetune_vqa_rad_config(compute_dtype="bf16")
m = M3AETransformerSS(cfg)
synth.fill_deterministic(m)
m.finalize("cuda", torch.bfloat16)
m.train()
m.set_task()
b = synth.synthetic_batch(B, text_len=32, image_size=384, rank=0)
b = {k: (v.cuda() if isinstance(v, torch.Tensor) else [t.cuda() for t in v] if isinstance(v, list) and v and isinstance(v[0], torch.Tensor) else v)
     for k, v in b.items()}


def step():
    m.store.zero_grad()
    loss = m.training_step(b)
    loss = loss["loss"] if isinstance(loss, dict) else loss
    loss.backward()
    m.store.adamw_step(max_steps=1000)


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
