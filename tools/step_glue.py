"""Which ATen ops (copies, fills, adds) still run inside the timed step, with shapes: torch.profiler over one training step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
import torch.profiler as tp  # noqa: E402
from m3ae_amd import synth  # noqa: E402
from m3ae_amd.config import finetune_vqa_rad_config  # noqa: E402
from m3ae_amd.modules import M3AETransformerSS  # noqa: E402
from m3ae_amd.modules.objectives import build_vqa_targets  # noqa: E402

B = int(os.environ.get("B", 256))
cfg = finetune_vqa_rad_config(compute_dtype="bf16")
m = M3AETransformerSS(cfg)
synth.fill_deterministic(m)
m.finalize("cuda", torch.bfloat16)
m.train()
b = synth.synthetic_batch(B, text_len=32, image_size=384, rank=0)
b = {k: (v.cuda() if isinstance(v, torch.Tensor) else ([t.cuda() for t in v] if isinstance(v, list) and v and isinstance(v[0], torch.Tensor) else v)) for k, v in b.items()}
b["vqa_targets"] = build_vqa_targets(b, cfg["vqa_label_size"], "cuda")


def step():
    m.store.zero_grad()
    m.training_step(b).backward()
    m.store.adamw_step(max_steps=100)


for _ in range(2):
    step()
torch.cuda.synchronize()
with tp.profile(activities=[tp.ProfilerActivity.CPU, tp.ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [(e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:110]) for e in ka if e.self_device_time_total > 0]
for t, c, n, sh in sorted(rows, reverse=True)[:45]:
    print(f"{t / 1e3:8.2f} ms  x{c:4d}  {n[:60]:60s} {sh}")
