"""GraphedStep (m3ae_amd/graph.py) against the eager step: per-GPU batch B (default 32), same weights, same batch.
 1. eval-mode (no dropout): three graphed steps == three eager steps (parameters bit for bit up to the order of fp32 atomics);
 2. train mode: the loss of consecutive replays differs (fresh dropout masks per replay) and stays close to the eager losses;
 3. timing: eager step vs graph replay (wall + HIP events), CPU enqueue share.
    B=32 python tools/graph_check.py [--head t5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mm-vqa-healthcare_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from m3ae_amd import ops, synth  # noqa: E402
from m3ae_amd.config import finetune_vqa_rad_config  # noqa: E402
from m3ae_amd.graph import GraphedStep  # noqa: E402
from m3ae_amd.modules import M3AETransformerSS  # noqa: E402
from m3ae_amd.modules.objectives import build_vqa_targets  # noqa: E402

B = int(os.environ.get("B", 32))
HEAD = "t5" if "--head" in sys.argv and sys.argv[sys.argv.index("--head") + 1] == "t5" else "cls"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def to_dev(batch):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else ([t.to(dev) for t in v] if isinstance(v, list) and v and isinstance(v[0], torch.Tensor) else v))
            for k, v in batch.items()}


def make(train):
    cfg = finetune_vqa_rad_config(compute_dtype="bf16", t5_model_name="t5-base")
    if HEAD == "t5":
        from m3ae_amd.modules import T5VQA_MMEncoderInput
        m = T5VQA_MMEncoderInput(cfg)
        m.unfreeze_top_layers(4, 4)
    else:
        m = M3AETransformerSS(cfg)
    synth.fill_deterministic(m)
    m.finalize(dev, torch.bfloat16)
    m.train(train)
    batch = to_dev(synth.synthetic_batch(B, text_len=32, image_size=384, rank=0))
    batch["vqa_targets"] = build_vqa_targets(batch, cfg["vqa_label_size"], dev)
    if HEAD == "t5":
        lab = synth.det_randint("t5_labels", 2, 32128, (B, 6), salt=31)
        lab[:, -1] = 1
        batch["t5_labels"] = lab.to(dev)
    return m, batch


def eager_step(m, batch, max_steps=100):
    m.store.zero_grad()
    out = m.training_step(batch)
    loss = out["loss"] if isinstance(out, dict) else out
    loss.backward()
    m.store.adamw_step(max_steps=max_steps)
    return loss.detach()


# ---- 1. eval mode: graphed == eager
ops.set_dropout_seed(7)
m1, b1 = make(False)
l_e = [eager_step(m1, b1).item() for _ in range(4)]
flat_e = m1.store.flat.clone()
del m1
torch.cuda.empty_cache()
m2, b2 = make(False)
l_g = [eager_step(m2, b2).item()]
gs = GraphedStep(m2, b2, max_steps=100)
for _ in range(3):
    l_g.append(gs.step().item())
torch.cuda.synchronize()
d = (m2.store.flat - flat_e).abs().max().item()
moved = flat_e.abs().max().item()
print(f"[eval] eager losses {l_e}\n[eval] graph losses {l_g}\n[eval] max |param difference| after 4 steps {d:.3e} (max |param| {moved:.3e}), "
      f"step_count {m2.store.step_count}", flush=True)
assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(l_e, l_g)), "graphed losses differ from eager"
assert d < 5e-3, d
del m2, gs
torch.cuda.empty_cache()

# ---- 2. train mode: fresh masks per replay
ops.set_dropout_seed(11)
m3, b3 = make(True)
eager_step(m3, b3)
gs = GraphedStep(m3, b3, max_steps=10 ** 6)     # ~constant (tiny) learning rate: loss differences come from the masks
losses = [gs.step().item() for _ in range(5)]
print(f"[train] losses of 5 graphed steps {losses}", flush=True)
assert len({round(x, 4) for x in losses}) >= 4, "replays repeat the same dropout masks"

# ---- 3. timing
def timed(fn, n):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    t_enq = time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, e0.elapsed_time(e1) / n, t_enq / n * 1e3


for _ in range(3):
    gs.step()
wall_g, ev_g, enq_g = timed(gs.step, 20)
m4, b4 = make(True)
for _ in range(3):
    eager_step(m4, b4)
wall_e, ev_e, enq_e = timed(lambda: eager_step(m4, b4), 20)
print(f"[time] B={B} head={HEAD}: eager {wall_e:.2f} ms/step wall ({ev_e:.2f} by events, host enqueue {enq_e:.2f} ms = {100 * enq_e / wall_e:.0f} %) "
      f"-> {B / wall_e * 1e3:.0f} pairs/s | graph replay {wall_g:.2f} ms/step wall ({ev_g:.2f} by events, host {enq_g:.2f} ms = {100 * enq_g / wall_g:.0f} %) "
      f"-> {B / wall_g * 1e3:.0f} pairs/s", flush=True)
