"""How host-bound is the training step at a small per-GPU batch?  Per step: the CPU time to ENQUEUE the step (no synchronisation
inside the loop) against the GPU time between the step's first and last kernel (HIP events), and the kernels' own time from the
torch profiler (sum of device durations = what a hipGraph replay of the same launches could approach).
    B=32 python tools/host_bound.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mm-vqa-healthcare_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from m3ae_amd import synth  # noqa: E402
from m3ae_amd.config import finetune_vqa_rad_config  # noqa: E402
from m3ae_amd.modules import M3AETransformerSS  # noqa: E402
from m3ae_amd.modules.objectives import build_vqa_targets  # noqa: E402
from bench import to_dev  # noqa: E402


def main():
    B = int(os.environ.get("B", 32))
    dev = torch.device("cuda", 0)
    cfg = finetune_vqa_rad_config(compute_dtype="bf16")
    model = M3AETransformerSS(cfg)
    synth.fill_deterministic(model)
    model.finalize(dev, torch.bfloat16)
    model.train(True)
    store = model.store
    batch = to_dev(synth.synthetic_batch(B, text_len=32, image_size=384, rank=0), dev)
    batch["vqa_targets"] = build_vqa_targets(batch, cfg["vqa_label_size"], dev)

    def step():
        store.zero_grad()
        loss = model.training_step(batch)
        loss.backward()
        store.adamw_step(max_steps=100)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1) / n
    import torch.profiler as tp
    with tp.profile(activities=[tp.ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    kern = [e for e in prof.key_averages() if e.device_time_total > 0]
    ksum = sum(e.device_time_total for e in kern) / 3 / 1e3
    nlaunch = sum(e.count for e in kern) / 3
    print(f"B={B}: step {t_all / n * 1e3:.2f} ms wall, {gpu_ms:.2f} ms by events; CPU enqueue {t_enq / n * 1e3:.2f} ms per step; "
          f"sum of kernel durations {ksum:.2f} ms over {nlaunch:.0f} launches per step "
          f"-> gaps {gpu_ms - ksum:.2f} ms ({(gpu_ms - ksum) / gpu_ms * 100:.0f} % of the step)")
    top = sorted(kern, key=lambda e: -e.device_time_total)[:12]
    for e in top:
        print(f"   {e.device_time_total / 3 / 1e3:7.2f} ms  x{e.count // 3:4d}  {e.key[:100]}")


if __name__ == "__main__":
    main()
