"""Times the wgrad (TN) GEMM on the step's shapes at per-GPU batch B (default 256).   B=256 python tools/tn_ab.py [tag]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
B = int(os.environ.get("B", 256))
m = int(os.environ.get("M", B * 577))   # M=1024: the text side's reduction at per-GPU batch 32
ops.GEMM_TN_VARIANT = int(os.environ.get("TNVAR", -1))   # 5: the 256 x 256 ping-pong kernel where its preconditions hold; 2: 128 x 128


def time_it(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


out = []
for (n, k) in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
    dy = torch.randn(m, n, device="cuda").to(torch.bfloat16)
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    g = torch.zeros(n, k, device="cuda")
    gb = torch.zeros(n, device="cuda")
    fn = lambda: ops.gemm(dy, 1, n, x, k, 1, g, k, n, k, m, accumulate=True, a_rowsum=gb)
    ms = min(time_it(fn) for _ in range(3))
    out.append(f"{n}x{k}: {ms * 1e3:7.1f} us {2.0 * m * n * k / ms / 1e9:6.0f} TF/s")
print(f"[{tag}] wgrad, reduction {m}: " + " | ".join(out), flush=True)
