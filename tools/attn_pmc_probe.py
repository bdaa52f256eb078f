"""A few launches of the image self-attention forward / backward (577 x 577, B = 256, 12 heads) for counter collection:
rocprofv3 --pmc ... -- python tools/attn_pmc_probe.py;  with a counter_collection.csv as argument: per-kernel averages."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

if len(sys.argv) > 1:
    import gemm_pmc_probe
    gemm_pmc_probe.summarize(sys.argv[1])
    sys.exit(0)
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402

B, H, D, L = int(os.environ.get("B", 256)), 12, 768, 577
qkv = torch.randn(B, L, 3 * D, device="cuda").to(torch.bfloat16)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
o, lse = ops.attn_forward(q, k, v, H, None)
do = torch.randn_like(o)
for _ in range(4):
    ops.attn_forward(q, k, v, H, None)
    ops.attn_backward(q, k, v, o, lse, do, dq, dk, dv, H, None)
torch.cuda.synchronize()
