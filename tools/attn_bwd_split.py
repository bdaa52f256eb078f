"""Per-kernel times of the self-attention backward (dQ kernel, dK/dV kernel) and forward at the step's image shape, torch profiler.
    B=256 python tools/attn_bwd_split.py [tag]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
import torch.profiler as tp  # noqa: E402
from m3ae_amd import ops  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else ""
B, H, D, L = int(os.environ.get("B", 256)), 12, 768, int(os.environ.get("L", 577))
qkv = torch.randn(B, L, 3 * D, device="cuda").to(torch.bfloat16)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
dqkv = torch.empty_like(qkv)
dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
fl = 4.0 * B * H * L * L * 64
for drop in (None, (0.1, 1234)):
    o, lse = ops.attn_forward(q, k, v, H, None, dropout=drop)
    do = torch.randn_like(o)
    def run():
        ops.attn_forward(q, k, v, H, None, dropout=drop)
        ops.attn_backward(q, k, v, o, lse, do, dq, dk, dv, H, None, dropout=drop)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    with tp.profile(activities=[tp.ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            run()
        torch.cuda.synchronize()
    rows = {e.key.split("<")[0].split("::")[-1]: e.device_time_total / e.count for e in prof.key_averages() if e.device_time_total > 0}
    f, a, b_ = rows.get("attn_fwd_coop_kernel", 0), rows.get("attn_bwd_dq_coop_kernel", 0), rows.get("attn_bwd_dkdv_coop_kernel", 0)
    print(f"[{tag}] B={B} L={L} drop={0 if drop is None else drop[0]}: fwd {f:7.1f} us ({fl / f / 1e6:5.0f} TF/s)  dQ {a:7.1f} us  dK/dV {b_:7.1f} us  "
          f"bwd {a + b_:7.1f} us ({2.5 * fl / (a + b_) / 1e6:5.0f} TF/s)", flush=True)
