"""Attention kernel timing (HIP events) on the hot-path shapes; random bf16 data."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import ops, _lib  # noqa: E402

if os.environ.get("M3AE_LIB"):   # A/B against another build of the library
    _lib.LIB_PATH = os.environ["M3AE_LIB"]

B, H, D = int(os.environ.get("B", 64)), 12, 768


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


SHAPES = [(577, 577, False), (577, 577, False), (32, 577, False), (577, 32, True), (32, 32, True)]   # first line = warm-up
for (Lq, Lk, masked) in (SHAPES[:2] if os.environ.get("SELF_ONLY") else SHAPES):
    dev = "cuda"
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, device=dev)
        mask[:, Lk - 5:] = -10000.0
    if Lq == Lk:
        qkv = torch.randn(B, Lq, 3 * D, device=dev).to(torch.bfloat16)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    else:
        q = torch.randn(B, Lq, D, device=dev).to(torch.bfloat16)
        kv = torch.randn(B, Lk, 2 * D, device=dev).to(torch.bfloat16)
        k, v = kv[..., :D], kv[..., D:]
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        dk, dv = dkv[..., :D], dkv[..., D:]
    o, lse = ops.attn_forward(q, k, v, H, mask)
    do = torch.randn_like(o)
    fl = 4.0 * B * H * Lq * Lk * 64
    for coop, drop in ((1, None), (1, (0.1, 1234))):
        tf = time_it(lambda: ops.attn_forward(q, k, v, H, mask, dropout=drop))
        tb = time_it(lambda: ops.attn_backward(q, k, v, o, lse, do, dq, dk, dv, H, mask, dropout=drop))
        print(f"attn coop={coop} drop={0.0 if drop is None else drop[0]} Lq={Lq:4d} Lk={Lk:4d}: fwd {tf*1e3:8.1f} us {fl/tf/1e9:7.1f} TF/s | "
              f"bwd {tb*1e3:8.1f} us {2.5*fl/tb/1e9:7.1f} TF/s", flush=True)
