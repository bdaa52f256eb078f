"""NT GEMM (K = 768) with the activation operand A hot in L2 / Infinity Cache (same buffer every launch) against cold
(rotating over buffers that together exceed the 256-MB Infinity Cache): how much of the kernel's time is A's fetch latency."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch
from m3ae_amd import ops
dev = "cuda"
K = 768
for M, N in [(16384, 3072), (16384, 768), (65536, 3072), (65536, 768), (147712, 3072), (147712, 768)]:
    nbuf = max(2, int(600e6 / (M * K * 2)) + 1)
    xs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(nbuf)]
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for mode in ("hot", "cold", "hot", "cold"):
        for i in range(nbuf):
            ops.gemm(xs[i % nbuf if mode == "cold" else 0], K, 1, w, 1, K, y, N, M, N, K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(20, 2 * nbuf)
        e0.record()
        for i in range(n):
            ops.gemm(xs[i % nbuf if mode == "cold" else 0], K, 1, w, 1, K, y, N, M, N, K)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1) / n * 1e3)
    fl = 2.0 * M * N * K
    print(f"NT {M}x{N}x{K} ({M * K * 2 / 1e6:.0f} MB of A, {nbuf} buffers): hot {min(res['hot']):7.1f} us {fl / min(res['hot']) / 1e6:7.1f} TF/s | "
          f"cold {min(res['cold']):7.1f} us {fl / min(res['cold']) / 1e6:7.1f} TF/s  path {ops.last_gemm_path()}", flush=True)
    del xs
