"""Minimal launches for PMC collection: one NT and one TN GEMM on the dominant shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch
from m3ae_amd import _lib, ops
L = _lib.lib()
m, n, k = 36928, 3072, 768
dev = "cuda"
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
g = torch.zeros(n, k, device=dev)
for v in (0, 4):
    ops.GEMM_NT_VARIANT = v
    for _ in range(3):
        ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k)
for _ in range(3):
    ops.gemm(dy, 1, n, x, k, 1, g, k, n, k, m, accumulate=True)
torch.cuda.synchronize()
