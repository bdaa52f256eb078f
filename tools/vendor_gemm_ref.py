"""Calibration only (NOT on the product path, which never calls a vendor BLAS): the path's large GEMM shapes on this library's
kernels next to torch.matmul (hipBLASLt / rocBLAS) on the same box, same timing loop, plain epilogue (no bias / activation /
residual -- the vendor GEMM has none of the fused epilogues the path uses).  bf16 operands; the wgrad (TN) product accumulates
into fp32 here and returns bf16 there."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402

dev = "cuda"
M = 147712


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def row(name, fl, us_mine, us_vendor):
    print(f"{name:44s} this library {us_mine:8.1f} us {fl / us_mine / 1e6:7.1f} TF/s | vendor {us_vendor:8.1f} us {fl / us_vendor / 1e6:7.1f} TF/s | "
          f"ratio {us_vendor / us_mine:5.2f}", flush=True)


for n, k in [(3072, 768), (768, 3072), (2304, 768), (768, 768)]:
    x = torch.randn(M, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    row(f"NT  Y = X W^T   {M} x {n} x {k}", 2.0 * M * n * k,
        t(lambda: ops.gemm(x, k, 1, w, 1, k, y, n, M, n, k)), t(lambda: torch.matmul(x, w.t())))
for n, k in [(3072, 768), (768, 3072), (2304, 768), (768, 768)]:
    dy = torch.randn(M, n, device=dev).to(torch.bfloat16)
    x = torch.randn(M, k, device=dev).to(torch.bfloat16)
    g = torch.zeros(n, k, device=dev)
    row(f"TN  dW += dY^T X   {n} x {k}, reduction {M}", 2.0 * M * n * k,
        t(lambda: ops.gemm(dy, 1, n, x, k, 1, g, k, n, k, M, accumulate=True)), t(lambda: torch.matmul(dy.t(), x)))
a = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
b = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
c = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
row("NT  8192^3", 2.0 * 8192 ** 3, t(lambda: ops.gemm(a, 8192, 1, b, 1, 8192, c, 8192, 8192, 8192, 8192)), t(lambda: torch.matmul(a, b.t())))
