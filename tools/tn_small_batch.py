"""wgrad (TN) kernel choice at small per-GPU batches: the 128x128 kernel (variant 2) against the 256x256 ping-pong kernel
(variant 5) per shape, reduction length = B * 577 image tokens or B * 32 text tokens.   python tools/tn_small_batch.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402


def time_it(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = "cuda"
    for B in (32, 64, 128):
        for rows in (B * 577, B * 32):
            for (n, k) in [(768, 768), (2304, 768), (3072, 768), (768, 3072), (1536, 768)]:
                dy = torch.randn(rows, n, device=dev).to(torch.bfloat16)
                x = torch.randn(rows, k, device=dev).to(torch.bfloat16)
                g = torch.zeros(n, k, device=dev)
                db = torch.zeros(n, device=dev)
                out = []
                for tv in (-1, 2, 5):
                    ops.GEMM_TN_VARIANT = tv
                    ms = min(time_it(lambda: ops.gemm(dy, 1, n, x, k, 1, g, k, n, k, rows, accumulate=True, a_rowsum=db)) for _ in range(2))
                    out.append(f"{'auto' if tv < 0 else 'v' + str(tv)} {ms * 1e3:7.1f} us {2.0 * rows * n * k / ms / 1e9:6.0f} TF/s")
                ops.GEMM_TN_VARIANT = -1
                print(f"B={B:3d} rows {rows:6d} out {n:4d}x{k:4d}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
