"""Runs N forward-only layer pairs of the fused cross-attention sub-block (text <- image, image <- text) at the bench batch -- the
workload of bench.py's cross_attention_fwd leg -- for profiling (rocprofv3 --kernel-trace / --pmc FETCH_SIZE / WRITE_SIZE):
    N=4 B=256 python tools/xattn_pair.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402
import xattn_bench as xb  # noqa: E402


def main():
    B, N, I, T, D = int(os.environ.get("B", 256)), int(os.environ.get("N", 4)), 577, 32, 768
    torch.manual_seed(0)
    att, store = xb.make(scale=2.0)
    P = att.block_params()
    xt = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
    xi = torch.randn(B * I, D, device="cuda").to(torch.bfloat16)
    mt = torch.zeros(B, T, device="cuda")
    mt[:, T - 9:] = -10000.0
    torch.cuda.synchronize()
    for _ in range(N):
        ops.xattn_fwd(xt, B, T, xi, I, None, P, 0.0, need_bwd=False)
        ops.xattn_fwd(xi, B, I, xt, T, mt, P, 0.0, need_bwd=False)
    torch.cuda.synchronize()
    print(f"ran {N} layer pairs at B={B}")


if __name__ == "__main__":
    main()
