"""Times the dominant NT GEMM (auto = persistent ping-pong kernel) on the step's shapes at per-GPU batch B (default 256):
plain, bias + residual, GELU + saved derivative, derivative multiply.   B=256 python tools/nt_exp.py [tag]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402


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


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    B = int(os.environ.get("B", 256))
    m, dev = B * 577, "cuda"
    out = []
    for (n, k, kind) in [(2304, 768, "plain"), (768, 768, "bias+res"), (3072, 768, "gelu+deriv"), (768, 3072, "bias+res"),
                         (3072, 768, "dmul"), (768, 3072, "plain")]:
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        b = torch.randn(n, device=dev)
        if kind == "plain":
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k)
        elif kind == "bias+res":
            aux = torch.randn(m, n, device=dev).to(torch.bfloat16)
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k, bias=b, residual=aux)
        elif kind == "gelu+deriv":
            pre = torch.empty_like(y)
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k, bias=b, act=ops.ACT_GELU, preact=pre, preact_grad=True)
        else:
            aux = torch.randn(m, n, device=dev).to(torch.bfloat16)
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k, dact_aux=aux, dact=ops.ACT_MULAUX)
        ms = min(time_it(fn), time_it(fn))
        out.append(f"{n}x{k} {kind}: {ms * 1e3:7.1f} us {2.0 * m * n * k / ms / 1e9:6.0f} TF/s")
        del x, w, y
    print(f"[{tag}] M={m} " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
