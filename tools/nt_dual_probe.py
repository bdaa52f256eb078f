"""Persistent ping-pong NT kernel (variant 8) against the dual kernel (variant 9: 128 x 256 tiles, two workgroups per CU) on the
path's large shapes and epilogue classes, with a sweep of the dual kernel's start-up stagger (m3ae_set_tuning key 4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import _lib, ops  # noqa: E402
from gemm_bench import time_it  # noqa: E402

B = int(os.environ.get("B", 256))
M = B * 577


def main():
    L = _lib.lib()
    dev = "cuda"
    cases = [(3072, 768, "gelu+preact"), (3072, 768, "dgelu"), (2304, 768, "plain"), (768, 768, "bias+res"),
             (768, 3072, "bias+res"), (1536, 768, "plain")]
    sweeps = [(int(v), -1) for v in os.environ.get('VARS', '8,9,8').split(',')] if not os.environ.get('GC') else [(8, int(g)) for g in os.environ['GC'].split(',')]
    for (n, k, kind) in cases:
        x = torch.randn(M, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
        b = torch.randn(n, device=dev)
        aux = torch.randn(M, n, device=dev).to(torch.bfloat16)
        pre = torch.empty_like(y)
        if kind == "gelu+preact":
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, M, n, k, bias=b, act=ops.ACT_GELU, preact=pre)
        elif kind == "dgelu":
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, M, n, k, dact_aux=aux, dact=ops.ACT_GELU)
        elif kind == "plain":
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, M, n, k, bias=b)
        else:
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, M, n, k, bias=b, residual=aux)
        out = []
        for d, tk in sweeps:
            L.m3ae_set_tuning(0, d)
            L.m3ae_set_tuning(7 if os.environ.get('GC') else 4, tk)
            ms = time_it(fn)
            out.append(f"variant {d} {'col-group' if os.environ.get('GC') else 'stagger'} {tk}: {ms * 1e3:7.1f}us {2.0 * M * n * k / ms / 1e9:6.0f}TF")
        print(f"NT {M}x{n}x{k} {kind:12s}\n   " + "\n   ".join(out), flush=True)
    L.m3ae_set_tuning(0, -1)
    L.m3ae_set_tuning(4, -1)
    L.m3ae_set_tuning(7, 0)


if __name__ == "__main__":
    main()
