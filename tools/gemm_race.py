"""Race screen for the ping-pong NT kernel (variant 7): its accumulation order is the 2-stage kernel's (variant 4), so
the two must agree BIT FOR BIT on random data; repeated over shapes and iterations, with other work in flight."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import _lib, ops  # noqa: E402

ITERS = int(os.environ.get("ITERS", 30))
VAR = int(os.environ.get("VAR", 7))   # 7: ping-pong kernel, 8: its persistent form (shapes with >= 512 tiles)
SHAPES = [(147712, 768, 768), (73856, 768, 3072), (36928, 768, 768), (36928, 3072, 768), (36928, 768, 3072), (4096, 4096, 4096), (2308, 2304, 768), (1000, 520, 64),
          (777, 1288, 128), (8192, 8192, 1024), (5000, 768, 192)]


def main():
    L = _lib.lib()
    bad = 0
    for (m, n, k) in SHAPES:
        x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
        ref = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        ops.GEMM_NT_VARIANT = 4
        ops.gemm(x, k, 1, w, 1, k, ref, n, m, n, k)
        ops.GEMM_NT_VARIANT = VAR
        nbad = 0
        for it in range(ITERS):
            y = torch.full((m, n), float("nan"), device="cuda", dtype=torch.bfloat16)
            ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k)
            if not torch.equal(y.view(torch.int16), ref.view(torch.int16)):
                d = (y.float() - ref.float()).abs()
                nbad += 1
                if nbad <= 3:
                    idx = torch.nonzero(d > 0)
                    print(f"  MISMATCH {m}x{n}x{k} iter {it}: {idx.shape[0]} elements, max {d.max().item():.3e}, first {idx[0].tolist()}", flush=True)
        print(f"{m}x{n}x{k}: {ITERS - nbad}/{ITERS} bit-identical to variant 4 (variant {VAR})", flush=True)
        bad += nbad
    ops.GEMM_NT_VARIANT = -1
    print("RACE SCREEN", "FAILED" if bad else "clean")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
