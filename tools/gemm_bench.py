"""Per-shape timing of the MFMA GEMM kernels (HIP events, interleaved variants in ONE process; random bf16 data)."""
import os
import sys
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import _lib, ops  # noqa: E402

B = int(os.environ.get("B", 64))
VARS = tuple(int(v) for v in os.environ.get('VARS', '0,4,7').split(','))
M = B * 577
NT_SHAPES = [(M, 3072, 768)] if os.environ.get('TN_ONLY') else [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (B * 32, 768, 768), (B * 32, 3072, 768), (B * 32, 768, 3072), (B * 32, 2304, 768), (B * 32, 1536, 768),
             (4096, 4096, 4096), (8192, 8192, 8192)]
TN_SHAPES = [(M, 768, 768), (M, 2304, 768), (M, 3072, 768), (M, 768, 3072), (B * 32, 768, 768)]


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
    L = _lib.lib()
    dev = "cuda"
    print(f"B={B}")
    for (m, n, k) in NT_SHAPES:
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        res = []
        for rnd in range(2):
            for v in VARS:
                ops.GEMM_NT_VARIANT = v
                ms = time_it(lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k))
                res.append((v, ms))
        best = {v: min(ms for vv, ms in res if vv == v) for v in VARS}
        print(f"NT {m:6d}x{n:5d}x{k:5d}: " + "  ".join(f"{('auto' if v < 0 else 'v' + str(v))}: {best[v]*1e3:8.1f} us {2.0*m*n*k/best[v]/1e9:7.1f} TF/s" for v in VARS), flush=True)
    # epilogue-heavy forms on the dominant shapes
    m = M
    for vv in tuple(int(v) for v in os.environ.get('EVARS', '0,4,7').split(',')):
      ops.GEMM_NT_VARIANT = vv
      print("variant", vv)
      for (n, k, kind) in [(3072, 768, "gelu+preact"), (3072, 768, "dgelu"), (768, 768, "bias+res"), (768, 3072, "bias+res")]:
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        b = torch.randn(n, device=dev)
        aux = torch.randn(m, n, device=dev).to(torch.bfloat16)
        pre = torch.empty_like(y)
        if kind == "gelu+preact":
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k, bias=b, act=ops.ACT_GELU, preact=pre)
        elif kind == "dgelu":
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k, dact_aux=aux, dact=ops.ACT_GELU)
        else:
            fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k, bias=b, residual=aux)
        ms = time_it(fn)
        print(f"NT {m:6d}x{n:5d}x{k:5d} {kind:12s}: {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF/s", flush=True)
    ops.GEMM_NT_VARIANT = 0
    for (m, n, k) in TN_SHAPES:
        dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        g = torch.zeros(n, k, device=dev)
        out = []
        for tv in (2, 5):
            ops.GEMM_TN_VARIANT = tv
            ms = time_it(lambda: ops.gemm(dy, 1, n, x, k, 1, g, k, n, k, m, accumulate=True))
            out.append(f"tv{tv}: {ms*1e3:8.1f} us {2.0*m*n*k/ms/1e9:7.1f} TF/s")
        print(f"TN red={m:6d} out {n:5d}x{k:5d}: " + "  ".join(out), flush=True)
    ops.GEMM_TN_VARIANT = -1


if __name__ == "__main__":
    main()
