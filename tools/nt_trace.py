"""Where a chunk of the persistent ping-pong NT GEMM spends its cycles (diagnostic build only):
    touch mm-vqa-healthcare_amd/csrc/gemm_mfma.hip; M3AE_EXTRA_HIPCC_FLAGS=-DM3AE_NT_TRACE python -m m3ae_amd.build
    python tools/nt_trace.py [tag]
Per workgroup and wave row the kernel sums the shader clocks of every section of its main loop (a chunk = two phases);
the table is the median over workgroups of (sum / chunks).  The stamps cost cycles themselves: read the SHARES."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from m3ae_amd import _lib, ops  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    L = _lib.lib()
    L.m3ae_nt_trace_dump.argtypes, L.m3ae_nt_trace_dump.restype = [C.c_void_p], C.c_int
    m = 256 * 577
    for (n, k) in [(2304, 768), (768, 3072)]:
        x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
        y = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k)
        e1.record()
        torch.cuda.synchronize()
        buf = np.zeros((1024, 2, 8), dtype=np.uint64)
        assert L.m3ae_nt_trace_dump(buf.ctypes.data) == 0
        t = buf[:256].astype(np.float64)
        names = ["fragment reads", "DMA issue (2 pieces)", "barrier 1", "lgkmcnt wait", "MFMA issue (16)", "barrier 2", "vmcnt wait"]
        print(f"[{tag}] {m}x{n}x{k}: {e0.elapsed_time(e1) * 1e3:.0f} us (traced build); clocks per CHUNK (two phases), median over 256 workgroups")
        for wr in (0, 1):
            per = t[:, wr, :7] / t[:, wr, 7:8]
            med = np.median(per, axis=0)
            print(f"   wave row {wr}: " + "  ".join(f"{nm} {v:5.0f}" for nm, v in zip(names, med)) + f"   total {med.sum():5.0f}")


if __name__ == "__main__":
    main()
