"""The clock the chip holds inside the dominant NT GEMM (MI355X_MICROARCH.md, 'DVFS give-back' item 6): a diagnostic build of
csrc/gemm_nt_pp2.hip (-DM3AE_EXP_PP2_CLOCK, lib_diag/) stamps s_memtime / s_memrealtime around each persistent workgroup's tile loop;
after >= 2 s of back-to-back launches on random data the median quotient x 100 MHz is the in-kernel shader clock.

    cd mm-vqa-healthcare_amd && M3AE_EXTRA_HIPCC_FLAGS=-DM3AE_EXP_PP2_CLOCK python -m m3ae_amd.build && cd .. &&
    M3AE_DIAGNOSTIC_LIB=1 python tools/nt_clock.py
"""
import ctypes
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import _lib, ops  # noqa: E402


def main():
    L = _lib.lib()
    fn = L.m3ae_diag_pp2_clock
    fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]
    m = 256 * 577
    for (n, k, fill) in ((2304, 768, "randn"), (768, 3072, "randn"), (3072, 768, "randn"), (2304, 768, "zeros")):
        x = (torch.randn(m, k, device="cuda") if fill == "randn" else torch.zeros(m, k, device="cuda")).to(torch.bfloat16)
        w = ((torch.randn(n, k, device="cuda") * k ** -0.5) if fill == "randn" else torch.zeros(n, k, device="cuda")).to(torch.bfloat16)
        y = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        ops.GEMM_NT_VARIANT = 10
        fn_run = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k)
        fn_run(); torch.cuda.synchronize()
        t0 = time.time()
        launches = 0
        while time.time() - t0 < 2.5:
            for _ in range(50):
                fn_run()
            torch.cuda.synchronize()
            launches += 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn_run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        buf = (ctypes.c_uint64 * 512)()
        assert fn(buf, 512) == 0
        mhz = [buf[2 * i] / buf[2 * i + 1] * 100.0 for i in range(256) if buf[2 * i + 1]]
        tf = 2.0 * m * n * k / us / 1e6
        med = statistics.median(mhz)
        print(f"[nt_clock] {m}x{n}x{k} {fill}: {us:7.1f} us {tf:6.0f} TF/s; in-kernel clock median {med:6.0f} MHz "
              f"(min {min(mhz):.0f}, max {max(mhz):.0f}, {len(mhz)} workgroups); bf16 MFMA peak at that clock "
              f"{2500.0 * med / 2400.0:6.0f} TF/s -> {tf / (2500.0 * med / 2400.0):.3f} of it", flush=True)
    ops.GEMM_NT_VARIANT = -1


if __name__ == "__main__":
    main()
