"""In-loop timeline of the TN (wgrad) ping-pong kernel (DIAGNOSTIC build):
    touch mm-vqa-healthcare_amd/csrc/gemm_mfma.hip; M3AE_EXTRA_HIPCC_FLAGS=-DM3AE_TN_TRACE python -m m3ae_amd.build
    python tools/tn_trace.py
Stamps (shader clocks) of reduction chunk 64 for one wave of each wave row (wave column 0) of every workgroup."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import _lib, ops  # noqa: E402

L = C.CDLL(_lib.LIB_PATH)
M = 147712
buf = np.zeros((512, 2, 16), dtype=np.uint64)
for n, k in [(3072, 768), (768, 768)]:
    dy = torch.randn(M, n, device="cuda").to(torch.bfloat16)
    x = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    g = torch.zeros(n, k, device="cuda")
    for _ in range(3):
        ops.gemm(dy, 1, n, x, k, 1, g, k, n, k, M, accumulate=True)
    torch.cuda.synchronize()
    L.m3ae_tn_trace_dump(buf.ctypes.data_as(C.c_void_p))
    ops.gemm(dy, 1, n, x, k, 1, g, k, n, k, M, accumulate=True)
    torch.cuda.synchronize()
    L.m3ae_tn_trace_dump(buf.ctypes.data_as(C.c_void_p))
    t = buf.astype(np.int64)
    names = ["frag reads + B DMA issue", "barrier", "lgkm wait + 16 MFMAs", "barrier", "A frag reads + A DMA issue + vmcnt wait", "barrier",
             "lgkm wait + 16 MFMAs + barrier + loop"]
    idx = [0, 1, 2, 3, 4, 5, 6, 8]
    print(f"wgrad {n}x{k}, reduction {M}: path {ops.last_gemm_path()}")
    for wr in (0, 1):
        q = t[:, wr]
        q = q[(q[:, 8] > 0) & (q[:, 0] > 0)]
        print(f"  wave row {wr}: {len(q)} workgroups; chunk 64, median shader clocks")
        for i, nm in enumerate(names):
            d = q[:, idx[i + 1]] - q[:, idx[i]]
            print(f"    {nm:42s} {np.median(d):7.0f}  (p10 {np.percentile(d, 10):6.0f}, p90 {np.percentile(d, 90):6.0f})")
        w = q[:, 8] - q[:, 0]
        print(f"    whole chunk                                {np.median(w):7.0f}  (p10 {np.percentile(w, 10):6.0f}, p90 {np.percentile(w, 90):6.0f})")
