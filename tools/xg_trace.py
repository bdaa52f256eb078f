"""Timeline of the fused cross-attention GEMM kernels (DIAGNOSTIC build: M3AE_EXTRA_HIPCC_FLAGS=-DM3AE_XG_TRACE).
Per launch: prologue (first chunks landed), main loop, epilogue -- median over workgroups, in us and shader clocks.
    touch mm-vqa-healthcare_amd/csrc/xattn.hip; M3AE_EXTRA_HIPCC_FLAGS=-DM3AE_XG_TRACE python -m m3ae_amd.build; python tools/xg_trace.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from m3ae_amd import _lib, ops  # noqa: E402
import xattn_bench as xb  # noqa: E402

L = C.CDLL(_lib.LIB_PATH)
att, store = xb.make(2.0)
B = int(os.environ.get("B", 256))
xt = torch.randn(B, xb.T, xb.D, device="cuda").to(torch.bfloat16)
xi = torch.randn(B, xb.I, xb.D, device="cuda").to(torch.bfloat16)
mt = torch.zeros(B, xb.T, device="cuda")
pd = float(os.environ.get("PDROP", 0.0))
buf = np.zeros((8, 4096, 16), dtype=np.uint64)
for name, x, y, mask in (("txt<-img", xt, xi, None), ("img<-txt", xi, xt, mt)):
    for _ in range(3):
        xb.run(att, x, y, mask, True, pd)
    torch.cuda.synchronize()
    L.m3ae_xg_trace_dump(buf.ctypes.data_as(C.c_void_p))
    xb.run(att, x, y, mask, True, pd)
    torch.cuda.synchronize()
    L.m3ae_xg_trace_dump(buf.ctypes.data_as(C.c_void_p))
    print(name)
    for slot in range(8):
        t = buf[slot].astype(np.int64)
        used = t[:, 3] > 0
        n = int(used.sum())
        if n == 0:
            continue
        t = t[used]
        rt, ck, lp = t[:, :4], t[:, 4:8], t[:, 8:16]
        d = lambda a, i, j: np.median(a[:, j] - a[:, i])
        span = (rt[:, 3].max() - rt[:, 0].min()) / 100.0
        print(f"  launch {slot}: {n:5d} WGs, kernel span {span:7.1f} us | per WG: prologue {d(rt,0,1)/100:6.2f} us, main loop {d(rt,1,2)/100:6.2f} us "
              f"({d(ck,1,2):8.0f} clk), epilogue {d(rt,2,3)/100:6.2f} us ({d(ck,2,3):8.0f} clk), total {d(rt,0,3)/100:6.2f} us; "
              f"clock {np.median((ck[:,3]-ck[:,0]) / np.maximum(rt[:,3]-rt[:,0],1)) * 100:6.0f} MHz")
        if (lp[:, 7] > 0).any():
            q = lp[lp[:, 7] > 0]
            names = ["reads issued", "dma issue", "vm wait", "lgkm wait", "barrier 1", "mfma issue", "barrier 2"]
            print("      chunk 8, wave 0 (clk): " + ", ".join(f"{nm} {np.median(q[:, i + 1] - q[:, i]):5.0f}" for i, nm in enumerate(names))
                  + f" | whole chunk {np.median(q[:, 7] - q[:, 0]):5.0f}")
    buf[:] = 0
