"""Placement and timeline of the dual NT kernel's tiles (m3ae_set_tuning key 5): every workgroup overwrites 28 B of its
tile with (HW_ID, XCC_ID, start, main-loop end, epilogue end, launch index, shader clocks); 100-MHz ticks.  Prints, per stagger
setting, how the two resident workgroups of a CU overlap."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from m3ae_amd import _lib, ops  # noqa: E402

B = int(os.environ.get("B", 256))
M = B * 577


def main():
    L = _lib.lib()
    dev = "cuda"
    n, k = int(os.environ.get("N", 3072)), int(os.environ.get("K", 768))
    x = torch.randn(M, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    y = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, device=dev)
    pre = torch.empty_like(y)
    L.m3ae_set_tuning(0, 9)
    L.m3ae_set_tuning(5, 1)
    for tk in (0, 600, -1):
        L.m3ae_set_tuning(4, tk)
        for _ in range(2):
            ops.gemm(x, k, 1, w, 1, k, y, n, M, n, k, bias=b, act=ops.ACT_GELU, preact=pre)
        torch.cuda.synchronize()
        tm, tn = (M + 127) // 128, n // 256
        v = y.view(torch.int16).cpu().numpy().view(np.uint16).reshape(M, n)
        rec = v[::128][:tm].reshape(tm, tn, 256)[:, :, :14].copy().view(np.uint32).reshape(-1, 7).astype(np.int64)
        hw, xcc, t0, t1, t2, blk, clk = rec.T
        t_base = t0.min()
        cu = (xcc & 0xf) * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xf)   # xcc, se, sh, cu
        print(f"stagger {tk}: {len(rec)} tiles on {len(set(cu.tolist()))} CUs; kernel span {(t2.max() - t_base) / 100:.1f} us; "
              f"main loop {np.median(t1 - t0) / 100:.2f} us median, epilogue {np.median(t2 - t1) / 100:.2f} us median")
        mhz = clk / np.maximum(t2 - t0, 1) * 100.0
        print(f"   shader clock while the tile ran (s_memtime / s_memrealtime): median {np.median(mhz):.0f} MHz, "
              f"p10 {np.percentile(mhz, 10):.0f}, p90 {np.percentile(mhz, 90):.0f}")
        # overlap on each CU: fraction of a workgroup's epilogue time during which the other resident is in its main loop
        fr, both_epi = [], []
        first = True
        for c in sorted(set(cu.tolist())):
            idx = np.nonzero(cu == c)[0]
            ev = sorted((t0[i], t1[i], t2[i], (hw[i] & 0xf), blk[i]) for i in idx)
            if first:
                first = False
                print("   CU", hex(c), "first tiles (start, main end, epi end in us; wave slot; launch index):")
                for e in ev[:8]:
                    print(f"      {(e[0] - t_base) / 100:8.2f} {(e[1] - t_base) / 100:8.2f} {(e[2] - t_base) / 100:8.2f}   slot {e[3]}  wg {e[4]}")
            for i, e in enumerate(ev):
                ov = 0
                for j, f in enumerate(ev):
                    if i != j and f[0] < e[2] and f[1] > e[1]:
                        ov += max(0, min(e[2], f[1]) - max(e[1], f[0]))
                if e[2] > e[1]:
                    fr.append(ov / (e[2] - e[1]))
        print(f"   epilogue time covered by the other workgroup's main loop: mean {np.mean(fr):.2f}")
    L.m3ae_set_tuning(5, 0)
    L.m3ae_set_tuning(4, -1)
    L.m3ae_set_tuning(0, -1)


if __name__ == "__main__":
    main()
