"""Phase timeline of the one-launch image-query kernel (csrc/xflash.hip), diagnostic build only:
    touch mm-vqa-healthcare_amd/csrc/xflash.hip; M3AE_EXTRA_HIPCC_FLAGS=-DM3AE_XF_TRACE python -m m3ae_amd.build
    B=256 python tools/xf_trace.py
Every workgroup stamps the 100-MHz real-time counter at its phase boundaries (early compute wave 0, late compute wave 4, loader
wave 8); the table is the median / p10 / p90 over workgroups of each phase's duration in microseconds."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from m3ae_amd import _lib, ops  # noqa: E402
import xattn_bench as xb  # noqa: E402


def main():
    B, I, T, D = int(os.environ.get("B", 256)), int(os.environ.get("I", 577)), 32, 768
    pd = float(os.environ.get("PDROP", 0.0))
    train = os.environ.get("TRAIN", "0") == "1"
    torch.manual_seed(0)
    att, store = xb.make(scale=2.0)
    P = att.block_params()
    xt = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
    xi = torch.randn(B * I, D, device="cuda").to(torch.bfloat16)
    mt = torch.zeros(B, T, device="cuda")
    mt[:, T - 9:] = -10000.0
    for _ in range(3):
        ops.xattn_fwd(xi, B, I, xt, T, mt, P, pd, need_bwd=train)
    torch.cuda.synchronize()
    L = _lib.lib()
    L.m3ae_xf_trace_dump.argtypes, L.m3ae_xf_trace_dump.restype = [C.c_void_p], C.c_int
    raw = np.zeros(4096 * 3 * 20, dtype=np.uint64)
    assert L.m3ae_xf_trace_dump(raw.ctypes.data) == 0
    buf = raw[:4096 * 3 * 16].reshape(4096, 3, 16)
    acc2 = raw[4096 * 3 * 16:].reshape(4096, 3, 4)
    nblk = min(4096, B * ((I + 127) // 128))
    t = buf[:nblk].astype(np.int64)
    t0 = t[:, :, 0].min()
    print(f"B={B} I={I} pdrop={pd} train={train}: {nblk} workgroups; kernel span {(t[:, :, 11].max() - t0) / 100:.1f} us")
    names = {0: "early compute (wave 0)", 1: "late compute (wave 4)", 2: "loader (wave 8)"}
    comp = [("prologue: launch -> chunk 0 landed", 0, 1), ("product 1 loop (24 chunks)", 1, 2), ("softmax in registers", 2, 3),
            ("P image (+ copies) + barrier B", 3, 4), ("pass 0 loop (12 chunks)", 4, 5), ("pass 0 epilogue", 5, 8),
            ("pass 1 loop", 8, 6), ("pass 1 epilogue", 6, 9), ("pass 2 loop", 9, 7), ("pass 2 epilogue", 7, 10),
            ("tail", 10, 11), ("whole workgroup", 0, 11)]
    load = [("product 1 (stage + wait)", 0, 2), ("first V' chunks issued -> landed", 2, 3), ("wait for barrier B", 3, 4),
            ("product 2 loop", 4, 11), ("whole", 0, 11)]
    for w in (0, 1, 2):
        print(f"-- {names[w]}")
        for label, a, b in (load if w == 2 else comp):
            d = (t[:, w, b] - t[:, w, a]) / 100.0
            print(f"   {label:38s} median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f} us")
    nchunks = {1: D // 32, 2: (D // 256) * 12}
    lab = {0: ["reads + lgkmcnt", "barrier 1", "MFMA cluster", "barrier 2"], 1: ["reads + lgkmcnt", "barrier 1", "MFMA cluster", "barrier 2"],
           2: ["issue DMA", "barrier 1", "wait landed", "barrier 2"]}
    print("-- shader clocks per chunk (median over workgroups of sum / chunks)")
    for w in (0, 1, 2):
        for prod in (1, 2):
            v = (t[:nblk, w, 12:16] if prod == 1 else acc2[:nblk, w, :].astype(np.int64)) / nchunks[prod]
            med = np.median(v, axis=0)
            print(f"   {names[w]:24s} product {prod}: " + "  ".join(f"{l} {m:6.0f}" for l, m in zip(lab[w], med)) + f"   total {med.sum():6.0f}")
    # rounds: start time of every workgroup relative to the first
    st = np.sort((t[:, 0, 0] - t0) / 100.0)
    print("workgroup start times (us), every 128th:", np.round(st[::128], 1).tolist())


if __name__ == "__main__":
    main()
