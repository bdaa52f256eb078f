"""EXPERIMENT: per-CU intake rate of LDS-DMA (and plain 16-B loads) vs issuing waves, pieces in flight, piece shape and source
residency (tools/experimental/dma_probe.hip; its own shared object, not part of the product build)."""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
import torch  # noqa: E402

so = os.path.join(HERE, "libdma_probe.so")
src = os.path.join(HERE, "dma_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
                           "-Wno-unused-value", f"-I{ROOT}/include", f"-I{ROOT}/mm-vqa-healthcare_amd/csrc", "-o", so, src])
L = C.CDLL(so)
L.dma_probe.restype = C.c_int
L.dma_probe.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                        C.c_void_p, C.c_void_p]
GRID = 256
buf = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")     # 3 GiB source
buf.random_(0, 255)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")


def run(seg, inflight, nwaves, shared, reg=0, iters=2016):
    """GB/s one CU takes in.  A wave walks `wave_span` bytes of its own region cyclically in 1-KiB pieces (row-shaped pieces: rows of
    1536 B, the reduction walked first, as a GEMM operand stream)."""
    ld, nk = 1536, (1536 // seg if seg < 1024 else 1)
    if shared:          # every workgroup re-reads the same small regions (144 KiB per wave): L2-resident
        wave_span, wg_span = 144 * 1024, 0
    else:               # private region per workgroup and wave, read once: HBM
        wave_span = iters * 1024
        wg_span = wave_span * nwaves
        if wg_span * GRID > buf.numel():
            return None
    fn = lambda: L.dma_probe(buf.data_ptr(), wg_span, wave_span, 1 if shared else 0, ld, nk, iters, nwaves, GRID, seg, inflight, reg,
                             sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    return nwaves * iters * 1024 / (ms * 1e-3) / 1e9


def main():
    print("per-CU intake, GB/s (x256 CUs = chip); B/clk at 2.1 GHz in brackets")
    for shared in (1, 0):
        print("== source:", "one small region re-read by every workgroup (L2 hits)" if shared else "private regions, read once (HBM)")
        for reg in (0, 1):
            for seg in (64, 128, 256, 1024):
                if reg and seg == 256:
                    continue
                for inflight in ((8,) if reg else (4, 8, 12)):
                    row = []
                    for nw in (1, 2, 4, 8, 12):
                        if nw * inflight > 160:
                            continue
                        r = run(seg, inflight, nw, shared, reg, 2016 if shared else 1008)
                        row.append("   n/a   " if r is None else f"{r:6.1f} [{r / 2.1:4.1f}]")
                    print(f"  {'reg-load' if reg else 'LDS-DMA '} piece = {64 * 16 // seg if seg < 1024 else 1:2d} rows x {seg:4d} B, {inflight:2d} in flight/wave:  "
                          + "  ".join(f"{nw}w {v}" for nw, v in zip((1, 2, 4, 8, 12), row)), flush=True)


if __name__ == "__main__":
    main()
