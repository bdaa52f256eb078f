"""EXPERIMENT: builds tools/experimental/nt_w4_proto.hip into its own shared object and times it next to the library's NT
kernel and torch.matmul on the path's shapes (plain epilogue).  Not part of the product build."""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402

so = os.path.join(HERE, "libnt_w4_proto.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
                           "-Wno-unused-value", f"-I{ROOT}/include", f"-I{ROOT}/mm-vqa-healthcare_amd/csrc", "-o", so,
                           os.path.join(HERE, "nt_w4_proto.hip")])
L = C.CDLL(so)
L.nt_w4_launch.restype = C.c_int
L.nt_w4_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = "cuda"


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def w4(x, w, y, M, N, K):
    rc = L.nt_w4_launch(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


for M, N, K in [(2048, 1024, 384), (512, 512, 1024), (8192, 8192, 8192), (147712, 3072, 768), (147712, 2304, 768), (147712, 768, 768), (147712, 768, 3072)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    y2 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K)
    w4(x, w, y2, M, N, K)
    torch.cuda.synchronize()
    same = torch.equal(y.view(torch.int16), y2.view(torch.int16))
    err = (y.float() - y2.float()).abs().max().item()
    fl = 2.0 * M * N * K
    u0, u1, u2 = t(lambda: ops.gemm(x, K, 1, w, 1, K, y, N, M, N, K)), t(lambda: w4(x, w, y2, M, N, K)), t(lambda: torch.matmul(x, w.t()))
    print(f"NT {M} x {N} x {K}: library {u0:8.1f} us {fl / u0 / 1e6:7.1f} TF/s | w4 proto {u1:8.1f} us {fl / u1 / 1e6:7.1f} TF/s | vendor {u2:8.1f} us "
          f"{fl / u2 / 1e6:7.1f} TF/s | w4 == library bitwise: {same} (max abs diff {err:.3g})", flush=True)
