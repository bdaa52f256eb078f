import os, sys
sys.path.insert(0, "mm-vqa-healthcare_amd")
import torch
from m3ae_amd import ops, _lib
L = _lib.lib()
M, N = 147712, 768
x = torch.randn(M, N, device="cuda").to(torch.bfloat16)
out = torch.zeros(N, device="cuda")
def run():
    ops.check(L.m3ae_colsum(ops._p(x), ops._p(out), M, N, N, ops._dt(x), 0, ops._stream()), "colsum")
run(); torch.cuda.synchronize()
ref = x.float().sum(0)
print("err", ((out - ref).abs().max() / ref.abs().max()).item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f"colsum {M}x{N}: {us:.1f} us  {M*N*2/us/1e6:.2f} TB/s")
