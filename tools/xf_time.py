"""Times the image-query direction of the fused cross-attention forward (m3ae_xattn_fwd, dir 1) at the bench batch, the launches
of one call itemised with the torch profiler.   B=256 python tools/xf_time.py [tag]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402
import xattn_bench as xb  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    B, I, T, D = int(os.environ.get("B", 256)), int(os.environ.get("I", 577)), 32, 768
    torch.manual_seed(0)
    att, store = xb.make(scale=2.0)
    P = att.block_params()
    xt = torch.randn(B * T, D, device="cuda").to(torch.bfloat16)
    xi = torch.randn(B * I, D, device="cuda").to(torch.bfloat16)
    mt = torch.zeros(B, T, device="cuda")
    mt[:, T - 9:] = -10000.0
    for direction, (h, L, o, Lo, m) in (("img<-txt", (xi, I, xt, T, mt)), ("txt<-img", (xt, T, xi, I, None))):
        fn = lambda: ops.xattn_fwd(h, B, L, o, Lo, m, P, 0.0, need_bwd=False)
        ms = xb.timeit(fn, 20)
        import torch.profiler as tp
        with tp.profile(activities=[tp.ProfilerActivity.CUDA]) as prof:
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
        rows = sorted(((e.key, e.device_time_total / e.count, e.count // 5) for e in prof.key_averages() if e.device_time_total > 0),
                      key=lambda r: -r[1] * r[2])
        print(f"[{tag}] B={B} {direction}: {ms * 1e3:7.1f} us per call")
        for k, us, n in rows[:9]:
            print(f"      {us:8.1f} us x{n}  {k[:110]}")


if __name__ == "__main__":
    main()
