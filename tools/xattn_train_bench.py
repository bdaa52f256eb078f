"""Training-mode timing of the cross-attention sub-block inside a BertCrossLayer pair: fused forward + backward vs the
composition (B=256, dropout 0.1), and with rocprofv3 --kernel-trace --stats the per-kernel split."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402
import xattn_bench as xb  # noqa: E402

ops.XATTN = "always"   # also where the product's auto rule prefers the composition (64 text tokens)
att, store = xb.make(2.0)
att.train()
B = int(os.environ.get("B", 256))
xt = torch.randn(B, xb.T, xb.D, device="cuda").to(torch.bfloat16)
xi = torch.randn(B, xb.I, xb.D, device="cuda").to(torch.bfloat16)
mt = torch.zeros(B, xb.T, device="cuda")
mt[:, xb.T - 9:] = -10000.0
P = att.block_params()


def fb(x, y, mask, fused, pd):
    Bq, L, D = x.shape
    Lo = y.shape[1]
    h2, o2 = x.view(Bq * L, D), y.view(Bq * Lo, D)
    out, saved = ops._attn_sub_fwd(h2, Bq, L, o2, Lo, mask, P, pd, fused_cross=fused)
    store.zero_grad() if False else None
    return ops._attn_sub_bwd(out, saved, Bq, L, Lo, P)


for pd in (0.1, 0.0):
    for name, x, y, mask in (("txt<-img", xt, xi, None), ("img<-txt", xi, xt, mt)):
        ts = []
        for fused in (False, True):
            if os.environ.get("FUSED_ONLY") and not fused:
                ts.append(1.0)
                continue
            ts.append(xb.timeit(lambda: fb(x, y, mask, fused, pd), int(os.environ.get("ITERS", 5))))
        print(f"B={B} T={xb.T} p={pd} {name} fwd+bwd: composition {ts[0] * 1e3:7.1f} us  fused {ts[1] * 1e3:7.1f} us", flush=True)
