"""Fused cross-attention sub-block (csrc/xattn.hip): accuracy against an fp32 torch reference of the reference's formulation
(bert_model.py:253-350, 353-364) on the same bf16-rounded inputs / weights, against the unfused composition (same dropout
seeds), and timing of both at the bench batch.   B=256 python tools/xattn_bench.py [--check-only]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import ops, synth  # noqa: E402
from m3ae_amd.modules.bert_model import BertAttention  # noqa: E402
from m3ae_amd.param_store import ParamStore  # noqa: E402

D, H, T, I = 768, 12, int(os.environ.get("T", 32)), int(os.environ.get("I", 577))
dev = "cuda"


def make(scale=1.0):
    att = BertAttention(D, H, 1e-12, cross=True)
    synth.fill_deterministic(att)
    with torch.no_grad():
        for n, p in att.named_parameters():   # livelier statistics than the deterministic fill: scores of a few units
            if p.dim() == 2:
                p.copy_(torch.randn_like(p) * (scale / math.sqrt(p.shape[1])))
            elif "LayerNorm.weight" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(0.1 * torch.randn_like(p))
    cfg = dict(learning_rate=1e-3, weight_decay=0.01, lr_multiplier_head=1, lr_multiplier_multi_modal=1)
    store = ParamStore(att, cfg, dev, torch.bfloat16, weight_units=att.weight_units)
    att.eval()
    return att, store


def reference(att, x, y, mask):
    """fp32, the reference's formulation, on the bf16-rounded weights."""
    f = lambda p: p.m3ae_c.float()
    sa, out = att.self, att.output
    q = x.float() @ f(sa.query.weight).t() + sa.query.bias
    k = y.float() @ f(sa.key.weight).t() + sa.key.bias
    v = y.float() @ f(sa.value.weight).t() + sa.value.bias
    B, Lq, Lk = x.shape[0], x.shape[1], y.shape[1]
    sp = lambda t, L: t.view(B, L, H, D // H).permute(0, 2, 1, 3)
    s = sp(q, Lq) @ sp(k, Lk).transpose(-1, -2) / math.sqrt(D // H)
    if mask is not None:
        s = s + mask[:, None, None, :]
    p = torch.softmax(s, -1)
    ctx = (p @ sp(v, Lk)).permute(0, 2, 1, 3).reshape(B, Lq, D)
    o = ctx @ f(out.dense.weight).t() + out.dense.bias + x.float()
    ln = out.LayerNorm
    return torch.nn.functional.layer_norm(o, (D,), ln.weight, ln.bias, ln.eps), p


def run(att, x, y, mask, fused, pdrop=0.0):
    ops.XATTN = "always" if fused else "off"
    with torch.no_grad():
        return att(x, None, y, mask, pdrop=pdrop)


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    torch.manual_seed(0)
    att, store = make(scale=float(os.environ.get("WSCALE", 2.0)))
    ok = True
    for B in (() if "--bench-only" in sys.argv else (3,)):
        xt = torch.randn(B, T, D, device=dev).to(torch.bfloat16)
        xi = torch.randn(B, I, D, device=dev).to(torch.bfloat16)
        mt = torch.zeros(B, T, device=dev)
        mt[:, T - 9:] = -10000.0
        mt[0, 5:] = -10000.0
        mi = torch.zeros(B, I, device=dev)
        mi[1, I - 100:] = -10000.0
        for name, x, y, mask in (("txt<-img", xt, xi, None), ("txt<-img masked", xt, xi, mi), ("img<-txt", xi, xt, mt),
                                 ("img<-txt nomask", xi, xt, None)):
            ref, _ = reference(att, x, y, mask)
            u = run(att, x, y, mask, False).float()
            f = run(att, x, y, mask, True).float()
            eu, ef = (u - ref).abs().max().item(), (f - ref).abs().max().item()
            ru, rf = (u - ref).pow(2).mean().sqrt().item(), (f - ref).pow(2).mean().sqrt().item()
            print(f"{name:18s}: max|err| unfused {eu:.4f} fused {ef:.4f} | rms unfused {ru:.5f} fused {rf:.5f} "
                  f"(ref rms {ref.pow(2).mean().sqrt().item():.3f})", flush=True)
            ok &= ef < max(2.0 * eu, 0.08) and rf < max(1.5 * ru, 0.01)
            # dropout: same seeds, same masks -> the two paths agree to bf16 noise
            ops.set_dropout_seed(77)
            ud = run(att, x, y, mask, False, 0.1).float()
            ops.set_dropout_seed(77)
            fd = run(att, x, y, mask, True, 0.1).float()
            ed, rd = (ud - fd).abs().max().item(), (ud - fd).pow(2).mean().sqrt().item()
            print(f"{'':18s}  dropout 0.1, same seeds: max|fused - unfused| {ed:.4f} rms {rd:.5f}; vs eval rms "
                  f"{(fd - f).pow(2).mean().sqrt().item():.3f}", flush=True)
            ok &= rd < 0.02
    print("CHECK", "ok" if ok else "FAILED", flush=True)
    if "--check-only" in sys.argv:
        return 0 if ok else 1
    B = int(os.environ.get("B", 256))
    xt = torch.randn(B, T, D, device=dev).to(torch.bfloat16)
    xi = torch.randn(B, I, D, device=dev).to(torch.bfloat16)
    mt = torch.zeros(B, T, device=dev)
    mt[:, T - 9:] = -10000.0
    gf = 17.922 / 12 * B / 1e3   # TFLOP algorithmic per direction-layer
    for pd in ((0.0,) if "--eval-only" in sys.argv else (0.0, 0.1)):
        for name, x, y, mask in (("txt<-img", xt, xi, None), ("img<-txt", xi, xt, mt)):
            tu = 1.0 if "--fused-only" in sys.argv else timeit(lambda: run(att, x, y, mask, False, pd))
            tf = timeit(lambda: run(att, x, y, mask, True, pd), int(os.environ.get("ITERS", 10)))
            extra = ""
            if name == "img<-txt":      # the round-2 chain (P through HBM) for comparison
                ops.XATTN_LEGACY_CHAIN = True
                tl = timeit(lambda: run(att, x, y, mask, True, pd), int(os.environ.get("ITERS", 10)))
                ops.XATTN_LEGACY_CHAIN = False
                extra = f"  round-2 chain {tl * 1e3:7.1f} us"
            print(f"B={B} p={pd} {name}: unfused {tu * 1e3:7.1f} us ({gf / tu * 1e3:6.0f} TF/s alg)  fused {tf * 1e3:7.1f} us "
                  f"({gf / tf * 1e3:6.0f} TF/s alg){extra}", flush=True)
    if os.environ.get("XATTN_PROFILE"):
        ops.XATTN = "auto"
        import torch.profiler as tp
        with tp.profile(activities=[tp.ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                run(att, xt, xi, None, True)
                run(att, xi, xt, mt, True)
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=20, max_name_column_width=90))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
