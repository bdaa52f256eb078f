"""Round-4 attention kernels against the round-3 ones (m3ae_attn_desc.launch_flags = M3AE_ATTN_LEGACY_KERNELS) on the same inputs:
outputs compared bit for bit (same math, same accumulation order) and by max |difference|; then both generations timed,
interleaved in one process.   B=256 python tools/attn_gen_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402

B, H, D = int(os.environ.get("B", 256)), 12, 768


def run(L, Lk, masked, drop, legacy):
    ops.ATTN_LEGACY = legacy
    torch.manual_seed(1)
    if L == Lk:
        qkv = torch.randn(B, L, 3 * D, device="cuda").to(torch.bfloat16)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    else:
        q = torch.randn(B, L, D, device="cuda").to(torch.bfloat16)
        kv = torch.randn(B, Lk, 2 * D, device="cuda").to(torch.bfloat16)
        k, v = kv[..., :D], kv[..., D:]
        dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
        dk, dv = dkv[..., :D], dkv[..., D:]
    mask = None
    if masked:
        mask = torch.zeros(B, Lk, device="cuda")
        mask[:, Lk - 5:] = -10000.0
    o, lse = ops.attn_forward(q, k, v, H, mask, dropout=drop)
    do = torch.randn_like(o)
    ops.attn_backward(q, k, v, o, lse, do, dq, dk, dv, H, mask, dropout=drop)
    torch.cuda.synchronize()

    def t_fwd():
        ops.attn_forward(q, k, v, H, mask, dropout=drop)

    def t_bwd():
        ops.attn_backward(q, k, v, o, lse, do, dq, dk, dv, H, mask, dropout=drop)
    return (o.clone(), lse.clone(), dq.clone(), dk.clone(), dv.clone()), t_fwd, t_bwd


def time_it(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


bad = 0
for (L, Lk, masked) in ((577, 577, False), (577, 577, True), (145, 145, False), (32, 32, True), (32, 577, False), (577, 32, True), (1025, 1025, False)):
    if L > 600 and B > 32:
        continue
    for drop in (None, (0.1, 77)):
        new, fn_f, fn_b = run(L, Lk, masked, drop, False)
        old, fo_f, fo_b = run(L, Lk, masked, drop, True)
        names = ("o", "lse", "dq", "dk", "dv")
        rep = []
        for n, a, b_ in zip(names, new, old):
            same = torch.equal(a, b_)
            err = (a.float() - b_.float()).abs().max().item()
            ref = b_.float().abs().max().item()
            rep.append(f"{n} {'==' if same else f'max|d| {err:.2e} / {ref:.2e}'}")
            if err > 2e-2 * max(ref, 1e-6):
                bad += 1
        tn_f, to_f = [], []
        tn_b, to_b = [], []
        for _ in range(3):
            ops.ATTN_LEGACY = False
            tn_f.append(time_it(fn_f)); tn_b.append(time_it(fn_b))
            ops.ATTN_LEGACY = True
            to_f.append(time_it(fo_f)); to_b.append(time_it(fo_b))
        ops.ATTN_LEGACY = False
        print(f"Lq {L:4d} Lk {Lk:4d} mask {int(masked)} drop {0 if drop is None else drop[0]}: " + "  ".join(rep) +
              f" | fwd new {min(tn_f):7.1f} us old {min(to_f):7.1f} | bwd new {min(tn_b):7.1f} us old {min(to_b):7.1f}", flush=True)
print("GENERATION CHECK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
