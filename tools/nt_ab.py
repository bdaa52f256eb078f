"""A/B of NT GEMM kernel variants on the step's shapes (per-GPU batch B, default 256): variants interleaved in ONE process
(rounds x variants), min and median per variant; every variant's output is first compared bit for bit with variant 4 (the
2-stage kernel with the same accumulation order).

    B=256 VARS=8,10 ROUNDS=5 python tools/nt_ab.py [tag]      (M=8192: other row counts; SHAPES=2304x768,...: a subset)
"""
import os
import sys
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))
import torch  # noqa: E402
from m3ae_amd import ops  # noqa: E402

VARS = tuple(int(v) for v in os.environ.get("VARS", "8,10").split(","))   # v + 100 * p: variant v with output-store policy p (1 plain, 2 nt, 3 sc1)
ROUNDS = int(os.environ.get("ROUNDS", 5))
ITERS = int(os.environ.get("ITERS", 10))
SHAPES = os.environ.get("SHAPES", "")


def time_it(fn, iters=ITERS):
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
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    B = int(os.environ.get("B", 256))
    m, dev = int(os.environ.get("M", B * 577)), "cuda"   # M=8192: the text side's rows at per-GPU batch 256
    cases = [(2304, 768, "plain"), (2304, 768, "bias"), (768, 768, "bias+res+drop"), (3072, 768, "gelu+deriv"), (3072, 768, "qgelu+pre"),
             (768, 3072, "bias+res+drop"), (3072, 768, "dmul"), (3072, 768, "dgelu"), (768, 3072, "plain"), (768, 768, "plain"), (768, 2304, "plain")]
    if SHAPES:
        keep = set(SHAPES.split(","))
        cases = [c for c in cases if f"{c[0]}x{c[1]}" in keep or c[2] in keep]
    print(f"[{tag}] M={m} variants {VARS} rounds {ROUNDS} x {ITERS} launches", flush=True)
    tot = {v: 0.0 for v in VARS}
    for (n, k, kind) in cases:
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        b = torch.randn(n, device=dev)
        aux = torch.randn(m, n, device=dev).to(torch.bfloat16)
        pre = torch.empty_like(y)
        kw = {}
        if kind == "bias":
            kw = dict(bias=b)
        elif kind == "bias+res+drop":
            kw = dict(bias=b, residual=aux, dropout=(0.1, 1234))
        elif kind == "gelu+deriv":
            kw = dict(bias=b, act=ops.ACT_GELU, preact=pre, preact_grad=True)
        elif kind == "qgelu+pre":
            kw = dict(bias=b, act=ops.ACT_QUICKGELU, preact=pre)
        elif kind == "dmul":
            kw = dict(dact_aux=aux, dact=ops.ACT_MULAUX)
        elif kind == "dgelu":
            kw = dict(dact_aux=aux, dact=ops.ACT_GELU)
        fn = lambda: ops.gemm(x, k, 1, w, 1, k, y, n, m, n, k, **kw)
        # bit identity against variant 4
        ops.GEMM_NT_VARIANT = 4
        fn()
        ref, ref_pre = y.clone(), pre.clone()
        ident = {}
        for v in VARS:
            ops.GEMM_NT_VARIANT, ops.GEMM_ST_POLICY = (v % 100 if v >= 0 else -1), (v // 100 if v >= 0 else 0)
            y.fill_(float("nan")); pre.fill_(float("nan"))
            fn()
            torch.cuda.synchronize()
            ok = torch.equal(y.view(torch.int16), ref.view(torch.int16))
            if "preact" in kw:
                ok = ok and torch.equal(pre.view(torch.int16), ref_pre.view(torch.int16))
            ident[v] = ok
        res = {v: [] for v in VARS}
        for r in range(ROUNDS):
            for v in VARS:
                ops.GEMM_NT_VARIANT, ops.GEMM_ST_POLICY = (v % 100 if v >= 0 else -1), (v // 100 if v >= 0 else 0)
                res[v].append(time_it(fn))
        line = []
        for v in VARS:
            mn, md = min(res[v]), statistics.median(res[v])
            tot[v] += md
            line.append(f"v{v}: {mn * 1e3:7.1f} / {md * 1e3:7.1f} us {2.0 * m * n * k / md / 1e9:6.0f} TF/s {'==' if ident[v] else 'DIFF'}")
        print(f"  {n:5d}x{k:5d} {kind:14s} " + " | ".join(line), flush=True)
        del x, w, y, aux, pre
    ops.GEMM_NT_VARIANT, ops.GEMM_ST_POLICY = -1, 0
    print(f"[{tag}] sum of medians: " + "  ".join(f"v{v}: {tot[v] * 1e3:8.1f} us" for v in VARS), flush=True)


if __name__ == "__main__":
    main()
