"""A few launches of the large NT and TN GEMM shapes for counter collection (rocprofv3 --pmc ... -- python tools/gemm_pmc_probe.py),
and, with a counter_collection.csv as argument, the per-kernel averages of every counter in it."""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-vqa-healthcare_amd"))


def summarize(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            d = row["Dispatch_Id"]
            per[d][row["Counter_Name"]] += float(row["Counter_Value"])
            n = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
            meta[d] = (re.sub(r"^void ", "", n).split("(")[0], int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                       row.get("Grid_Size", "?"))
    agg = collections.defaultdict(lambda: [0, 0.0, collections.defaultdict(float)])
    for d, c in per.items():
        n, ns, grid = meta[d]
        a = agg[(n, grid)]
        a[0] += 1
        a[1] += ns
        for k, v in c.items():
            a[2][k] += v
    for (n, grid), (k, ns, c) in sorted(agg.items(), key=lambda t: -t[1][1]):
        if not (n.startswith("gemm_") or n.startswith("xg_") or n.startswith("attn_")):
            continue
        print(f"{n[:70]:70s} grid {grid:>9s} x{k:3d} avg {ns / k / 1e3:8.1f} us  " +
              "  ".join(f"{cn}={cv / k:.4g}" for cn, cv in sorted(c.items())))


def main():
    if len(sys.argv) > 1:
        return summarize(sys.argv[1])
    import torch
    from m3ae_amd import ops
    dev = "cuda"
    M = 147712
    x = torch.randn(M, 768, device=dev).to(torch.bfloat16)
    w = (torch.randn(3072, 768, device=dev) * 768 ** -0.5).to(torch.bfloat16)
    y = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, 3072, device=dev).to(torch.bfloat16)
    g = torch.zeros(3072, 768, device=dev)
    g2 = torch.zeros(768, 768, device=dev)
    a8 = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    b8 = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    c8 = torch.empty(8192, 8192, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(x, 768, 1, w, 1, 768, y, 3072, M, 3072, 768)                      # NT, K = 768 (persistent ping-pong)
        ops.gemm(a8, 8192, 1, b8, 1, 8192, c8, 8192, 8192, 8192, 8192)              # NT, 8192^3
        ops.gemm(dy, 1, 3072, x, 768, 1, g, 768, 3072, 768, M, accumulate=True)     # TN wgrad 3072 x 768, reduction 147712
        ops.gemm(x, 1, 768, x, 768, 1, g2, 768, 768, 768, M, accumulate=True)       # TN wgrad 768 x 768
        ops.gemm(a8, 1, 8192, b8, 8192, 1, c8, 8192, 8192, 8192, 8192)              # TN, 8192^3 (no split)
    torch.cuda.synchronize()
    print("last path", ops.last_gemm_path())


if __name__ == "__main__":
    main()
