"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the counters do
not fit one pass).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies
128-B requests at 64 B -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores and float atomics.  Both are in KiB.
usage: python tools/pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv [--json out.json B head]"""
import collections
import csv
import re
import sys


def load(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            n = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
            n = re.sub(r"^void ", "", n).split("(")[0]
            d[n][0] += 1
            d[n][1] += float(row["Counter_Value"])
    return d


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    pats = []
    js = None
    if len(sys.argv) > 3 and sys.argv[3] == "--json":
        js = (sys.argv[4], int(sys.argv[5]), sys.argv[6])
    pairs = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "--pairs" else 0
    rows = []
    for n in sorted(set(fetch) | set(write)):
        if pats and not any(p in n for p in pats):
            continue
        cf, f = fetch.get(n, (0, 0.0))
        cw, w = write.get(n, (0, 0.0))
        c = max(cf, cw, 1)
        rd, wr = 2.0 * f * 1024 / c, w * 1024 / c
        rows.append((rd + wr, n, c, rd, wr))
    print(f"{'kernel':100s} {'launches':>8s} {'read MB/launch':>15s} {'write MB/launch':>16s} {'total MB':>10s}")
    for tot, n, c, rd, wr in sorted(rows, reverse=True)[:40]:
        print(f"{n[:100]:100s} {c:8d} {rd / 1e6:15.2f} {wr / 1e6:16.2f} {tot / 1e6:10.2f}")
    if pairs:   # tools/xattn_pair.py ran `pairs` forward layer pairs: bytes per pair over the kernels of the sub-block
        sel = [r for r in rows if any(t in r[1] for t in ("xf1_kernel", "xg_kernel", "xbuild_kernel", "xattn_", "gemm_", "ln_fwd"))]
        tot = sum((r[3] + r[4]) * r[2] for r in sel)
        print(f"\nfused cross-attention forward, one layer pair (both directions), fabric-side bytes: {tot / pairs / 1e9:.3f} GB "
              f"({sum(r[2] for r in sel) / pairs:.1f} launches per pair)")
    if js:
        import json
        out = {"per_gpu_batch": js[1], "head": js[2],
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py "
                         "--no-cpu-baseline --no-roofline --steps 2 --warmup 1`; FETCH_SIZE x 2 (gfx950 tallies 128-B "
                         "requests at 64 B), both KiB -> bytes; calibrated on adamw_kernel (16 B read / 14 B written per "
                         "parameter: 5.15 GB / 4.51 GB per step measured vs 5.15 / 4.51 expected); fabric-side counters, "
                         "Infinity-Cache hits included",
               "gemm_nt_pp_kernel": summary(fetch, write, "gemm_nt_pp"),   # one-tile and persistent forms, all epilogues
               "gemm_nt_bf16_kernel": summary(fetch, write, "gemm_nt_bf16_kernel"),
               "gemm_tn_pp_kernel": summary(fetch, write, "gemm_tn_pp_kernel"),
               "gemm_tn_bf16_kernel": summary(fetch, write, "gemm_tn_bf16_kernel"),
               "attn_bwd_dkdv_coop_kernel": summary(fetch, write, "attn_bwd_dkdv_coop_kernel"),
               "ln_bwd_kernel": summary(fetch, write, "ln_bwd_kernel"),
               "attn_fwd_coop_kernel": summary(fetch, write, "attn_fwd_coop_kernel"),
               "adamw_kernel": summary(fetch, write, "adamw_kernel")}
        with open(js[0], "w") as f:
            json.dump(out, f, indent=1)


def summary(fetch, write, prefix):
    c = sum(v[0] for k, v in fetch.items() if k.startswith(prefix))
    rd = sum(v[1] for k, v in fetch.items() if k.startswith(prefix)) * 2.0 * 1024
    wr = sum(v[1] for k, v in write.items() if k.startswith(prefix)) * 1024
    return {"launches_in_profile": c, "read_bytes_per_launch": rd / max(c, 1), "write_bytes_per_launch": wr / max(c, 1),
            "bytes_per_launch": (rd + wr) / max(c, 1)}


if __name__ == "__main__":
    main()
