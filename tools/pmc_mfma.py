"""MFMA-pipe utilisation and effective clock per kernel from one rocprofv3 PMC pass
(`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`, /opt/skills/guides/MI355X_MICROARCH.md: the busy counter adds one per
cycle an MFMA occupies a SIMD's matrix pipe, summed over the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs):
    clock      = GRBM_GUI_ACTIVE / 8 / kernel duration
    MFMA util  = SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs x 4 SIMDs x GRBM_GUI_ACTIVE / 8)
usage: python tools/pmc_mfma.py mfma_counter_collection.csv [out.json]"""
import collections
import csv
import json
import re
import sys

CUS, SIMDS, XCDS = 256, 4, 8


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch -> counter -> value
    meta = {}
    with open(sys.argv[1], newline="") as f:
        for row in csv.DictReader(f):
            d = row["Dispatch_Id"]
            per[d][row["Counter_Name"]] += float(row["Counter_Value"])
            n = re.sub(r"\(anonymous namespace\)::", "", row["Kernel_Name"])
            n = re.sub(r"^void ", "", n).split("(")[0]
            meta[d] = (n, int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])   # kernel -> launches, busy, gui, ns
    for d, c in per.items():
        n, ns = meta[d]
        a = agg[n]
        a[0] += 1
        a[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
        a[3] += ns
    rows = []
    for n, (k, busy, gui, ns) in agg.items():
        if gui <= 0 or ns <= 0:
            continue
        cyc = gui / XCDS
        rows.append((ns, n, k, busy / (CUS * SIMDS * cyc), cyc / ns * 1e3))
    rows.sort(reverse=True)
    print(f"{'kernel':90s} {'launches':>8s} {'total ms':>9s} {'MFMA util':>10s} {'clock MHz':>10s}")
    for ns, n, k, util, mhz in rows[:30]:
        print(f"{n[:90]:90s} {k:8d} {ns / 1e6:9.2f} {util:10.3f} {mhz:10.0f}")
    if len(sys.argv) > 2:
        out = {"method": __doc__.split("usage")[0].strip(),
               "kernels": {n: {"launches": k, "total_ms": ns / 1e6, "mfma_util": util, "clock_mhz": mhz}
                           for ns, n, k, util, mhz in rows[:30]}}
        with open(sys.argv[2], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
