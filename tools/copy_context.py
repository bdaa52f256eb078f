"""Which kernels run right before / after each __amd_rocclr_copyBuffer in a rocprofv3 kernel trace (to find their origin)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60] for r in rows]
ctx = collections.Counter()
dur = collections.defaultdict(float)
for i, n in enumerate(names):
    if "copyBuffer" in n:
        k = (names[i - 1] if i else "", names[i + 1] if i + 1 < len(names) else "")
        ctx[k] += 1
        dur[k] += (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
for k, c in ctx.most_common(25):
    print(f"x{c:4d}  {dur[k] / c:8.1f} us avg   after [{k[0]}]   before [{k[1]}]")
