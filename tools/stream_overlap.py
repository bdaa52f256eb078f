"""How the step's two HIP streams share the GPU: from a rocprofv3 --kernel-trace database (rocpd sqlite, the default output format)
of `bench.py`, per timed step: wall between the first and last kernel, busy time of each hardware queue, time both are busy,
idle gaps.     python tools/stream_overlap.py <results.db> [steps_to_skip]"""
import collections
import sqlite3
import statistics
import sys


def union(ev):
    ev = sorted(ev)
    out, (cs, ce) = [], ev[0]
    for s, e in ev[1:]:
        if s > ce:
            out.append((cs, ce)); cs, ce = s, e
        else:
            ce = max(ce, e)
    out.append((cs, ce))
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append((s, e))
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = db.execute("select name,start,end,queue_id from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if "adamw" in r[0].lower()]
    per = 6   # optimizer launches per step (six parameter-group segments)
    ends = adam[per - 1::per]
    lines = []
    for k in range(skip, len(ends) - 1):
        sel = rows[ends[k] + 1:ends[k + 1] + 1]
        qs = collections.Counter(r[3] for r in sel).most_common(2)
        by_q = {q: union([(r[1], r[2]) for r in sel if r[3] == q]) for q, _ in qs}
        allu = union([(r[1], r[2]) for r in sel])
        wall = sel[-1][2] - sel[0][1]
        q0, q1 = qs[0][0], qs[1][0]
        both = length(intersect(by_q[q0], by_q[q1]))
        gaps = [b[0] - a[1] for a, b in zip(allu, allu[1:])]
        lines.append((wall / 1e6, length(by_q[q0]) / 1e6, length(by_q[q1]) / 1e6, both / 1e6, (wall - length(allu)) / 1e6, len(sel),
                      statistics.median(gaps) / 1e3 if gaps else 0.0, len(gaps)))
    print(f"{len(lines)} steps (first {skip} skipped); per step, ms: wall | busy queue A (most launches) | busy queue B | both busy | GPU idle | launches | median idle gap us | gaps")
    for ln in lines:
        print("  %8.2f | %8.2f | %8.2f | %8.2f | %6.2f | %5d | %6.1f | %4d" % ln)
    med = [statistics.median(c) for c in zip(*lines)]
    print("median: wall %.2f ms, queue A busy %.2f, queue B busy %.2f, both %.2f (%.0f %% of B), idle %.2f ms (%.1f %% of wall)"
          % (med[0], med[1], med[2], med[3], 100 * med[3] / med[2], med[4], 100 * med[4] / med[0]))


if __name__ == "__main__":
    main()
