#!/usr/bin/env python
"""Kernel-by-kernel anatomy of ONE decoder call from a rocprofv3 --kernel-trace CSV: name, duration, gap to the previous kernel.
usage: trace_seq.py <kernel_trace.csv> [which cast_kernel occurrence to start from (default: the 15th)]"""
import csv, re, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 15
starts = [i for i, r in enumerate(rows) if "cast_kernel" in r[2]]
if len(starts) <= which + 1:
    which = max(0, len(starts) - 2)
a, b = starts[which], starts[which + 1]
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    n = n.replace("m3r::", "")
    return n[:70]
print(f"# decoder call #{which}: kernels {a}..{b - 1} of {len(rows)}; durations / gaps in us")
tot = gaps = 0.0
agg = {}
for i in range(a, b):
    s, e, n = rows[i]
    gap = (s - rows[i - 1][1]) / 1e3 if i > 0 else 0.0
    d = (e - s) / 1e3
    tot += d; gaps += max(gap, 0.0)
    k = short(n)
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += d
    if i - a < 60:
        print(f"{i - a:4d} {d:8.2f} {gap:7.2f}  {k}")
print(f"# total kernel {tot:.1f} us, gaps {gaps:.1f} us, wall {(rows[b - 1][1] - rows[a][0]) / 1e3:.1f} us")
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"#   {c:4d} x {d / c:8.2f} us = {d:9.1f} us  {k}")
