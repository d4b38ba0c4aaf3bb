#!/usr/bin/env python
"""Turn a rocprofv3 results .db (gpurun_out/prof/*.db) into the text summary committed under profiles/."""
import glob
import re
import sqlite3
import sys

db = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/prof/*.db"))[-1]
out = sys.argv[2] if len(sys.argv) > 2 else None
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
lines = [f"# rocprofv3 --kernel-trace --stats summary ({db}); top_kernels view, durations in us; total kernel time {tot / 1e3:.2f} ms",
         f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}"]
for name, calls, total, avg, pct in rows:
    if pct < 0.01:
        continue
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^void ", "", short)[:100]
    lines.append(f"{short:100s} {calls:7d} {total / 1e3:10.3f} {avg:9.2f} {pct:6.2f}")
txt = "\n".join(lines) + "\n"
if out:
    open(out, "w").write(txt)
print(txt)
