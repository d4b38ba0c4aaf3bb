#!/bin/bash
# round-2 GPU call 10: kernel-by-kernel durations of one update call + kernel stats with the current kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
rm -rf gpurun_out/trace; mkdir -p gpurun_out/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-alt > gpurun_out/trace.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1); echo "$f"
python - "$f" <<'P'
import csv, re, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Workgroup_Size_X"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "cast_kernel" in r[2]]
cands = [k for k in range(len(starts) - 1) if 140 < starts[k + 1] - starts[k] < 200]
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    return n.replace("_ZN3m3r", "").replace("EEvNS_8GemmArgsE", "").replace("EEvNS_8AttnArgsEiii", "")[:64]
for which in (cands[len(cands) // 4], cands[-2]):
    a, b = starts[which], starts[which + 1]
    print(f"# decoder call {which}: {b - a} kernels")
    tot = 0
    agg = collections.OrderedDict()
    for i in range(a, b):
        s, e, n, g, w = rows[i]
        d = (e - s) / 1e3; tot += d
        k = f"{short(n)} grid {g}x{w}"
        agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += d
        if 14 <= i - a < 28: print(f"{i - a:4d} {d:8.2f} {k}")
    print(f"# total kernel time {tot:.1f} us")
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"#   {c:4d} x {d / c:8.2f} us = {d:9.1f} us  {k}")
P
find gpurun_out/trace -name "*.csv" -size +30M -delete
