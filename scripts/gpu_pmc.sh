#!/bin/bash
# HBM traffic counters (separate passes for FETCH_SIZE and WRITE_SIZE: they do not fit one pass on gfx950),
# kernel-trace only (no sys/hip traces with --pmc).  Summaries -> gpurun_out/pmc_<counter>.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for CTR in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$CTR
  timeout 420 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d gpurun_out/pmc_$CTR -o run -- \
      python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-configs --no-single > gpurun_out/pmc_$CTR.log 2>&1
  python - "$CTR" <<'PY'
import sys, glob, json, csv, re, collections
ctr = sys.argv[1]
files = glob.glob(f"gpurun_out/pmc_{ctr}/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != ctr:
                continue
            name = re.sub(r"\(.*", "", row.get("Kernel_Name", ""))[:120]
            a = agg[name]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
out = {k: {"launches": v[0], "sum": v[1], "mean_per_launch": v[1] / max(1, v[0])} for k, v in agg.items()}
json.dump({"counter": ctr, "files": files, "kernels": out}, open(f"gpurun_out/pmc_{ctr}.json", "w"), indent=1)
top = sorted(out.items(), key=lambda kv: -kv[1]["sum"])[:8]
for k, v in top:
    print(ctr, k[:70], v["launches"], round(v["mean_per_launch"], 1))
PY
  find gpurun_out/pmc_$CTR -name "*.csv" -size +8M -delete
done
