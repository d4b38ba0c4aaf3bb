#!/bin/bash
# the default bench line on the library as committed at the end of the round
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
SECONDS=0; timeout 200 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench rc=$? wall=${SECONDS}s"
tail -1 gpurun_out/bench_default.log > gpurun_out/r02_bench_line.json
python - <<'P'
import json
d = json.loads(open("gpurun_out/r02_bench_line.json").read())
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms")}); print(d["roofline"]["achieved"], d["parity_vs_cpu_oracle"]["fp16w2"]["rel_inf"], d["cpu_baseline"]["value"])
P
