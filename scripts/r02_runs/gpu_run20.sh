#!/bin/bash
# round-2: the default benchmark command exactly as the driver runs it (N = 1), timed by the shell
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "rc=$? wall=${SECONDS}s"

tail -1 gpurun_out/bench_default.log > gpurun_out/r02_bench_line.json
python - <<'P'
import json
d = json.loads(open("gpurun_out/r02_bench_line.json").read())
print({k: d[k] for k in ("value", "ms_per_step", "dtype", "roofline", "cpu_baseline")})
print(d["parity_vs_cpu_oracle"]["fp16w2"] if d.get("parity_vs_cpu_oracle") else None)
print([(c["config"][:40], c.get("value") or c.get("modes")) for c in d["configs"]])
P
