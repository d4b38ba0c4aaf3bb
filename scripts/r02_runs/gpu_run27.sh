#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/tests.log
timeout 300 python scripts/bench_224.py 2>&1 | grep -v amdgpu | tee gpurun_out/bench_224b.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench27.log 2>&1; python - <<P
import json
d = json.loads(open("gpurun_out/bench27.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stages_ms"])
P
