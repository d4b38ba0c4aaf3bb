#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/tests.log
for v in 1 0 1 0; do
M3R_G256_GELU_OCC2=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench24_$v.log 2>&1; python - <<P
import json
d = json.loads(open("gpurun_out/bench24_$v.log").read().strip().splitlines()[-1])
print("gelu_occ2", $v, d["value"], d["ms_per_step"], d["stages_ms"])
P
done
