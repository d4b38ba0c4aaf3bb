#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "fixture or full_depth" 2>&1 | tail -3
for v in 1 1; do
M3R_LNFOLD=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench30_$v.log 2>&1; python - <<P
import json
d = json.loads(open("gpurun_out/bench30_$v.log").read().strip().splitlines()[-1])
print("lnfold", $v, d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernel_classes"].items()}, d["stages_ms"])
P
done
