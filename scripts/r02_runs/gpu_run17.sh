#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "attention or more_views or fixture" 2>&1 | tail -4
timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu | tee gpurun_out/attn_max16.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench17.log 2>&1; echo "rc=$?"; python - <<P
import json
d = json.loads(open("gpurun_out/bench17.log").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "kernel_classes", "stages_ms")})
P
