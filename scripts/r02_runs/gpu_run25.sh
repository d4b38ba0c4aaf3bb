#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
SPLIT=1 timeout 300 python scripts/bench_gemm_small.py 2>&1 | grep -v amdgpu | tee gpurun_out/gemm_small_pre.txt
timeout 300 python scripts/bench_attn.py 2>&1 | grep -E "update" | tee gpurun_out/attn_inline.txt
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "fixture or full_depth or oracle" 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench25.log 2>&1; python - <<P
import json
d = json.loads(open("gpurun_out/bench25.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernel_classes"].items()}, d["stages_ms"])
P
done
