#!/bin/bash
# round-2 GPU call 14: fp16 split-KV partials, pipelined gemm96; full GPU suite + bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 300 python scripts/bench_attn.py 2>&1 | grep -E "update CA" | tee gpurun_out/attn_part16.txt
M3R_ATTN_PART16=0 timeout 300 python scripts/bench_attn.py 2>&1 | grep -E "update CA" | tee -a gpurun_out/attn_part16.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench14.log 2>&1; echo "rc=$?"; python - <<P
import json
d = json.loads(open("gpurun_out/bench14.log").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "kernel_classes", "stages_ms")})
P
