#!/bin/bash
# round-2 GPU call 8: attn3_kernel v2 (nm state, single max, 3 waves/SIMD)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "== attn3 op tests"
M3R_ATTN=2 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "attention" 2>&1 | tail -5
for v in 2 0 2; do
  M3R_ATTN=$v timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/attn3_ab.txt
done
for v in 2; do
  M3R_ATTN=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench8_$v.log 2>&1; echo "rc=$?"; python - <<P
import json
d = json.loads(open("gpurun_out/bench8_$v.log").read().strip().splitlines()[-1])
print($v, {k: d[k] for k in ("value", "ms_per_step", "kernel_classes", "stages_ms")})
P
done
