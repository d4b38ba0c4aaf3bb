#!/bin/bash
# round-2 GPU call 6: attention with 48 / 64 query rows per wave (fewer LDS / DMA / loop instructions per MFMA)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
rm -f gpurun_out/attn_qw.txt
for qw in 32 48 64; do
  echo "== QW=$qw attention op tests"
  M3R_ATTN_QW=$qw timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "attention" 2>&1 | tail -3
  M3R_ATTN_QW=$qw timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/attn_qw.txt
done
for qw in 32 48 64; do
  M3R_ATTN_QW=$qw timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench6_$qw.log 2>&1; echo "rc=$?"; python - <<P
import json
d = json.loads(open("gpurun_out/bench6_$qw.log").read().strip().splitlines()[-1])
print($qw, {k: d[k] for k in ("value", "ms_per_step", "kernel_classes", "stages_ms")})
P
done
