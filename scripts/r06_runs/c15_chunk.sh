#!/bin/bash
# r06 call 15: encoder chunk size at the S = 28 step (560 views): 40-view chunks (default: 93.75 % tile fill) vs 63 (98.4 % for every encoder GEMM and for the self attention) vs 21
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
for R in 32768 49152 16384 32768 49152; do
  M3R_ENC_CHUNK_ROWS=$R timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --step-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ENC_CHUNK_ROWS=$R value',d['value'],'ms',d['ms_per_step'],d['stages_ms'])"
done
