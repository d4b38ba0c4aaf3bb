#!/bin/bash
# r04 GPU call 11: plain-weight GELU launches (fc1) with two 256 x 128 blocks per CU against gemm256k
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
for V in 0 1; do echo "== plain fp16 shapes, M3R_PLAIN_GELU_OCC2=$V"; PLAIN16=1 M3R_PLAIN_GELU_OCC2=$V timeout 300 python scripts/exp_gemm256.py 2>&1 | grep -v amdgpu.ids | grep "fc1\|big shapes"; done 2>&1 | tee $O/r04_plain_gelu_occ2.txt
for V in 0 1; do echo "== S=20 step M3R_PLAIN_GELU_OCC2=$V"; M3R_PLAIN_GELU_OCC2=$V timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --scenes 20 --step-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stages_ms'], d['kernel_classes']['gemm128'])"; done 2>&1 | tee -a $O/r04_plain_gelu_occ2.txt
echo "== done"
