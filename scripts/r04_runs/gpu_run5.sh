#!/bin/bash
# r04 GPU call 5: two blocks per CU (OCC = 2, 256 x 128 tiles) for the non-GELU split-weight epilogues; new edge test; traffic counters via raw TCC_EA0 requests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
for MSK in 0 8 1 4 13; do echo "== split shapes, M3R_OCC2_EPI=$MSK"; SPLIT=1 M3R_OCC2_EPI=$MSK timeout 300 python scripts/exp_gemm256.py 2>&1 | grep -v amdgpu.ids | grep -v "tail M\|enc18\|k192"; done > $O/r04_occ2_ab.txt 2>&1; cat $O/r04_occ2_ab.txt
for MSK in 0 8 13; do echo "== S=20 step M3R_OCC2_EPI=$MSK"; M3R_OCC2_EPI=$MSK timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --scenes 20 --step-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stages_ms'], d['kernel_classes']['gemm128'])"; done
echo "== new edge test"; timeout 600 python -m pytest tests/test_edge_gpu.py -m gpu -q -p no:cacheprovider -k "token_constant" 2>&1 | tail -3
echo "== TCC_EA0 counters"; rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*" | sort -u | head -40 | tr '\n' ' '
echo "== done"
