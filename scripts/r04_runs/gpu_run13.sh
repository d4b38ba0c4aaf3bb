#!/bin/bash
# r04 GPU call 13: residual-row prefetch in the middle of the K loop (split-weight RESID launches), in the step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== op tests"; M3R_RESID_PREFETCH=1 timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -1
for V in 0 1 0 1; do echo "== S=20 step M3R_RESID_PREFETCH=$V"; M3R_RESID_PREFETCH=$V timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --scenes 20 --step-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stages_ms'], d['kernel_classes']['gemm128']['ms']); print([ (r['kernel'], r['avg_launch_us']) for r in d['roofline']['per_symbol'] if '/e3/w2' in r['kernel'] or '/e1/' in r['kernel']])"; done 2>&1 | tee $O/r04_resid_prefetch.txt
echo "== done"
