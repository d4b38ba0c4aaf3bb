#!/bin/bash
# r04 GPU call 7: row-walking LayerNorm (next row in flight, gamma / beta in registers) against the one-row-per-wave kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
for R in 0 1; do M3R_LN_ROWS=$R timeout 300 python scripts/bench_ln.py 2>&1 | grep -v amdgpu.ids; done > $O/r04_ln_rows_ab.txt; cat $O/r04_ln_rows_ab.txt
echo "== layernorm op tests"; timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "layernorm or ln" 2>&1 | tail -2
for R in 0 1; do echo "== S=20 step M3R_LN_ROWS=$R"; M3R_LN_ROWS=$R timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --scenes 20 --step-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stages_ms'], d['kernel_classes']['layernorm'])"; done
echo "== done"
