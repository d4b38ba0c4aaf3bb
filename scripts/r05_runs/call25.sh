#!/bin/bash
# r05 GPU call 25: attn3 with the two 16-query halves of a wave one stage apart (M3R_ATTN_LZ=2) vs the row-sum rule alone (1): tests, microbench, step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
M3R_ATTN_LZ=2 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention" 2>&1 | tail -5
for lz in 1 2 1 2; do
  echo "== M3R_ATTN_LZ=$lz"
  M3R_ATTN_LZ=$lz timeout 300 python scripts/bench_attn.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/r05_attn_halves_lz$lz.txt
done
step() {
  timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --step-only "${@:2}" > $O/r05_step_$1.json 2> $O/r05_step_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r05_step_{tag}.json").read().strip().splitlines()[-1])
    print(tag, "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "stages", {k: round(v, 1) for k, v in d.get("stages_ms", {}).items()})
except Exception as e:
    print(tag, "failed", e)
PY
}
M3R_ATTN_LZ=1 step h1
M3R_ATTN_LZ=2 step h2
M3R_ATTN_LZ=1 step h1b
M3R_ATTN_LZ=2 step h2b
echo "== done"
