#!/bin/bash
# r05 GPU call 22: attn3 with the references moved on the tile's row sums (M3R_ATTN_LZ=1) vs on the per-lane score maxima (0): tests, microbench, step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -5
for lz in 0 1 0 1; do
  echo "== M3R_ATTN_LZ=$lz"
  M3R_ATTN_LZ=$lz timeout 300 python scripts/bench_attn.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/r05_attn_lz$lz.txt
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
M3R_ATTN_LZ=0 step lz0
M3R_ATTN_LZ=1 step lz1
M3R_ATTN_LZ=0 step lz0b
M3R_ATTN_LZ=1 step lz1b
echo "== done"
