#!/bin/bash
# r05 GPU call 23: attn3 K/V tiles staged two tiles ahead (three buffers, M3R_ATTN_NB=3) vs one (2), both with the row-sum reference rule
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -5
for nb in 2 3 2 3; do
  echo "== M3R_ATTN_NB=$nb"
  M3R_ATTN_NB=$nb timeout 300 python scripts/bench_attn.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/r05_attn_nb$nb.txt
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
M3R_ATTN_NB=2 step nb2
M3R_ATTN_NB=3 step nb3
M3R_ATTN_NB=2 step nb2b
M3R_ATTN_NB=3 step nb3b
echo "== done"
