#!/bin/bash
# r05 GPU call 24: attn3 s_setprio experiment: 0 none, 1 the MFMA clusters raised, 2 the exp2 cluster raised
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
for nb in 0 1 2 0 1 2; do
  echo "== M3R_ATTN_PR=$nb"
  M3R_ATTN_PR=$nb timeout 300 python scripts/bench_attn.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/r05_attn_pr$nb.txt
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
M3R_ATTN_PR=0 step pr0
M3R_ATTN_PR=1 step pr1
M3R_ATTN_PR=2 step pr2
M3R_ATTN_PR=0 step pr0b
M3R_ATTN_PR=1 step pr1b
M3R_ATTN_PR=2 step pr2b
echo "== done"
