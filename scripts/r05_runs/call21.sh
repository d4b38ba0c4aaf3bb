#!/bin/bash
# r05 GPU call 21: encoder chunk size with the r05 kernels (64 views = 192 row blocks: every encoder GEMM fills its rounds exactly)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
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
step chunk40
M3R_ENC_CHUNK_ROWS=49152 step chunk64
M3R_ENC_CHUNK_ROWS=24576 step chunk32
step chunk40b
M3R_ENC_CHUNK_ROWS=49152 step chunk64b
step s32 --scenes 32
step s24 --scenes 24
echo "== done"
