#!/bin/bash
# r05 GPU call 8: the new default dispatch (gemm256p for plain weights and for the split shapes it fills better): op tests, step at S = 20 and S = 28, A/B against the r04 dispatch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== op tests"
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 2>&1 | tail -4
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
echo "== step"
M3R_G256P=0 M3R_G256P_SPLIT=0 step r04dispatch
step new
M3R_G256P=0 M3R_G256P_SPLIT=0 step r04dispatch_b
step new_b
step new_s28 --scenes 28
echo "== done"
