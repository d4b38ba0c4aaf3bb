#!/bin/bash
# r05 GPU call 13: 2:4-sparse low part (M3R_SPARSE_LO) A/B on the step; the GPU test suite with it on
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
    print(tag, "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "stages", {k: round(v, 1) for k, v in d.get("stages_ms", {}).items()}, "gemm", d["roofline"]["achieved"])
    for r in d["roofline"]["per_symbol"][:14]:
        print("   ", r["kernel"], r["launches"], r["ms"], r["avg_launch_us"], r["achieved_tflops"])
except Exception as e:
    print(tag, "failed", e)
PY
}
M3R_SPARSE_LO=0 step dense28
step sparse28
M3R_SPARSE_LO=0 step dense28b
step sparse28b
step sparse20 --scenes 20
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout=900 > $O/r05_gputests2.log 2>&1; echo "rc=$?"; tail -5 $O/r05_gputests2.log | cut -c1-300
cp $O/test_metrics.jsonl $O/r05_test_metrics.jsonl 2>/dev/null
echo "== done"
