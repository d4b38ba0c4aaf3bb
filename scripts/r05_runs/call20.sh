#!/bin/bash
# r05 GPU call 20: GPU test suite (no -x: all failures), quick step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
rm -f $O/test_metrics.jsonl
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 > $O/r05_gputests.log 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED" $O/r05_gputests.log | tail -12
cp $O/test_metrics.jsonl $O/r05_test_metrics.jsonl 2>/dev/null
timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --step-only > $O/r05_step_cat.json 2> $O/r05_step_cat.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_step_cat.json").read().strip().splitlines()[-1])
print("value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "stages", d.get("stages_ms"))
PY
echo "== done"
