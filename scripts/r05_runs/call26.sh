#!/bin/bash
# r05 GPU call 26: split-KV partials merged by the last block of a query block (no attn_combine launch): tests, one-scene and streaming numbers with / without
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
for f in 0 1 0 1; do
  echo "== M3R_ATTN_FUSED_COMBINE=$f"
  M3R_ATTN_FUSED_COMBINE=$f timeout 300 python scripts/bench_attn.py 2>&1 | grep "update CA 1v" | tee -a $O/r05_attn_fused_combine_$f.txt
  M3R_ATTN_FUSED_COMBINE=$f timeout 400 python bench.py --gpus 1 --steps 10 --warmup 3 --scenes 1 --step-only > $O/r05_single_fc$f.json 2> $O/r05_single_fc$f.err
  python - "$f" <<'PY'
import json, sys
f = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r05_single_fc{f}.json").read().strip().splitlines()[-1])
    print("single", d.get("value"), d.get("ms_per_step"), d.get("stages_ms"), {k: (round(v["ms"], 2), v["calls"]) for k, v in d.get("kernel_classes", {}).items()})
except Exception as e:
    print("failed", e, open(f"gpurun_out/r05_single_fc{f}.err").read()[-400:])
PY
done
echo "== done"
