#!/bin/bash
# r05 GPU call 14: the 256 x 256 sparse-low-part kernel: parity test, step A/B (M3R_SPARSE_256 = 0 / 1)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -p no:cacheprovider -k "sparse_low" 2>&1 | tail -15
step() {
  timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --step-only "${@:2}" > $O/r05_step_$1.json 2> $O/r05_step_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r05_step_{tag}.json").read().strip().splitlines()[-1])
    print(tag, "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "stages", {k: round(v, 1) for k, v in d.get("stages_ms", {}).items()}, "gemm", d["roofline"]["achieved"])
    for r in d["roofline"]["per_symbol"][:10]:
        print("   ", r["kernel"], r["launches"], r["ms"], r["avg_launch_us"], r["achieved_tflops"])
except Exception as e:
    print(tag, "failed", e, open(f"gpurun_out/r05_step_{tag}.err").read()[-600:])
PY
}
M3R_SPARSE_256=0 step sp128
step sp256
M3R_SPARSE_256=0 step sp128b
step sp256b
step sp256_s20 --scenes 20
echo "== done"
