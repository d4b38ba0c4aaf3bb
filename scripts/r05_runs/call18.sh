#!/bin/bash
# r05 GPU call 18: straight-line RoPE epilogue: op tests (RoPE parity, bit identity), per-shape timing, step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6
timeout 300 python scripts/bench_sparse_gemm.py 2>&1 | grep -v amdgpu.ids | grep "qkv"
step() {
  timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --step-only "${@:2}" > $O/r05_step_$1.json 2> $O/r05_step_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r05_step_{tag}.json").read().strip().splitlines()[-1])
    print(tag, "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "stages", {k: round(v, 1) for k, v in d.get("stages_ms", {}).items()}, "gemm", d["roofline"]["achieved"])
    for r in d["roofline"]["per_symbol"][:6]:
        print("   ", r["kernel"], r["launches"], r["ms"], r["avg_launch_us"], r["achieved_tflops"])
except Exception as e:
    print(tag, "failed", e, open(f"gpurun_out/r05_step_{tag}.err").read()[-600:])
PY
}
step rope1
step rope2
echo "== done"
