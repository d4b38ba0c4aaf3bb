#!/bin/bash
# round-2 GPU call 19: fused head GEMM, DPP LayerNorm reductions; full suite; split-K fc2 A/B in situ; alt precisions
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/tests.log
for v in 1 0 1 0; do
M3R_FC2_SPLITK=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench19_$v.log 2>&1; python - <<P
import json
d = json.loads(open("gpurun_out/bench19_$v.log").read().strip().splitlines()[-1])
print("splitk", $v, d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernel_classes"].items()}, d["stages_ms"])
P
done
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > gpurun_out/bench19_alt.log 2>&1; python - <<P
import json
d = json.loads(open("gpurun_out/bench19_alt.log").read().strip().splitlines()[-1])
print("alt", d["value"], d["alt"])
P
