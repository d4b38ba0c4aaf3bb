#!/bin/bash
# LN fold with mean-shifted rows: op tests, model tests, two default-precision bench lines (shift on by construction)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_edge_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5
for i in 0 1; do
python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-configs 2>/dev/null | tail -1 > gpurun_out/bench40_$i.json
python - <<P
import json
d = json.loads(open("gpurun_out/bench40_$i.json").read())
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms")}); print(d["parity_vs_cpu_oracle"]["fp16w2"])
P
done
