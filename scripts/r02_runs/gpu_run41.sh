#!/bin/bash
# retrieval multi-layer / l2norm tests; A/B of the HIP runtime's kernel-argument placement (HIP_FORCE_DEV_KERNARG) on the scene benchmark
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_retrieval_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6
for i in 0 1; do
for ka in 0 1 unset; do
if [ $ka = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$ka; fi
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-configs 2>/dev/null | tail -1 > gpurun_out/bench41.json
python - <<P
import json
d = json.loads(open("gpurun_out/bench41.json").read())
print("KERNARG=$ka", {k: d[k] for k in ("value", "ms_per_step", "stages_ms")})
P
done
done 2>&1 | tee gpurun_out/kernarg_ab.txt
