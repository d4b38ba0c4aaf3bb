#!/bin/bash
# r04 GPU call 10: views/s against the number of scenes in flight (one box): where the tile grid of the sequential update fills its rounds
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
for S in 2 4 8 12 16 20 24 28 32 40 56; do timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --scenes $S --step-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; n=d['config']['scenes_in_flight_per_gpu']
print(f\"S={n:3d} {d['value']:8.2f} views/s  per scene: encode {s['encode']/n:6.2f} update {s['update']/n:6.2f} render {s['render']/n:6.2f} ms  gemm {d['kernel_classes']['gemm128']['tflops']} TF/s\")"; done 2>&1 | tee $O/r04_scenes_sweep.txt
echo "== done"
