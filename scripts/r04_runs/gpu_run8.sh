#!/bin/bash
# r04 GPU call 8: render pass of the S = 20 step cut into k scenes per native call (MALL residency of the activations vs rounds per launch)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
for K in 20 10 4 2 1; do echo "== S=20 step, render $K scenes per call"; M3R_RENDER_SCENES=$K timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --scenes 20 --step-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_classes'])"; done 2>&1 | tee $O/r04_render_chunks.txt
echo "== done"
