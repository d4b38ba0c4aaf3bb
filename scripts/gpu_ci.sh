#!/bin/bash
# Staged GPU check used through gpurun: op-level parity first, the model-level parity / smoke / bench only if it passed.
# Logs go to gpurun_out/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
STAGE=${1:-all}
echo "== rocminfo"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "gfx950|Compute Unit" | head -4
nproc
if [ "$STAGE" = "all" ] || [ "$STAGE" = "ops" ]; then
  echo "== ops"
  timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 > gpurun_out/ops.log 2>&1; rc=$?; tail -40 gpurun_out/ops.log
  if [ $rc -ne 0 ] && [ "$STAGE" = "all" ]; then echo "ops failed (rc=$rc): stopping"; exit 1; fi
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "model" ]; then
  echo "== model"
  timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 > gpurun_out/model.log 2>&1; rc_model=$?; tail -40 gpurun_out/model.log
  echo "== smoke"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20 | tee gpurun_out/smoke.log
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "bench" ]; then
  echo "== bench"
  timeout 1200 python bench.py --gpus 1 --steps 3 --warmup 1 2>&1 | tail -20 | tee gpurun_out/bench.log
fi
echo "== done"
exit ${rc_model:-0}
