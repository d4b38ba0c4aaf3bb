#!/bin/bash
# round-2 GPU call 1: fp8 semantics probe, full GPU test suite (incl. the new full-depth fixtures), bench with configs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "== probe"; timeout 60 scripts/probes/build/fp8_probe > gpurun_out/fp8_probe.txt 2>&1; echo "rc=$?"; head -8 gpurun_out/fp8_probe.txt
echo "== tests"
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout=600 > gpurun_out/tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/tests.log
echo "== bench"
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -c 6000 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
echo "== done"
