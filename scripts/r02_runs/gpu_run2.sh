#!/bin/bash
# round-2 GPU call 2: attention kernels -- op parity (old / new / fp8), A/B microbench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "== attention op tests (new kernel)"
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -p no:cacheprovider -k "attention" > gpurun_out/attn_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/attn_tests.log
echo "== A/B"
for v in 0 1; do M3R_ATTN=$v timeout 300 python scripts/bench_attn.py 2>&1 | tee -a gpurun_out/attn_ab.txt; done
FP8=1 timeout 300 python scripts/bench_attn.py 2>&1 | tee -a gpurun_out/attn_ab.txt
for v in 0 1; do M3R_ATTN=$v timeout 300 python scripts/bench_attn.py 2>&1 | tee -a gpurun_out/attn_ab.txt; done
echo "== model tests with the new kernel"
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -p no:cacheprovider > gpurun_out/model_tests.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/model_tests.log
echo "== bench (new attention)"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > gpurun_out/bench_attn2.log 2>&1; echo "rc=$?"; tail -c 3000 gpurun_out/bench_attn2.log
