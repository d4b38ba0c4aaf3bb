#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 300 python scripts/bench_224.py 2>&1 | grep -v amdgpu | tee gpurun_out/bench_224.txt
rm -rf gpurun_out/trace224; mkdir -p gpurun_out/trace224
N=4 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/trace224 -o t -- python scripts/bench_224.py > gpurun_out/trace224.log 2>&1
grep "224x224" gpurun_out/trace224.log | tee -a gpurun_out/bench_224.txt
python scripts/prof_summary.py $(ls gpurun_out/trace224/*.db | tail -1) | head -14 | tee -a gpurun_out/bench_224.txt
find gpurun_out/trace224 -size +5M -delete
