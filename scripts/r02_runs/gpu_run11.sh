#!/bin/bash
# round-2 GPU call 11: software-pipelined gemm96 (PF) vs lock-step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "== gemm tests"
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm96 or splitk" 2>&1 | tail -4
rm -f gpurun_out/gemm96_pf.txt
for v in 1 0 1 0; do echo "M3R_GEMM96_PF=$v" | tee -a gpurun_out/gemm96_pf.txt; M3R_GEMM96_PF=$v SPLIT=1 timeout 300 python scripts/bench_gemm_small.py 2>&1 | grep -E "fc1|fc2|qkv" | tee -a gpurun_out/gemm96_pf.txt; done
