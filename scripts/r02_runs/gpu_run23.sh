#!/bin/bash
# round-2 GPU call 23: gemm256 with two blocks per CU (BN = 128 split, 2-slot ring, <= 128 VGPRs)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "== gemm tests with OCC2"
M3R_GEMM256=2 M3R_G256_BN=128 M3R_G256_OCC2=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -3
rm -f gpurun_out/gemm_occ2.txt
for i in 1 2; do
echo "-- default selection" | tee -a gpurun_out/gemm_occ2.txt
SPLIT=1 timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm_occ2.txt
echo "-- BN=128 one block per CU" | tee -a gpurun_out/gemm_occ2.txt
M3R_GEMM256=2 M3R_G256_BN=128 SPLIT=1 timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm_occ2.txt
echo "-- BN=128 two blocks per CU" | tee -a gpurun_out/gemm_occ2.txt
M3R_GEMM256=2 M3R_G256_BN=128 M3R_G256_OCC2=1 SPLIT=1 timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm_occ2.txt
done
