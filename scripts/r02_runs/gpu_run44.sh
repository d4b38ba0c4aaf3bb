#!/bin/bash
# gemm256: early barrier in the multiply interval (4 / 8 / 12 MFMAs before its end) -- op tests, A/B on the chip-filling shapes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
M3R_G256_EARLYBAR=2 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -3
rm -f gpurun_out/gemm256_early_ab.txt
for e in 0 1 2 3 0 2; do
echo "-- early_barrier $e" | tee -a gpurun_out/gemm256_early_ab.txt
M3R_G256_EARLYBAR=$e SPLIT=1 timeout 120 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | grep -E "variant|enc qkv|enc fc2" | tee -a gpurun_out/gemm256_early_ab.txt
done
