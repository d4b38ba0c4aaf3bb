#!/bin/bash
# round-2 GPU call 3: attention (163-VGPR pipelined kernel, fp8 path at op and model level), 192-column GEMM tiles
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "== attention op tests"
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "attention" > gpurun_out/attn_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/attn_tests.log
echo "== fp8 model tests"
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "fp8" > gpurun_out/fp8_model_tests.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/fp8_model_tests.log
echo "== attention A/B"
rm -f gpurun_out/attn_ab.txt
for v in 0 1 0 1; do M3R_ATTN=$v timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/attn_ab.txt; done
FP8=1 timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/attn_ab.txt
echo "== gemm 192 A/B"
rm -f gpurun_out/gemm192_ab.txt
for i in 1 2; do
SPLIT=1 timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm192_ab.txt
M3R_G256_NO192=1 SPLIT=1 timeout 300 python scripts/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm192_ab.txt
done
echo "== full tests"
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout=600 > gpurun_out/tests.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/tests.log
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench3.log 2>&1; echo "rc=$?"; tail -c 2500 gpurun_out/bench3.log
M3R_ATTN=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench3_oldattn.log 2>&1; echo "rc=$?"; tail -c 1200 gpurun_out/bench3_oldattn.log
