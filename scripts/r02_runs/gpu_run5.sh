#!/bin/bash
# round-2 GPU call 5: gemm96 (96x96 tiles, split-K fc2 + LayerNorm slab reduction), attention cycle trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "== gemm / layernorm op tests"
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm or layernorm" > gpurun_out/gemm_tests.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/gemm_tests.log
echo "== small-M GEMM microbench"
for v in 1 0 1 0; do echo "M3R_GEMM96=$v"; M3R_GEMM96=$v SPLIT=1 timeout 300 python scripts/bench_gemm_small.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm96_ab.txt; done
echo "== attention trace"
for args in "1 7680 7" "1 14592 8" "20 15360 1"; do timeout 120 scripts/probes/build/attn_trace $args 2>&1 | tee -a gpurun_out/attn_trace.txt; done
echo "== model tests"
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_zz_drivers_gpu.py tests/test_zz_batch_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/model_tests.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/model_tests.log
echo "== bench"
for v in 1 0; do
M3R_FC2_SPLITK=$v M3R_GEMM96=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench5_$v.log 2>&1; echo "rc=$?"; python - <<P
import json
d = json.loads(open("gpurun_out/bench5_$v.log").read().strip().splitlines()[-1])
print($v, {k: d[k] for k in ("value", "ms_per_step", "kernel_classes", "stages_ms")})
P
done
