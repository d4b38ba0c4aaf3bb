#!/bin/bash
# r06 call 6: new r06 tests (context parallel, fold256, parked fp8), fp8-skipping suites, self-attention tile sweep
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
echo "== r06 tests + fp8-touching suites"
timeout 2400 python -m pytest tests/test_zz_r06_gpu.py tests/test_zz_batch_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 -k "r06 or fp8 or context or fold256 or refuses or batched_decode" > gpurun_out/r06_c06_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r06_c06_tests.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 -k "attention_fp8 or persistent" 2>&1 | tail -3
echo "== self-attention tile sweep"
SELF_SWEEP=1 timeout 600 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c06_attn_self_sweep.txt
echo "== done"
