#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_zz_r06_gpu.py tests/test_zz_abi_gpu.py tests/test_zz_batch_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k "causal or abi or overrun or batched" > gpurun_out/r06_c18.log 2>&1; echo "rc=$?"; grep -E "passed|failed|Error|assert" gpurun_out/r06_c18.log | tail -8
grep causal_forward gpurun_out/test_metrics.jsonl | cut -c1-400
