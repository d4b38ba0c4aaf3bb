#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_zz_r06_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k "context_parallel" 2>&1 | tail -4
