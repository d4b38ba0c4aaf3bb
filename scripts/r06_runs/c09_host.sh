#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 600 python scripts/probes/host_ahead.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_host_ahead.txt
