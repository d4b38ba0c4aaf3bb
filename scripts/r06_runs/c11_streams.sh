#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 900 python scripts/probes/two_streams_update.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_two_streams_update.txt
