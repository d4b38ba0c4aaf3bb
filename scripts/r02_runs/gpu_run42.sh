#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 120 scripts/probes/build/launch_floor 2>&1 | tee gpurun_out/launch_floor.txt
