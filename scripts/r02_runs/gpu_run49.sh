#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 60 scripts/probes/build/gemm256_trace 2>&1 | grep -E "^M |steps |^   (14|17):" | tee gpurun_out/gemm256_trace_fewcus.txt
