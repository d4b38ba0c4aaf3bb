#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 60 scripts/probes/build/gemm256_trace 2>&1 | grep -E "^M |steps |^   (14|17):" | tee gpurun_out/gemm256_trace_ladder.txt
echo "== one unconditional counted wait instead of the branch ladder" | tee -a gpurun_out/gemm256_trace_ladder.txt
timeout 60 scripts/probes/build/gemm256_trace_noladder 2>&1 | grep -E "^M |steps |^   (14|17):" | tee -a gpurun_out/gemm256_trace_ladder.txt
