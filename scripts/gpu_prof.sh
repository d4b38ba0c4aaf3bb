#!/bin/bash
# rocprofv3 kernel-trace summary of the benchmark command (per-kernel time); output copied to gpurun_out/prof_*
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-configs > gpurun_out/prof_bench.log 2>&1
tail -3 gpurun_out/prof_bench.log
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
# keep only the small summaries (the raw trace is large)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
