#!/bin/bash
# round-2 profile artifacts: rocprofv3 kernel stats of the bench command, HBM traffic PMC passes, attention SQ counters
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
rm -rf gpurun_out/prof gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
bash scripts/gpu_prof.sh 2>&1 | tail -5
python scripts/prof_summary.py "" gpurun_out/r02_bench_kernel_stats.txt > /dev/null 2>&1 || python scripts/prof_summary.py $(ls gpurun_out/prof/*.db 2>/dev/null | tail -1) gpurun_out/r02_bench_kernel_stats.txt | head -30
bash scripts/gpu_pmc.sh 2>&1 | tail -20
python scripts/pmc_summary.py gpurun_out/r02_pmc_traffic.json | head -12
bash scripts/gpu_pmc_attn.sh 2>&1 | tee gpurun_out/r02_attn_pmc_raw.txt | tail -30
find gpurun_out -name "*.db" -size +20M -delete; find gpurun_out -name "*.csv" -size +8M -delete
