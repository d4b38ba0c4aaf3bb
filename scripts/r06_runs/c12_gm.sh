#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
for V in 8 16 2; do
  echo "== G256_GM 4 vs $V"
  OPT=G256_GM VALS=4,$V ROUNDS=5 timeout 900 python scripts/r06_gemm_persist_ab.py 2>&1 | grep -v amdgpu.ids | sed 's/   | vendor.*//' | tee gpurun_out/r06_gm_ab_$V.txt
done
