#!/bin/bash
# r06 call 16: the hand GEMMs (with their epilogues) beside the vendor's plain product on the same box, same rounds (third column of scripts/r06_gemm_persist_ab.py)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
OPT=PERSIST VALS=0,1 ROUNDS=5 timeout 900 python scripts/r06_gemm_persist_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_vendor_yardstick.txt
