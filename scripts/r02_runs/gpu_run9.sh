#!/bin/bash
# round-2 GPU call 9: attn3 ablations (where do the remaining cycles go) and QW = 48 / 64
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
rm -f gpurun_out/attn3_abl.txt
for a in 0 1 2 3 4; do
  echo "ABL=$a" | tee -a gpurun_out/attn3_abl.txt
  M3R_ATTN_ABL=$a timeout 300 python scripts/bench_attn.py 2>&1 | grep -E "render CA 20v nk15360  |enc SA" | tee -a gpurun_out/attn3_abl.txt
done
for q in 48 64; do
  echo "QW=$q" | tee -a gpurun_out/attn3_abl.txt
  M3R_ATTN_QW=$q timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "attention and not fp8" 2>&1 | tail -2
  M3R_ATTN_QW=$q timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/attn3_abl.txt
done
