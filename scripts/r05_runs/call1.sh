#!/bin/bash
# r05 GPU call 1: vendor yardstick; gemm256p (phase-staggered 64-deep kernel) against gemm256k / gemm256 on the chip-filling shapes (bits + time);
# the S = 20 step with each variant; encoder chunk size.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
echo "== yardstick"
timeout 400 python scripts/yardstick_vendor.py > $O/r05_vendor_yardstick.txt 2>&1; tail -5 $O/r05_vendor_yardstick.txt
echo "== plain-weight shapes"
for mode in 0 4 2; do
  echo "== plain, M3R_G256P=$mode"
  M3R_G256P=$mode M3R_GEMM256=2 PLAIN16=1 EXTRA=1 timeout 300 python scripts/exp_gemm256.py 2>&1 | grep -v amdgpu.ids
done > $O/r05_g256p_plain.txt 2>&1
grep -E "RACE|mode|== " $O/r05_g256p_plain.txt
echo "== split-weight shapes"
for mode in 0 2; do
  echo "== split, M3R_G256P_SPLIT=$mode"
  M3R_G256P_SPLIT=$mode SPLIT=1 EXTRA=1 timeout 300 python scripts/exp_gemm256.py 2>&1 | grep -v amdgpu.ids
done > $O/r05_g256p_split.txt 2>&1
grep -E "RACE|mode|== " $O/r05_g256p_split.txt
echo "== bit-identity tests with the new kernels on"
M3R_G256P=4 M3R_G256P_SPLIT=2 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm256_tiles or gemm_store_gelu or gemm_split" 2>&1 | tail -5
M3R_G256P=2 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm256_tiles" 2>&1 | tail -3
echo "== step, S = 20"
step() {   # $1 = tag; environment = the variant
  timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --step-only > $O/r05_step_$1.json 2> $O/r05_step_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r05_step_{tag}.json").read().strip().splitlines()[-1])
    print(tag, "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "gemm frac", d.get("roofline", {}).get("frac"))
except Exception as e:
    print(tag, "failed", e)
PY
}
step base
M3R_G256P=4 step p4
M3R_G256P=2 step p2
M3R_G256P=4 M3R_G256P_SPLIT=2 step p4s2
M3R_ENC_CHUNK_ROWS=16384 step chunk16k
M3R_ENC_CHUNK_ROWS=8192 step chunk8k
echo "== done"
