#!/bin/bash
# r05 GPU call 2: gemm256p one-barrier form (M3R_G256P=1) against the two-barrier form (=2) and gemm256k (=0); which kernels the vendor library runs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
for mode in 0 2 1; do
  echo "== plain, M3R_G256P=$mode"
  M3R_G256P=$mode M3R_GEMM256=2 PLAIN16=1 EXTRA=1 timeout 300 python scripts/exp_gemm256.py 2>&1 | grep -v amdgpu.ids
done > $O/r05_g256p1_plain.txt 2>&1
grep -E "RACE|mode|== " $O/r05_g256p1_plain.txt
for mode in 0 2 1; do
  echo "== split, M3R_G256P_SPLIT=$mode"
  M3R_G256P_SPLIT=$mode SPLIT=1 EXTRA=1 timeout 300 python scripts/exp_gemm256.py 2>&1 | grep -v amdgpu.ids
done > $O/r05_g256p1_split.txt 2>&1
grep -E "RACE|mode|== " $O/r05_g256p1_split.txt
echo "== bit-identity tests"
M3R_G256P=1 M3R_G256P_SPLIT=1 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm256_tiles or gemm_store_gelu or gemm_split" 2>&1 | tail -3
echo "== step, S = 20"
step() {
  timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --step-only > $O/r05_step_$1.json 2> $O/r05_step_$1.err
  python - "$1" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r05_step_{tag}.json").read().strip().splitlines()[-1])
    print(tag, "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "stages", {k: round(v, 1) for k, v in d.get("stages_ms", {}).items()})
except Exception as e:
    print(tag, "failed", e)
PY
}
step base2
M3R_G256P=1 step p1
M3R_G256P=1 M3R_G256P_SPLIT=1 step p1s1
M3R_G256P=1 M3R_G256P_SPLIT=2 step p1s2
echo "== vendor kernel names"
cd /tmp
ITERS=3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_vendor -o vendor -- python $GRAFT_REPO_ROOT/scripts/yardstick_vendor.py > /tmp/vendor_prof.log 2>&1
cd "$GRAFT_REPO_ROOT"
python scripts/prof_summary.py $(ls /tmp/prof_vendor/*.db /tmp/prof_vendor/*/*.db 2>/dev/null | tail -1) $O/r05_vendor_kernel_stats.txt | head -5
python - <<'PY' > $O/r05_vendor_kernel_names.txt 2>&1
import glob, sqlite3
db = sorted(glob.glob("/tmp/prof_vendor/*.db") + glob.glob("/tmp/prof_vendor/*/*.db"))[-1]
c = sqlite3.connect(db)
for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    if pct >= 0.05:
        print(f"{calls:6d} calls  avg {avg:9.2f} us  {pct:6.2f} %  {name}")
PY
head -c 6000 $O/r05_vendor_kernel_names.txt
echo "== done"
