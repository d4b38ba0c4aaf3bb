#!/bin/bash
# r04 GPU call 1: MX-fp4 MFMA probe, full suite at the round's first state (ADVICE fixes, LINEAR, single-collective sharding, r04 tests),
# CPU-baseline thread sweep, first pass of the closing profile script (kernel trace + PMC incl. TCC hit rate and MFMA busy)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== mx4 probe"; timeout 120 scripts/probes/build/mx4_probe > $O/r04_mx4_probe.txt 2>&1; echo "rc=$?"; cat $O/r04_mx4_probe.txt | cut -c1-400
echo "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 -x > $O/r04_tests1.log 2>&1; echo "tests rc=$?"; tail -15 $O/r04_tests1.log | cut -c1-300
echo "== cpu baseline thread sweep (2-view scene)"
for T in 16 32 64 128; do timeout 200 python oracle/cpu_baseline.py --views 2 --threads $T --out /tmp/cpu_$T.npz 2>/dev/null | tail -1 | cut -c1-200; done
M3R_COMMIT=r04-call1 bash scripts/gpu_profile_r04.sh 2>&1 | tail -60
echo "== done"
