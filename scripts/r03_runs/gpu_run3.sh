#!/bin/bash
# r03 GPU call 3: gemm256s (streamed fragments, all eight waves multiplying) against gemm256 (two staggered groups): operator tests,
# bit-identity, time on the chip-filling shapes, bench A/B; the C-driven ABI test.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== ops gemm with gemm256s"; M3R_G256S=2 timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 -k "gemm" > $O/r03_ops_g256s.log 2>&1; echo "rc=$?"; tail -4 $O/r03_ops_g256s.log
{ for g in 0 1; do PLAIN16=1 M3R_G256S=$g timeout 300 python scripts/exp_gemm256.py; done; for g in 0 2; do SPLIT=1 M3R_G256S=$g timeout 300 python scripts/exp_gemm256.py; done; } > $O/r03_gemm256s_ab.txt 2>&1; grep -v amdgpu.ids $O/r03_gemm256s_ab.txt
echo "== abi test"; timeout 600 python -m pytest tests/test_zz_abi_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -5
for g in 0 1 2; do
  echo "== bench G256S=$g"; M3R_G256S=$g timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --scenes 8 --no-cpu-baseline --no-configs --no-alt 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stages_ms'], d['kernel_classes']['gemm128'], d['scenes_in_flight']['value'], d['scenes_in_flight']['kernel_classes']['gemm128'])"
done
echo "== done"
