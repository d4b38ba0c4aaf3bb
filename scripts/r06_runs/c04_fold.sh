#!/bin/bash
# r06 call 4: fold256 with asm prologue loads: op/model tests, step A/B, then the fixture parity tests with the fold on
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
echo "== fold256 tests"
timeout 1500 python -m pytest tests/test_zz_r06_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 -k "fold256" > gpurun_out/r06_c04_tests.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r06_c04_tests.log
for F in 1 0 1 0; do
  M3R_LNFOLD256=$F timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --step-only > gpurun_out/r06_c04_bench_fold$F.json 2> gpurun_out/r06_c04_bench_fold$F.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r06_c04_bench_fold$F.json').read().strip().splitlines()[-1])
    print('LNFOLD256=$F value',d['value'],'ms',d['ms_per_step'],'stages',d['stages_ms'],'gemm',d['roofline']['achieved'],d['roofline']['ms_per_step'])
    print('   classes',{k:(v['ms'],v['calls']) for k,v in d['kernel_classes'].items()})
    for r in d['roofline']['per_symbol'][:6]: print('     ',r['kernel'],r['launches'],r['avg_launch_us'],r['achieved_tflops'])
except Exception as e:
    print('LNFOLD256=$F failed',e); print(open('gpurun_out/r06_c04_bench_fold$F.err').read()[-1500:])
PY
done
echo "== fixture parity with the fold on"
timeout 2400 python -m pytest tests/test_zz_r04_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 > gpurun_out/r06_c04_parity.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r06_c04_parity.log
grep -h "benched_configuration\|full_depth\|fold256" gpurun_out/test_metrics.jsonl | tail -12
echo "== done"
