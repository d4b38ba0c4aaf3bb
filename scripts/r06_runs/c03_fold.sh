#!/bin/bash
# r06 call 3: fold256 op + model tests, persistent-loop test (take_ix fix), then a bench A/B of LNFOLD256
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
echo "== fold256 + persist tests"
timeout 1500 python -m pytest tests/test_zz_r06_gpu.py tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 -x -k "fold256 or persistent or ln_fold" > gpurun_out/r06_c03_tests.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/r06_c03_tests.log
for F in 1 0 1 0; do
  M3R_LNFOLD256=$F timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --step-only > gpurun_out/r06_c03_bench_fold$F.json 2> gpurun_out/r06_c03_bench_fold$F.err; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r06_c03_bench_fold$F.json').read().strip().splitlines()[-1])
    print('LNFOLD256=$F value',d['value'],'ms',d['ms_per_step'],'stages',d['stages_ms'],'gemm',d['roofline']['achieved'],d['roofline']['ms_per_step'])
    print('   classes',{k:(v['ms'],v['calls']) for k,v in d['kernel_classes'].items()})
    for r in d['roofline']['per_symbol']: print('     ',r['kernel'],r['launches'],r['avg_launch_us'],r['achieved_tflops'])
    print('   parity',d['config'].get('parity',{}).get('rel_inf_worst_view'), d.get('parity_vs_cpu_oracle',{}).get('step_scene0_of_28'))
except Exception as e:
    print('LNFOLD256=$F failed',e); print(open('gpurun_out/r06_c03_bench_fold$F.err').read()[-1500:])
PY
done
echo "== done"
