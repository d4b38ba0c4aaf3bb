#!/bin/bash
# r06 call 17: the driver's round-end sequence at HEAD: pytest -m gpu, smoke, default bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
timeout 2700 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r06_c17_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r06_c17_tests.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_c17_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06_c17_smoke.log
SECONDS=0
timeout 1800 python bench.py > gpurun_out/r06_c17_bench.json 2> gpurun_out/r06_c17_bench.err; echo "bench rc=$? in ${SECONDS}s"
python -c "
import json; d=json.loads(open('gpurun_out/r06_c17_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'single',d['value_single_scene'],'frac',d['roofline']['frac'],'parity',d['config']['parity']['rel_inf_worst_view'],'cpu',d['cpu_baseline']['value'],'torch',d['torch_rocm_baseline']['value'])"
