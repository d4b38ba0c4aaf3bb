#!/bin/bash
# r06 call 1: full -m gpu suite at the r06 tree (persistent GEMM loop on by default), persistent-loop A/B on the step's shapes, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout=900 > gpurun_out/r06_c01_tests.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r06_c01_tests.log
echo "== persist A/B"
timeout 600 python scripts/r06_gemm_persist_ab.py > gpurun_out/r06_c01_persist_ab.txt 2>&1; echo "rc=$?"; cat gpurun_out/r06_c01_persist_ab.txt
echo "== bench (PERSIST=1 default)"
timeout 1500 python bench.py --gpus 1 --steps 10 --warmup 3 > gpurun_out/r06_c01_bench.json 2> gpurun_out/r06_c01_bench.err; echo "rc=$?"; tail -3 gpurun_out/r06_c01_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_c01_bench.json').read().strip().splitlines()[-1])
print("value",d["value"],"single",d["value_single_scene"],"r04def",d.get("value_r04_definition"),"ms",d["ms_per_step"],"stages",d["stages_ms"])
print("roofline",d["roofline"]["achieved"],d["roofline"]["frac"],"attn",d["roofline_attention"]["achieved"])
for r in d["roofline"]["per_symbol"]: print("  ",r["kernel"],r["launches"],r["avg_launch_us"],r["achieved_tflops"])
for r in d["roofline_attention"]["per_symbol"]: print("  ",r["kernel"],r["launches"],r["avg_launch_us"],r["achieved_tflops"])
print("classes",d["kernel_classes"])
print("config.parity",d["config"].get("parity"))
print("cpu",d["cpu_baseline"]); print("torch_rocm",d.get("torch_rocm_baseline"))
p=d["parity_vs_cpu_oracle"]; print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if 'view' in kk}) for k,v in p.items()})
PY
echo "== step-only bench with PERSIST=0 (same box A/B)"
M3R_PERSIST=0 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --step-only > gpurun_out/r06_c01_bench_persist0.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r06_c01_bench_persist0.json').read().strip().splitlines()[-1]);print('PERSIST=0 value',d['value'],'gemm',d['roofline']['achieved'],d['roofline']['ms_per_step'])"
M3R_PERSIST=1 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --step-only > gpurun_out/r06_c01_bench_persist1.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r06_c01_bench_persist1.json').read().strip().splitlines()[-1]);print('PERSIST=1 value',d['value'],'gemm',d['roofline']['achieved'],d['roofline']['ms_per_step'])"
echo "== done"
