#!/bin/bash
# round-2 GPU call 4: edge tests, attention XCD-balanced block map (microbench + scene), kernel-by-kernel trace of one update call
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "== edge + attention tests"
timeout 900 python -m pytest tests/test_edge_gpu.py tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "edge or attention or massive or rope" > gpurun_out/edge_tests.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/edge_tests.log
grep massive gpurun_out/test_metrics.jsonl | tail -4
echo "== attention microbench"
timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_map.txt
M3R_ATTN_OLDMAP=1 timeout 300 python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/attn_map.txt
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-alt > gpurun_out/bench4.log 2>&1; echo "rc=$?"; python - <<'P'
import json
d = json.loads(open("gpurun_out/bench4.log").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "kernel_classes", "stages_ms")})
P
echo "== trace"
rm -rf gpurun_out/trace; mkdir -p gpurun_out/trace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-configs --no-alt > gpurun_out/trace.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1); echo "$f"
[ -n "$f" ] && python scripts/trace_seq.py "$f" 30 | tee gpurun_out/trace_seq.txt | tail -40
[ -n "$f" ] && head -2 "$f"
find gpurun_out/trace -name "*.csv" -size +30M -delete
