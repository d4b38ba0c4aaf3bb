#!/bin/bash
# r04 GPU call 12: single-transcendental erf GELU epilogue (erfc(z) ~ 2^(-z P4(z))): op tests, fixture parity, fc1 shapes, S = 20 step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -2
echo "== plain fc1 shapes"; PLAIN16=1 timeout 300 python scripts/exp_gemm256.py 2>&1 | grep -v amdgpu.ids | grep "fc1\|qkv\|big shapes" | tee $O/r04_gelu_erfc.txt
echo "== split fc1 shapes"; SPLIT=1 timeout 300 python scripts/exp_gemm256.py 2>&1 | grep -v amdgpu.ids | grep "fc1\|big shapes" | tee -a $O/r04_gelu_erfc.txt
echo "== fixture + benched-configuration parity"; timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_zz_r04_gpu.py tests/test_zz_batch_gpu.py -m gpu -q -p no:cacheprovider -k "fixture or benched or in_flight" 2>&1 | tail -2
grep "benched_configuration" $O/../gpurun_out/test_metrics.jsonl | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('benched worst update', max(d['update_per_view']), 'render', max(d['render_per_view']))"
echo "== S=20 step"; timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --scenes 20 --step-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stages_ms'], d['kernel_classes']['gemm128']); print([ (r['kernel'], r['avg_launch_us'], r['achieved_tflops']) for r in d['roofline']['per_symbol'] if '/e1/' in r['kernel']])" | tee -a $O/r04_gelu_erfc.txt
echo "== done"
