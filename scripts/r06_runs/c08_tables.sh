#!/bin/bash
# r06 call 8: view tables by copy kernel instead of hipMemcpyAsync: seam trace + one-scene bench + tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "== tests"
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_zz_batch_gpu.py tests/test_zz_drivers_gpu.py tests/test_edge_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 -x > gpurun_out/r06_c08_tests.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r06_c08_tests.log
echo "== seam trace"
rm -rf gpurun_out/prof_seam
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_seam -o run -- python bench.py --gpus 1 --scenes 1 --steps 3 --warmup 1 --step-only > gpurun_out/r06_seam_line.json 2> gpurun_out/r06_seam.err; echo "rc=$?"
python scripts/seam_trace.py gpurun_out/prof_seam | head -6 | tee gpurun_out/r06_single_scene_seams_after.txt
find gpurun_out/prof_seam -name "*.db" -delete
echo "== one scene at a time, untraced"
for i in 1 2; do timeout 600 python bench.py --gpus 1 --scenes 1 --steps 20 --warmup 3 --step-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=1 value',d['value'],'ms',d['ms_per_step'],d['stages_ms'])"; done
echo "== S=28 step"
timeout 600 python bench.py --gpus 1 --steps 8 --warmup 3 --step-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=28 value',d['value'],'ms',d['ms_per_step'],d['stages_ms'])"
echo "== done"
