#!/bin/bash
# r06 call 7: context-parallel tests (both partial formats), closing evidence (kernel stats + PMC passes at S = 28), seam trace of the one-scene pass, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
echo "== r06 tests"
timeout 1500 python -m pytest tests/test_zz_r06_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 > gpurun_out/r06_c07_tests.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r06_c07_tests.log
echo "== closing evidence"
bash scripts/gpu_profile_r06.sh 2>&1 | tail -60
echo "== seam trace of the one-scene pass"
rm -rf gpurun_out/prof_seam
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_seam -o run -- python bench.py --gpus 1 --scenes 1 --steps 3 --warmup 1 --step-only > gpurun_out/r06_seam_line.json 2> gpurun_out/r06_seam.err; echo "rc=$?"
python scripts/seam_trace.py gpurun_out/prof_seam | tee gpurun_out/r06_single_scene_seams.txt
find gpurun_out/prof_seam -name "*.csv" -size +8M -delete; find gpurun_out/prof_seam -name "*.db" -delete
echo "== default bench line"
timeout 1800 python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err; echo "rc=$?"; tail -c 600 gpurun_out/r06_bench_line.json
echo "== done"
