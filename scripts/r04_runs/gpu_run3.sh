#!/bin/bash
# r04 GPU call 3: 128-deep K-tiles in the M = 768 kernels: op tests, microbench per depth, S = 1 scene per depth
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== gemm op tests"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_edge_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k "gemm or lnfold or ln_fold or fold" > $O/r04_tests3.log 2>&1; echo "tests rc=$?"; tail -6 $O/r04_tests3.log | cut -c1-300
for B in 0 1 2; do echo "== M3R_BK128=$B"; M3R_BK128=$B timeout 300 python scripts/bench_gemm_m768.py 2>&1 | grep -v amdgpu.ids; done > $O/r04_bk128_m768.txt 2>&1; cat $O/r04_bk128_m768.txt
for B in 0 1 2; do echo "== S=1 step, M3R_BK128=$B"; M3R_BK128=$B timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --scenes 1 --step-only > $O/r04_s1_bk$B.json 2> $O/r04_s1_bk$B.err
python - $O/r04_s1_bk$B.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms")}, d["kernel_classes"]["gemm64"])
P
done
echo "== model tests (S=1 parity with the new depth)"; timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 -k "fixture or oracle" > $O/r04_tests3b.log 2>&1; echo "tests rc=$?"; tail -4 $O/r04_tests3b.log | cut -c1-300
echo "== done"
