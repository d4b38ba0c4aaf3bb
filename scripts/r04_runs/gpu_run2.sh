#!/bin/bash
# r04 GPU call 2: the r04 tests, the S = 1 scene per kernel symbol, scenes-in-flight / encoder-chunk sweeps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== r04 tests"; timeout 900 python -m pytest tests/test_zz_r04_gpu.py tests/test_cam_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 > $O/r04_tests2.log 2>&1; echo "tests rc=$?"; tail -15 $O/r04_tests2.log | cut -c1-300
show() { python - "$1" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms")}, d["config"]["scenes_in_flight_per_gpu"])
print("classes", d["kernel_classes"])
for r in d["roofline"].get("per_symbol", [])[:40]: print("  ", r["kernel"], r["launches"], r["ms"], r["avg_launch_us"], r["achieved_tflops"])
for r in d["roofline_attention"].get("per_symbol", []): print("  ", r["kernel"], r["launches"], r["ms"], r["avg_launch_us"], r["achieved_tflops"])
P
}
echo "== S=1 step per symbol"; timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --scenes 1 --step-only > $O/r04_s1.json 2> $O/r04_s1.err; show $O/r04_s1.json
for S in 24 28; do echo "== S=$S"; timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --scenes $S --step-only > $O/r04_s$S.json 2> $O/r04_s$S.err; show $O/r04_s$S.json | head -3; done
for R in 65536 163840 327680; do echo "== S=20 enc chunk rows $R"; M3R_ENC_CHUNK_ROWS=$R timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --scenes 20 --step-only > $O/r04_chunk$R.json 2> $O/r04_chunk$R.err; show $O/r04_chunk$R.json | head -2; done
echo "== done"
