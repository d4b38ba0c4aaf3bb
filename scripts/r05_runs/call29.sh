#!/bin/bash
# r05 GPU call 29: kernel trace of the one-scene-at-a-time pass (S = 1) at the final kernels: where the 58-60 ms go
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
rm -rf $O/prof_s1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_s1 -o s1 -- python bench.py --gpus 1 --steps 10 --warmup 3 --scenes 1 --step-only > $O/r05_s1_line.json 2> $O/r05_s1.err
echo "rc=$?"
python scripts/prof_summary.py $(ls $O/prof_s1/*.db $O/prof_s1/*/*.db 2>/dev/null | tail -1) $O/r05_single_scene_kernel_stats.txt | head -30
find $O/prof_s1 -name "*.db" -size +20M -delete; find $O/prof_s1 -name "*.csv" -size +8M -delete
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_s1_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["stages_ms"], {k: (round(v["ms"], 2), v["calls"], v["tflops"]) for k, v in d["kernel_classes"].items()})
PY
echo "== done"
