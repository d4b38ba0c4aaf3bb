#!/bin/bash
# r05 GPU call 28: scenes-in-flight sweep at the final r05 kernels (the sawtooth of the round quantisation; r04: profiles/r04_scenes_sweep.txt)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
: > $O/r05_scenes_sweep.txt
for S in 4 8 12 16 20 24 28 32 40 56; do
  timeout 400 python bench.py --gpus 1 --steps 2 --warmup 1 --scenes $S --step-only > $O/r05_sweep_$S.json 2> $O/r05_sweep_$S.err
  python - "$S" <<'PY' | tee -a gpurun_out/r05_scenes_sweep.txt
import json, sys
S = int(sys.argv[1])
try:
    d = json.loads(open(f"gpurun_out/r05_sweep_{S}.json").read().strip().splitlines()[-1])
    st = d["stages_ms"]
    print(f"S={S:3d}  {d['value']:7.2f} views/s  per scene: encode {st['encode']/S:6.2f} update {st['update']/S:6.2f} render {st['render']/S:6.2f} ms  gemm {d['roofline']['achieved']:.1f} TF/s  attn {d['roofline_attention']['achieved']:.1f} TF/s")
except Exception as e:
    print(f"S={S:3d} failed {e}")
PY
done
echo "== done"
