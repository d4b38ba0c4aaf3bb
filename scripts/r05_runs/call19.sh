#!/bin/bash
# r05 GPU call 19: closing evidence at the final kernels: GPU test suite, the default bench line, rocprofv3 kernel trace + PMC passes of the benched step (S = 28)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== gpu tests"
rm -f $O/test_metrics.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout=900 > $O/r05_gputests.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/r05_gputests.log | tail -3
cp $O/test_metrics.jsonl $O/r05_test_metrics.jsonl 2>/dev/null
echo "== bench (the driver's command)"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_line.json 2> $O/r05_bench.err; echo "rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1])
    print("value", d["value"], "single", d["value_single_scene"], "ms", d["ms_per_step"], "gemm", d["roofline"]["achieved"], d["roofline"]["frac"], "attn", d["roofline_attention"]["achieved"], "e2e", d["end_to_end_mfma_frac"])
    print("stages", d["stages_ms"], "sweep", d["scenes_in_flight_sweep"], "alt", d["alt"])
    print("single", d["single_scene"]["stages_ms"], d["single_scene"]["value"])
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["seconds"])
    p = d["parity_vs_cpu_oracle"]
    print("parity", {k: (round(v["render_per_view_max"], 6), round(v["update_per_view_max"], 6)) for k, v in p.items() if isinstance(v, dict) and "render_per_view_max" in v})
    for c in d["configs"]:
        sf = c.get("scenes_in_flight") if isinstance(c.get("scenes_in_flight"), dict) else {}
        print(" ", c["config"][:60], c.get("value"), sf.get("value"), [m.get("value") for m in c.get("modes", [])], [m.get("value") for m in sf.get("modes", [])])
except Exception as e:
    print("bench parse failed", e)
PY
echo "== profile"
M3R_COMMIT=$(cat .commit) bash scripts/gpu_profile_r05.sh 2>&1 | tail -40
echo "== done"
