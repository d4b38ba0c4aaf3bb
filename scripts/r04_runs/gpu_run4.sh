#!/bin/bash
# r04 GPU call 4: full suite + smoke + default bench (driver's command) + closing profile (kernel trace + PMC) at the round's kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== full GPU suite"; timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r04_tests4.log 2>&1; echo "tests rc=$?"; tail -8 $O/r04_tests4.log | cut -c1-300
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench default"; SECONDS=0; timeout 1500 python bench.py > $O/r04_bench4.json 2> $O/r04_bench4.err; echo "bench rc=$? wall=${SECONDS}s"; tail -3 $O/r04_bench4.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r04_bench4.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "value_single_scene", "ms_per_step", "stages_ms", "alt", "end_to_end_mfma_frac")})
    print("classes", d["kernel_classes"])
    print("single", d["single_scene"])
    r = dict(d["roofline"]); ps = r.pop("per_symbol"); print("roofline", r)
    for x in ps[:16]: print("   ", x)
    r = dict(d["roofline_attention"]); ps = r.pop("per_symbol"); print("roofline_attention", r)
    for k, v in (d["parity_vs_cpu_oracle"] or {}).items():
        print("parity", k, v if not isinstance(v, dict) else {a: (round(b, 6) if isinstance(b, float) else b) for a, b in v.items()})
    print("cpu", d["cpu_baseline"])
    for c in d["configs"]:
        print(c["config"][:70], c.get("value") or [(m["dtype"][:24], m["value"], m.get("render_rel_inf_vs_16bit_path")) for m in c["modes"]], c.get("scenes_in_flight"))
except Exception as e:
    print("bench parse failed", repr(e))
P
M3R_COMMIT=$(cat .commit 2>/dev/null || echo r04) bash scripts/gpu_profile_r04.sh 2>&1 | tail -45
echo "== done"
