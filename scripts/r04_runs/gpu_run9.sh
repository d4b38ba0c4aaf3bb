#!/bin/bash
# r04 GPU call 9: the driver's own sequence at the final commit -- pytest -m gpu, smoke(), python bench.py
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== full GPU suite"; timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/r04_tests9.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" $O/r04_tests9.log | tail -2
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench default"; SECONDS=0; timeout 1500 python bench.py > $O/r04_bench9.json 2> $O/r04_bench9.err; echo "bench rc=$? wall=${SECONDS}s"
python - <<'P'
import json
d = json.loads(open("gpurun_out/r04_bench9.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "value_single_scene", "ms_per_step", "stages_ms", "alt", "scenes_in_flight_sweep", "end_to_end_mfma_frac")})
r = dict(d["roofline"]); r.pop("per_symbol"); print("roofline", {k: r[k] for k in ("achieved", "frac", "traffic", "ms_per_step")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
P
echo "== done"
