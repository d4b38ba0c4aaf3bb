#!/bin/bash
# r03 GPU call 17: full suite + bench + rocprofv3 kernel stats + PMC traffic at the final kernels of the round
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== full GPU suite"; timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r03_tests17.log 2>&1; echo "tests rc=$?"; tail -12 $O/r03_tests17.log | cut -c1-300
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench default"; SECONDS=0; timeout 1200 python bench.py --gpus 1 --steps 5 --warmup 2 > $O/r03_bench17.log 2> $O/r03_bench17.err; echo "bench rc=$? wall=${SECONDS}s"; tail -3 $O/r03_bench17.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r03_bench17.log").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "stages_ms", "alt", "end_to_end_mfma_frac")})
    print("config", d["config"])
    print("classes", d["kernel_classes"])
    print("single", d["single_scene"])
    print("roofline", d["roofline"])
    for k, v in (d["parity_vs_cpu_oracle"] or {}).items():
        print("parity", k, v if not isinstance(v, dict) else {a: (round(b, 6) if isinstance(b, float) else b) for a, b in v.items()})
    print("cpu", d["cpu_baseline"])
    for c in d["configs"]:
        print(c["config"][:60], c.get("value") or [(m["dtype"][:16], m["value"], m.get("render_rel_inf_vs_16bit_path")) for m in c["modes"]], c.get("scenes_in_flight"))
except Exception as e:
    print("bench parse failed", e)
P
rm -rf $O/prof; bash scripts/gpu_prof.sh > /dev/null 2>&1
python scripts/prof_summary.py $(ls $O/prof/*.db 2>/dev/null | tail -1) $O/r03_bench_kernel_stats.txt | head -24
bash scripts/gpu_pmc.sh > $O/pmc.log 2>&1; tail -4 $O/pmc.log
python scripts/pmc_summary.py $O/r03_pmc_traffic.json | head -6
find $O -name "*.db" -size +20M -delete; find $O -name "*.csv" -size +8M -delete
echo "== done"
