#!/bin/bash
# round-2 closing run: full GPU suite + smoke, rocprofv3 kernel stats and PMC traffic of the bench command, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > gpurun_out/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
rm -rf gpurun_out/prof gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
bash scripts/gpu_prof.sh > /dev/null 2>&1
python scripts/prof_summary.py $(ls gpurun_out/prof/*.db 2>/dev/null | tail -1) gpurun_out/r02_bench_kernel_stats.txt | head -16
bash scripts/gpu_pmc.sh > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/r02_pmc_traffic.json | head -4
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench rc=$? wall=${SECONDS}s"
tail -1 gpurun_out/bench_default.log > gpurun_out/r02_bench_line.json
python - <<'P'
import json
d = json.loads(open("gpurun_out/r02_bench_line.json").read())
print({k: d[k] for k in ("value", "ms_per_step", "kernel_classes", "stages_ms", "alt")})
print(d["roofline"]); print(d["parity_vs_cpu_oracle"]["fp16w2"]); print(d["cpu_baseline"]["value"], d["cpu_baseline"]["seconds"])
print([(c["config"][:34], c.get("value") or c["modes"][0]["value"]) for c in d["configs"]])
P
find gpurun_out -name "*.db" -size +20M -delete; find gpurun_out -name "*.csv" -size +8M -delete
