#!/bin/bash
# r03 GPU call 4: full suite after the clean-up (split-K route, experiment switches, overlap option removed), the new bench line
# (S scenes in flight per rank as the step), 2-rank gloo dry run of the N > 1 paths, rocprofv3 kernel stats + PMC traffic of the bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== full GPU suite"; timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r03_tests4.log 2>&1; echo "tests rc=$?"; tail -12 $O/r03_tests4.log | cut -c1-300
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench default"; SECONDS=0; timeout 1200 python bench.py --gpus 1 --steps 5 --warmup 2 > $O/r03_bench4.log 2> $O/r03_bench4.err; echo "bench rc=$? wall=${SECONDS}s"; tail -3 $O/r03_bench4.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r03_bench4.log").read().strip().splitlines()[-1])
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
echo "== 2-rank gloo dry run of bench.py --gpus 2 (both ranks on this one GPU)"
M3R_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --scenes 4 --stream-frames 40 > $O/r03_bench_2rank_gloo_dryrun.log 2>&1; echo "rc=$?"
tail -1 $O/r03_bench_2rank_gloo_dryrun.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['parallelism']); [print(c['config'][:70], c['value']) for c in d['configs']]
except Exception as e: print('parse failed', e)"
rm -rf $O/prof; bash scripts/gpu_prof.sh > /dev/null 2>&1
python scripts/prof_summary.py $(ls $O/prof/*.db 2>/dev/null | tail -1) $O/r03_bench_kernel_stats.txt | head -24
bash scripts/gpu_pmc.sh > $O/pmc.log 2>&1; tail -4 $O/pmc.log
python scripts/pmc_summary.py $O/r03_pmc_traffic.json | head -6
find $O -name "*.db" -size +20M -delete; find $O -name "*.csv" -size +8M -delete
echo "== done"
