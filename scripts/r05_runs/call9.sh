#!/bin/bash
# r05 GPU call 9: the GPU test suite, the full default bench line, the 2-rank gloo dry run of `bench.py --gpus 2`, and one PMC pass at 28 scenes (does rocprofv3 survive it?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== gpu tests"
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout=900 > $O/r05_gputests.log 2>&1; echo "rc=$?"; tail -5 $O/r05_gputests.log
cp $O/test_metrics.jsonl $O/r05_test_metrics.jsonl 2>/dev/null
echo "== bench (default command)"
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/r05_bench_line.json 2> $O/r05_bench.err; echo "rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench_line.json").read().strip().splitlines()[-1])
    print("value", d["value"], "single", d["value_single_scene"], "ms", d["ms_per_step"], "gemm", d["roofline"]["achieved"], d["roofline"]["frac"], "attn", d["roofline_attention"]["achieved"])
    print("stages", d["stages_ms"], "sweep", d["scenes_in_flight_sweep"], "alt", d["alt"])
    print("single stages", d["single_scene"]["stages_ms"])
    print("cpu", d["cpu_baseline"])
    p = d["parity_vs_cpu_oracle"]
    print("parity", {k: (round(v["render_per_view_max"], 6), round(v["update_per_view_max"], 6)) for k, v in p.items() if isinstance(v, dict) and "render_per_view_max" in v})
    for c in d["configs"]:
        print(" ", c["config"][:60], c.get("value"), c.get("scenes_in_flight", {}).get("value") if isinstance(c.get("scenes_in_flight"), dict) else "", [m.get("value") for m in c.get("modes", [])])
except Exception as e:
    print("bench parse failed", e)
PY
echo "== 2-rank gloo dry run"
M3R_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --scenes 4 --stream-frames 40 > $O/r05_bench_2rank_gloo_dryrun.log 2>&1; echo "rc=$?"
tail -1 $O/r05_bench_2rank_gloo_dryrun.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'rccl', d['rccl'], 'view_sharded', d['view_sharded'])"
echo "== one PMC pass at 28 scenes"
cd /tmp
timeout 420 rocprofv3 --kernel-trace --kernel-include-regex "gemm|attn" --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_try -o run -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 1 --warmup 1 --scenes 28 --step-only > /tmp/pmc_try.log 2>&1; echo "pmc rc=$?"
ls -la /tmp/pmc_try/*/ 2>/dev/null | head; find /tmp/pmc_try -name "*counter_collection.csv" | head -2 | xargs -r wc -l
echo "== done"
