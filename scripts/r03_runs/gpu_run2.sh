#!/bin/bash
# r03 GPU call 2: the fp16wa precision mode (plain Mlp weights), full suite with attn3 as the 16-bit kernel again, bench + kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== full GPU suite"; timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r03_tests2.log 2>&1; echo "tests rc=$?"; tail -15 $O/r03_tests2.log | cut -c1-400
echo "== bench default"; SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 8 --warmup 2 > $O/r03_bench2.log 2> $O/r03_bench2.err; echo "bench rc=$? wall=${SECONDS}s"; tail -3 $O/r03_bench2.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r03_bench2.log").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "stages_ms", "alt", "dtype")})
    print("classes", d["kernel_classes"])
    print("flight", d["scenes_in_flight"])
    for k, v in (d["parity_vs_cpu_oracle"] or {}).items():
        print("parity", k, v if not isinstance(v, dict) else {a: (round(b, 6) if isinstance(b, float) else b) for a, b in v.items()})
    print([(c["config"][:34], c.get("value") or [(m["dtype"][:12], m["value"], m.get("render_rel_inf_vs_16bit_path")) for m in c["modes"]]) for c in d["configs"]])
except Exception as e:
    print("bench parse failed", e)
P
echo "== bench scenes=8"; timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --scenes 8 --no-cpu-baseline --no-configs --no-alt 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['scenes_in_flight'])"
echo "== bench fp16w2 scenes=4"; timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --precision fp16w2 --no-cpu-baseline --no-configs --no-alt 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stages_ms'], d['scenes_in_flight']['value'])"
rm -rf $O/prof; bash scripts/gpu_prof.sh > /dev/null 2>&1
python scripts/prof_summary.py $(ls $O/prof/*.db 2>/dev/null | tail -1) $O/r03_bench_kernel_stats_run2.txt | head -40
find $O -name "*.db" -size +20M -delete; find $O -name "*.csv" -size +8M -delete
echo "== done"
