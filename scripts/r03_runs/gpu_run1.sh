#!/bin/bash
# r03 GPU call 1: the 32x32-tile attention kernel (attn4, 16-bit and fp8 Q/K) at operator level against attn3, the attention
# microbenchmark for both, the full GPU suite (batched scenes, ABI 5, fp8 layout), then the bench with S scenes in flight.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out
echo "== ops attention, attn4"; M3R_ATTN=4 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 -k "attention" > $O/r03_ops_attn4.log 2>&1; rc4=$?; tail -6 $O/r03_ops_attn4.log
echo "== ops attention, attn3"; M3R_ATTN=2 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider --timeout=300 -k "attention" > $O/r03_ops_attn3.log 2>&1; tail -3 $O/r03_ops_attn3.log
echo "== bench_attn"
{ M3R_ATTN=2 timeout 300 python scripts/bench_attn.py; M3R_ATTN=4 timeout 300 python scripts/bench_attn.py; M3R_ATTN=4 M3R_ATTN_QF=2 timeout 300 python scripts/bench_attn.py; M3R_ATTN=4 FP8=1 timeout 300 python scripts/bench_attn.py; } > $O/r03_attn_ab.txt 2>&1; cat $O/r03_attn_ab.txt | grep -v amdgpu.ids
ATT=4; if [ $rc4 -ne 0 ]; then ATT=2; echo "attn4 failed at operator level: the rest runs with M3R_ATTN=2"; fi
export M3R_ATTN=$ATT
echo "== full GPU suite (M3R_ATTN=$ATT)"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 > $O/r03_tests1.log 2>&1; echo "tests rc=$?"; tail -30 $O/r03_tests1.log
echo "== bench default"; SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 8 --warmup 2 > $O/r03_bench1.log 2> $O/r03_bench1.err; echo "bench rc=$? wall=${SECONDS}s"; tail -3 $O/r03_bench1.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r03_bench1.log").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "stages_ms", "alt")})
    print("classes", d["kernel_classes"])
    print("flight", d["scenes_in_flight"])
    print("roofline", d["roofline"])
    for k, v in (d["parity_vs_cpu_oracle"] or {}).items():
        print("parity", k, v if not isinstance(v, dict) else {a: (round(b, 6) if isinstance(b, float) else b) for a, b in v.items()})
    print([(c["config"][:34], c.get("value") or [(m["dtype"][:12], m["value"], m.get("render_rel_inf_vs_16bit_path")) for m in c["modes"]]) for c in d["configs"]])
except Exception as e:
    print("bench parse failed", e)
P
for S in 2 8; do
  echo "== bench scenes=$S"; timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --scenes $S --no-cpu-baseline --no-configs --no-alt 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['scenes_in_flight'])"
done
if [ "$ATT" = "4" ]; then
  echo "== bench with attn3"; M3R_ATTN=2 timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --scenes 4 --no-cpu-baseline --no-configs --no-alt 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stages_ms'], d['kernel_classes'], d['scenes_in_flight']['value'])"
fi
echo "== done"
