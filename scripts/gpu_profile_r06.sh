#!/bin/bash
# r06 closing evidence (r06: the same passes at the r06 closing commit): everything the bench line's roofline numbers can be recomputed from, taken at ONE commit.
#   1. rocprofv3 --kernel-trace --stats of `bench.py --step-only` (the S-scene step and nothing else: per-symbol averages = the line's)
#   2. separate --pmc passes of the SAME command (kernel-trace only; --steps 1 --warmup 1):
#        L2 -> fabric read requests by size | write requests by size | TCC_HIT_sum TCC_MISS_sum | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE
#   S (scenes in flight) for the PMC passes: $M3R_PMC_SCENES (default 28 = the benched step, r06: counters restricted to the GEMM and attention symbols with
#   --kernel-include-regex so that rocprofv3 survives the step -- through r04 it hung at 20 scenes with every kernel counted and the passes ran at 8);
#   a pass that fails is retried at 8 scenes; the scenes of every pass are recorded in its summary.
# Output: gpurun_out/r06_step_* ; scripts/prof_match.py joins them into profiles/r06_roofline_evidence.{json,txt}
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
export M3R_COMMIT=${M3R_COMMIT:-$(cat .commit 2>/dev/null || echo "?")}
O=gpurun_out
S_STEP=${M3R_STEP_SCENES:-28}
S_PMC=${M3R_PMC_SCENES:-28}
mkdir -p $O
rm -rf $O/prof_step
echo "== kernel trace of the step (S=$S_STEP)"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_step -o step -- python bench.py --gpus 1 --steps 3 --warmup 1 --scenes $S_STEP --step-only > $O/r06_step_line.json 2> $O/r06_step.err
echo "rc=$?"; tail -c 300 $O/r06_step_line.json | head -c 300; echo
python scripts/prof_summary.py $(ls $O/prof_step/*.db $O/prof_step/*/*.db 2>/dev/null | tail -1) $O/r06_step_kernel_stats.txt | head -14
find $O/prof_step -name "*.db" -size +20M -delete; find $O/prof_step -name "*.csv" -size +8M -delete
pmc_pass() {   # $1 = tag, rest = counters
  local TAG=$1; shift
  rm -rf $O/pmc_$TAG
  echo "== pmc $TAG: $* (S=$S_PMC)"
  timeout 420 rocprofv3 --kernel-trace --kernel-include-regex "gemm|attn" --pmc "$@" --output-format csv -d $O/pmc_$TAG -o run -- \
      python bench.py --gpus 1 --steps 1 --warmup 1 --scenes $S_PMC --step-only > $O/pmc_$TAG.log 2>&1
  echo "rc=$?"
  python - "$TAG" "$S_PMC" "$@" <<'PY'
import sys, glob, json, csv, re, collections
tag, scenes, ctrs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
files = glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", row.get("Kernel_Name", ""))[:140]
            agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[name][row["Counter_Name"]] += 1
out = {k: {"launches": max(cnt[k].values()), **{c: v / max(1, cnt[k][c]) for c, v in d.items()}} for k, d in agg.items() if "m3r" in k}
import os
if out or not os.path.exists(f"gpurun_out/r06_pmc_{tag}.json"):   # a failed pass (rocprofv3 crash) does not overwrite an earlier good one
    json.dump({"tag": tag, "counters": ctrs, "scenes": scenes, "commit": os.environ.get("M3R_COMMIT", "?"), "per_launch_means": out},
              open(f"gpurun_out/r06_pmc_{tag}.json", "w"), indent=1)
print(tag, "kernels:", len(out))
sys.exit(0 if out else 3)
PY
  local RC=$?
  find $O/pmc_$TAG -name "*.csv" -size +4M -delete 2>/dev/null; find $O/pmc_$TAG -name "*.db" -size +20M -delete 2>/dev/null
  return $RC
}
# fabric traffic: the derived FETCH_SIZE / WRITE_SIZE passes crash rocprofv3 on this image (segfault ~9 s in; r03: hangs) -> on failure the raw
# L2 -> fabric request counters they are derived from (MI355X_MICROARCH.md "HBM": FETCH_SIZE = TCC_EA0_RDREQ x 64 B)
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\(_sum\)\?" | sort -u | head -40 > $O/r06_tcc_ea0_counters.txt
# (rocprofv3 on this image crashes in about one pass out of three with the TCC_EA0 read counters: one retry, at 8 scenes)
pmc_pass fetch TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum || { echo "fetch pass failed at S=$S_PMC -> retry at 8 scenes"; S_PMC=8; pmc_pass fetch TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum; }
S_KEEP=$S_PMC
pmc_pass write TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum || { S_PMC=8; pmc_pass write TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum; S_PMC=$S_KEEP; }
pmc_pass tcc TCC_HIT_sum TCC_MISS_sum || { S_PMC=8; pmc_pass tcc TCC_HIT_sum TCC_MISS_sum; S_PMC=$S_KEEP; }
pmc_pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE || { S_PMC=8; pmc_pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE; S_PMC=$S_KEEP; }
python scripts/prof_match_r06.py
echo "== profile done"
