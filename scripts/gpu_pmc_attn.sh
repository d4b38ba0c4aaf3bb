#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python scripts/bench_attn.py 2>&1 | grep -v amdgpu.ids
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  rm -rf gpurun_out/pmc_attn
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d gpurun_out/pmc_attn -o run -- python scripts/bench_attn.py > /dev/null 2>&1
  python - <<'PY'
import glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_attn/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "attn3_kernel" not in row["Kernel_Name"]: continue
        key = (row["Grid_Size"] if "Grid_Size" in row else "")
        agg[key][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(key, row["Counter_Name"])] += 1
for key, d in agg.items():
    print("grid", key, {k: round(v / max(1, cnt[(key, k)])) for k, v in d.items()})
PY
done
