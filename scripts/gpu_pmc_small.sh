#!/bin/bash
# SQ counters of the M = 768 GEMM launches (scripts/bench_gemm_small.py, split weights)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1 SPLIT=${SPLIT:-1}
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES_EQ_64 SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  rm -rf gpurun_out/pmc_small
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OLDPWD/gpurun_out/pmc_small" -o run -- python "$OLDPWD/scripts/bench_gemm_small.py" > /dev/null 2>&1)
  python - <<'PY'
import glob, csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmc_small/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm_kernel" not in row["Kernel_Name"]: continue
        key = row["Kernel_Name"][20:75] + " grid " + row.get("Grid_Size", "")
        agg[key][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(key, row["Counter_Name"])] += 1
for key, d in sorted(agg.items()):
    print(key, {k: round(v / max(1, cnt[(key, k)])) for k, v in d.items()})
PY
done
