#!/bin/bash
# r03 run 10: cache policy of the 16-bit output stores (0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt); a chain qkv -> consumer shows the boundary cost too
mkdir -p gpurun_out
{
for pol in 0 1 2 3 4; do echo "== policy $pol"; M3R_ST_POLICY=$pol M3R_GEMM256=2 PLAIN16=1 ONLY="enc qkv,dec qkv,dec kv,k64,enc fc1" timeout 300 python scripts/exp_gemm256.py; done
} > gpurun_out/r03_store_policy.txt 2>&1
cat gpurun_out/r03_store_policy.txt
