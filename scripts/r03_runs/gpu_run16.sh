#!/bin/bash
# r03 run 16: the decoder GEMM shapes at M = S * 768 (batched update, S = 8 / 16) under the three tile families: which one should the rule pick now?
mkdir -p gpurun_out
{
for m in 6144 12288; do
for mode in 1 0 2; do
echo "== M $m plain mode $mode"; MROWS=$m M3R_GEMM256=$mode PLAIN16=1 timeout 300 python scripts/exp_gemm256.py
echo "== M $m split mode $mode"; MROWS=$m M3R_GEMM256=$mode SPLIT=1 timeout 300 python scripts/exp_gemm256.py
done; done
} > gpurun_out/r03_gemm_m6144.txt 2>&1
grep -E "==|us " gpurun_out/r03_gemm_m6144.txt
