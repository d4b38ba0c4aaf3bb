#!/bin/bash
# r03 run 11: does staggering the tiles' epilogues pay?  First-round blocks of gemm256k sleep phase x D (4 phases); D in 0.1 us units x 100
mkdir -p gpurun_out
{
for d in 0 40000 80000 160000 240000; do echo "== delay step $d"; M3R_ST_POLICY=$d M3R_GEMM256=2 M3R_G256K=1 PLAIN16=1 ONLY="enc qkv,dec qkv,enc fc1,enc proj,dec fc1" timeout 300 python scripts/exp_gemm256.py | grep -v "^mode"; done
} > gpurun_out/r03_stagger.txt 2>&1
cat gpurun_out/r03_stagger.txt
