#!/bin/bash
# r03 run 8: what is the fixed per-tile cost made of?  K = 64 launches with / without the epilogue (ablation 32), MFMA-only (39)
export M3R_GEMM256=2 PLAIN16=1 M3R_G256K=1 ONLY="enc qkv,k64,k128,k4096"
mkdir -p gpurun_out
{
for a in 0 32 39; do echo "== gemm256k ablation $a"; M3R_G256K_ABL=$a timeout 300 python scripts/exp_gemm256.py | grep -v "^mode"; done
} > gpurun_out/r03_gemm256k_fixed2.txt 2>&1
cat gpurun_out/r03_gemm256k_fixed2.txt
