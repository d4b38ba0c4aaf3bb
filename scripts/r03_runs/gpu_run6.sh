#!/bin/bash
# r03 run 6: timing ablations of gemm256k (results wrong by construction): which of DMA / fragment reads / barrier / MFMA costs what
export M3R_GEMM256=2 PLAIN16=1 M3R_G256K=1 ONLY="enc qkv,dec qkv"
mkdir -p gpurun_out
{
for a in 0 1 2 4 8 3 5 6 7 9 10 12 20 21 23; do echo "== ablation $a"; M3R_G256K_ABL=$a timeout 300 python scripts/exp_gemm256.py | grep -v "^mode"; done
} > gpurun_out/r03_gemm256k_ablation.txt 2>&1
cat gpurun_out/r03_gemm256k_ablation.txt
