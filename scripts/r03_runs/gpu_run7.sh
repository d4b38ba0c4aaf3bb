#!/bin/bash
# r03 run 7: fixed per-tile cost (K = 64 / 128) and long-K rate of gemm256k and gemm256; MFMA-only / no-MFMA ablations at long K
export M3R_GEMM256=2 PLAIN16=1 ONLY="enc qkv,k64,k128,k4096,k64 f32,k16384"
mkdir -p gpurun_out
{
echo "== gemm256"; M3R_G256K=0 timeout 300 python scripts/exp_gemm256.py | grep -v "^mode"
for a in 0 7 8 1 2 4; do echo "== gemm256k ablation $a"; M3R_G256K=1 M3R_G256K_ABL=$a timeout 300 python scripts/exp_gemm256.py | grep -v "^mode"; done
} > gpurun_out/r03_gemm256k_fixed.txt 2>&1
cat gpurun_out/r03_gemm256k_fixed.txt
