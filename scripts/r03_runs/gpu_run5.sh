#!/bin/bash
# r03 run 5: gemm256k (64-deep chunks, 128-byte DMA rows) against gemm256 on the chip-filling shapes, plain fp16 weights
export M3R_GEMM256=2 PLAIN16=1
mkdir -p gpurun_out
{
echo "== gemm256 (K-tile 32)"; M3R_G256K=0 timeout 300 python scripts/exp_gemm256.py
for v in 0 1 2 3; do echo "== gemm256k variant $v"; M3R_G256K=1 M3R_G256K_VAR=$v timeout 300 python scripts/exp_gemm256.py; done
} > gpurun_out/r03_gemm256k_ab.txt 2>&1
cat gpurun_out/r03_gemm256k_ab.txt
