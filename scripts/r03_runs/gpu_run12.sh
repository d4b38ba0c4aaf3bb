#!/bin/bash
# r03 run 12: gemm256k on split weights (256 x 128 tiles, hi+lo) against the gemm256 rule
mkdir -p gpurun_out
{
echo "== split, gemm256 rule"; SPLIT=1 timeout 300 python scripts/exp_gemm256.py
echo "== split, gemm256k"; SPLIT=1 M3R_G256K=2 timeout 300 python scripts/exp_gemm256.py
echo "== split, gemm256k forced"; SPLIT=1 M3R_G256K=2 M3R_GEMM256=2 timeout 300 python scripts/exp_gemm256.py
echo "== split, 4-wave"; SPLIT=1 M3R_GEMM256=0 timeout 300 python scripts/exp_gemm256.py
} > gpurun_out/r03_gemm256k_split.txt 2>&1
cat gpurun_out/r03_gemm256k_split.txt
