#!/bin/bash
# r03 run 9: 16-byte-per-lane stores in the 16-bit epilogues: same bits? how much faster?
mkdir -p gpurun_out
{
echo "== plain, gemm256"; M3R_GEMM256=2 PLAIN16=1 timeout 300 python scripts/exp_gemm256.py
echo "== plain, gemm256k"; M3R_GEMM256=2 PLAIN16=1 M3R_G256K=1 timeout 300 python scripts/exp_gemm256.py
echo "== plain, 4-wave kernels"; M3R_GEMM256=0 PLAIN16=1 timeout 300 python scripts/exp_gemm256.py
echo "== split, default rule"; SPLIT=1 timeout 300 python scripts/exp_gemm256.py
echo "== bf16, default rule"; timeout 300 python scripts/exp_gemm256.py
echo "== fixed cost"; M3R_GEMM256=2 PLAIN16=1 ONLY="k64,k64 f32" timeout 300 python scripts/exp_gemm256.py
} > gpurun_out/r03_epilogue16_ab.txt 2>&1
cat gpurun_out/r03_epilogue16_ab.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -5
