#!/bin/bash
# r03 run 14: timing ablations of attn3_kernel (experiment build; results wrong by construction): 5 no DMA, 6 no barrier, 7 both, 8 + no LDS reads, 9 MFMAs only
mkdir -p gpurun_out
{
for a in 0 3 5 6 7 8 9; do echo "ABL=$a"; M3R_ATTN_ABL=$a timeout 300 python scripts/bench_attn.py 2>&1 | grep -E "render CA 20v nk15360 |enc SA|update CA 4v nk7680 \(S"; done
} > gpurun_out/r03_attn3_ablation.txt 2>&1
cat gpurun_out/r03_attn3_ablation.txt
