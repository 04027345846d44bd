#!/bin/bash
# round 4, GPU batch 9: whole tiles at 32 points per thread WITH hints (A/B 8) on the tiled 2048-point passes; parity of the shipped
# configurations; the bench lines of N ranks sharing this GPU (gloo; relay leg behind its watchdog)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b9
mkdir -p $OUT
cd $R
K=$R/tools/kbench_exp
{
echo "== 2048^3 fp32, rank 0 of 2x4: y, x, x^-1 on configuration 6 | 9 | 0 | 8 (set order); y^-1 stays on 7"
timeout 300 $K --size 2048 --prec f32 --iters 5 --ranks 2x4 --sweep "variant_fy=6,variant_fx=6,variant_ix=6;variant_fy=9,variant_fx=9,variant_ix=9;variant_fy=0,variant_fx=0,variant_ix=0;variant_fy=8,variant_fx=8,variant_ix=8" 2>&1 | grep -E "^PLAN|FFT|total" | cut -c1-150
} > $OUT/r4_f32_2048_whole_tiles_32_hints.txt 2>&1
grep -E "y-FFT |x-FFT" $OUT/r4_f32_2048_whole_tiles_32_hints.txt
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_placement.py tests/test_gpu_round3.py -m gpu -q -x --durations=6 -k "configuration or bench or tune" > $OUT/r4_pytest_b9.txt 2>&1
tail -14 $OUT/r4_pytest_b9.txt
