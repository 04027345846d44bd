#!/bin/bash
# round 4, GPU batch 6: the inverse y pass of C5 (tiled load, transposed-tile store) with whole-line stores at 32 points per thread
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b6
mkdir -p $OUT
cd $R
K=$R/tools/kbench_exp
S=""; C=""
for v in 0 5 15 10 11 3 2 7; do S="$S${S:+;}variant_iy=$v"; C="$C${C:+;}debug_skip=1,variant_iy=$v"; done
{
echo "== 2048^3 fp32, rank 0 of 2x4: inverse y pass on configuration 0 5 15 10 11 3 2 7 (set order), transforms"
timeout 300 $K --size 2048 --prec f32 --iters 5 --ranks 2x4 --sweep "$S" 2>&1 | grep -E "^PLAN|y-FFT\^-1"
echo "== as copies"
timeout 300 $K --size 2048 --prec f32 --iters 5 --ranks 2x4 --sweep "$C" 2>&1 | grep -E "^PLAN|y-FFT\^-1"
echo "== slab 8x1 rank 0 (no second exchange on y: the same pass with P2 = 1 segments)"
timeout 300 $K --size 2048 --prec f32 --iters 5 --ranks 8x1 --sweep "$S" 2>&1 | grep -E "^PLAN|y-FFT\^-1"
} > $OUT/r4_f32_2048_inverse_y_whole_lines.txt 2>&1
grep -E "y-FFT" $OUT/r4_f32_2048_inverse_y_whole_lines.txt | cut -c1-100
