#!/bin/bash
# round 2, GPU batch 26: fp32 mixed-radix configurations with fewer points per thread (generator preference 24 instead of 32):
# 1000, 800, 400, 200, 2000-point lines; before: 1000^3 fp32 C2C 35.8 ms (z 7.39 y 5.73 x 4.79), R2C+C2R 17.8 ms,
# 2000x1600x1280 C2C 136.3 ms
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b26
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 600 python -m pytest tests -x -q -m gpu -k "mixed" > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt )
{
$K --size 1000 --prec f32 --mode c2c --iters 5 --check --label e20
$K --size 1000 --prec f32 --mode r2c --iters 5 --check --label e20
$K --size 800 --prec f32 --mode c2c --iters 5 --check --label e20
$K --size 400 --prec f32 --mode c2c --iters 10 --check --label e20
$K --size 2000x1600x1280 --prec f32 --mode c2c --iters 3 --check --label e20
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-6s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
