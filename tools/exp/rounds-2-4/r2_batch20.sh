#!/bin/bash
# round 2, GPU batch 20: Bluestein lines of 1025..4096 points (inner transforms of 4096 / 8192 points on sub-tile workgroups)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b20
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu -k "any_length or any_size or mixed" > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt )
{
$K --size 256x256x1500 --prec f64 --mode c2c --iters 5 --check --label z1500
$K --size 256x1500x256 --prec f64 --mode c2c --iters 5 --check --label y1500
$K --size 3000x128x128 --prec f64 --mode c2c --iters 5 --check --label x3000
$K --size 128x128x3001 --prec f64 --mode r2c --iters 5 --check --label r2c3001
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-12s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
