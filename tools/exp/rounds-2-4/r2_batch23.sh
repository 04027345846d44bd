#!/bin/bash
# (historical: the option y_pad existed only for this measurement and was removed afterwards, profiles/r2_y_pad.txt)
# round 2, GPU batch 23: padded rows in the y pass's private output (single-rank C2C, order z, y, x): row stride 16 MiB + pad
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b23
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
for rep in 1 2 3; do
for pad in 0 128 384 1152; do
$K --size 1024 --prec f64 --mode c2c --iters 5 --check --label rep$rep-pad$pad --opt y_pad=$pad
done
done
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label f32-pad0
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label f32-pad128 --opt y_pad=128
$K --size 512 --prec f64 --mode c2c --iters 10 --check --label 512-pad0
$K --size 512 --prec f64 --mode c2c --iters 10 --check --label 512-pad128 --opt y_pad=128
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-14s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
