#!/bin/bash
# round 2, GPU batch 13: fp64 R2C / C2R z passes with the Hermitian split / merge in registers (conjugate-pair butterflies)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b13
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt )
{
for rep in 1 2; do
$K --size 1024 --prec f64 --mode r2c --iters 5 --check --label paired
$K --size 1024 --prec f64 --mode r2c --iters 5 --check --label lds-split --opt real_variant=1
$K --size 512x512x2048 --prec f64 --mode r2c --iters 5 --check --label paired
$K --size 512x512x2048 --prec f64 --mode r2c --iters 5 --check --label lds-split --opt real_variant=1
done
for rep in 1 2 3; do $K --size 1024 --prec f32 --mode r2c --iters 5 --check --label f32; done
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-12s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
