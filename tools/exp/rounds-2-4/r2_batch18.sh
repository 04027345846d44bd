#!/bin/bash
# round 2, GPU batch 18: full GPU suite after the mixed-radix generalisation of the butterflies; 1024^3 regression check;
# one-plane split of the mixed real z passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b18
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt )
{
$K --size 1024 --prec f64 --mode c2c --iters 5 --check --label pow2
$K --size 1024 --prec f64 --mode r2c --iters 5 --check --label pow2
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label pow2
$K --size 1000 --prec f64 --mode r2c --iters 5 --check --label oneplane
$K --size 1000 --prec f32 --mode r2c --iters 5 --check --label oneplane
$K --size 1536 --prec f64 --mode r2c --iters 5 --check --label oneplane
$K --size 2000x1600x1280 --prec f32 --mode c2c --iters 3 --check --label native
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-12s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
