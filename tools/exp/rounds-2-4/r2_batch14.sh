#!/bin/bash
# round 2, GPU batch 14: fp32 R2C / C2R z passes -- in-register split / merge vs LDS split, after moving the Y_Then_ZX
# address forms into their own instantiation (z kernel back to 166 VGPRs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b14
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu -k "r2c or real or sequences or single_rank or golden or cpp" > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt )
{
for rep in 1 2 3; do
$K --size 1024 --prec f32 --mode r2c --iters 5 --check --label paired
$K --size 1024 --prec f32 --mode r2c --iters 5 --check --label lds-split --opt real_variant=1
done
$K --size 512x512x2048 --prec f32 --mode r2c --iters 5 --check --label f32-2048
$K --size 1024 --prec f64 --mode r2c --iters 5 --check --label f64
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-12s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
