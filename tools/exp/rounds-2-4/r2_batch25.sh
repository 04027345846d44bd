#!/bin/bash
# round 2, GPU batch 25: packed real z passes of the mixed-radix lengths whose kernels spill (30+ points per thread) against
# the Bluestein kernel's real mode on the same Nz (compare the z-FFT / z-FFT^-1 columns)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b25
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
for nz in 1200 1440 600 400; do
for prec in f64 f32; do
$K --size 256x256x$nz --prec $prec --mode r2c --iters 5 --check --label native
$K --size 256x256x$nz --prec $prec --mode r2c --iters 5 --check --label bluestein --opt native_mixed=0
done
done
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-10s %-20s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
