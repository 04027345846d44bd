#!/bin/bash
# round 2, GPU batch 27: fp32 mixed-radix configurations, "40+ points per thread count as one more pass" in the generator;
# run once with the library before and once after the change (label given as $1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b27
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
for n in 1280 1200 896 720 600 250; do
$K --size $n --prec f32 --mode c2c --iters 5 --check --label $1
done
$K --size 1200 --prec f32 --mode r2c --iters 5 --check --label $1
} > $OUT/kbench_$1.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench_$1.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-7s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
