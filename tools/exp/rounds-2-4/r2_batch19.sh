#!/bin/bash
# round 2, GPU batch 19: 4096- and 8192-point lines (sub-tile workgroups), R2C with Nz = 4096
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b19
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu -k "fft1d_batched or single_rank_3d or r2c_c2r_vs_oracle or distributed_vs_oracle" > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt )
{
$K --size 256x256x4096 --prec f64 --mode c2c --iters 5 --check --label z4096
$K --size 256x4096x256 --prec f64 --mode c2c --iters 5 --check --label y4096
$K --size 4096x256x256 --prec f64 --mode c2c --iters 5 --check --label x4096
$K --size 128x128x8192 --prec f64 --mode c2c --iters 5 --check --label z8192
$K --size 128x8192x128 --prec f64 --mode c2c --iters 5 --check --label y8192
$K --size 256x256x4096 --prec f64 --mode r2c --iters 5 --check --label r2c4096
$K --size 256x4096x256 --prec f32 --mode c2c --iters 5 --check --label y4096
$K --size 256x256x4096 --prec f32 --mode r2c --iters 5 --check --label r2c4096
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-12s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
