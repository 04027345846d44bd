#!/bin/bash
# round 2, GPU batch 17: packed real z passes on mixed-radix lengths (R2C / C2R at Nz = 1000, 1536, 2000 ...), sub-tile
# fp32 configurations for 1600 / 2000
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b17
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu -k "mixed or any_length or any_size or r2c" > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt )
{
$K --size 1000 --prec f64 --mode r2c --iters 5 --check --label native
$K --size 1000 --prec f64 --mode r2c --iters 5 --check --label bluestein --opt native_mixed=0
$K --size 1536 --prec f64 --mode r2c --iters 5 --check --label native
$K --size 1000 --prec f32 --mode r2c --iters 5 --check --label native
$K --size 2000x1600x1280 --prec f32 --mode c2c --iters 3 --check --label native
$K --size 768x800x2000 --prec f64 --mode r2c --iters 3 --check --label native
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-12s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
