#!/bin/bash
# round 2, GPU batch 16: native mixed-radix passes (radix 2, 3, 5, 7 butterflies in the Stockham chain) -- parity, then
# per-pass times against the Bluestein kernel on the same grids
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b16
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu -k "mixed or any_length or any_size or fft1d" > $OUT/pytest_gpu.txt 2>&1; tail -15 $OUT/pytest_gpu.txt )
{
for sz in 1000 768 1536x1000x1200 ; do
$K --size $sz --prec f64 --mode c2c --iters 5 --check --label native
done
$K --size 1000 --prec f64 --mode c2c --iters 5 --check --label bluestein --opt native_mixed=0
$K --size 768 --prec f64 --mode c2c --iters 5 --check --label bluestein --opt native_mixed=0
$K --size 1000 --prec f32 --mode c2c --iters 5 --check --label native
$K --size 1000 --prec f32 --mode c2c --iters 5 --check --label bluestein --opt native_mixed=0
$K --size 1536 --prec f32 --mode c2c --iters 5 --check --label native
$K --size 2000x1600x1280 --prec f32 --mode c2c --iters 3 --check --label native
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-12s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
