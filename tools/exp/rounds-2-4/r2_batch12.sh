#!/bin/bash
# round 2, GPU batch 12: packed fp32 complex arithmetic (native 2-vectors) -- parity and per-pass times
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b12
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt )
{
$K --size 1024 --prec f32 --iters 5 --check --label c2c
$K --size 1024 --prec f32 --iters 5 --check --label c2c-multirank --opt mirror_inverse=1 --opt pipeline_chunks=8
$K --size 1024 --prec f32 --mode r2c --iters 5 --check --label r2c
$K --size 2048 --prec f32 --iters 3 --check --label c2c-auto
$K --size 2048 --prec f32 --iters 3 --check --label c2c-zyx --opt single_order=0
$K --size 2048 --prec f32 --iters 3 --label c2c-multirank --opt mirror_inverse=1 --opt pipeline_chunks=8
$K --size 1024 --prec f64 --iters 5 --check --label c2c
$K --size 1024 --prec f64 --mode r2c --iters 5 --check --label r2c
$K --line 2048 --batch 131072 --prec f32 --variant 4 --iters 5 --check
$K --line 1024 --batch 262144 --prec f32 --variant 4 --iters 5 --check
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-16s %-22s %s: ", $2, $3" "$4" "$5, $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo; grep LINE $OUT/kbench.txt
