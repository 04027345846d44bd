#!/bin/bash
# (historical: the FLAT configurations existed only for this measurement and were removed afterwards, profiles/r2_flat_staging.txt)
# round 2, GPU batch 24: FLAT natural-line staging for the mixed-radix z passes (roles 7 / 4 / 5) against the direct form
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b24
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu -k "flat or mixed or any_size or partial_dimension" > $OUT/pytest_gpu.txt 2>&1; tail -8 $OUT/pytest_gpu.txt )
{
for rep in 1 2; do
$K --size 1000 --prec f64 --mode c2c --iters 5 --check --label flat
$K --size 1000 --prec f64 --mode c2c --iters 5 --check --label direct --opt variant_fz=0 --opt variant_iz=0
$K --size 1000 --prec f32 --mode c2c --iters 5 --check --label flat
$K --size 1000 --prec f32 --mode c2c --iters 5 --check --label direct --opt variant_fz=0 --opt variant_iz=0
done
$K --size 1536 --prec f32 --mode c2c --iters 3 --check --label flat
$K --size 1536 --prec f32 --mode c2c --iters 3 --check --label direct --opt variant_fz=0 --opt variant_iz=0
$K --size 512x512x2000 --prec f64 --mode c2c --iters 5 --check --label flat
$K --size 512x512x2000 --prec f64 --mode c2c --iters 5 --check --label direct --opt variant_fz=0 --opt variant_iz=0
$K --size 768 --prec f32 --mode c2c --iters 5 --check --label flat
$K --size 768 --prec f32 --mode c2c --iters 5 --check --label direct --opt variant_fz=0 --opt variant_iz=0
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-8s %-22s %s %s: ", $2, $3" "$4" "$5, $(NF-4), $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
