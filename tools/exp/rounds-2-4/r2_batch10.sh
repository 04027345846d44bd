#!/bin/bash
# round 2, GPU batch 10: single-rank pass order z, x, y with padded private layouts against z, y, x
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b10
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt )
{
for rep in 1 2 3; do
  $K --size 1024 --prec f64 --iters 5 --check --label rep$rep-zyx --opt single_order=0
  $K --size 1024 --prec f64 --iters 5 --check --label rep$rep-zxy-tileouter-pad128
  $K --size 1024 --prec f64 --iters 5 --label rep$rep-zxy-rows-pad128 --opt single_layout=0
done
$K --size 1024 --prec f64 --iters 5 --label zxy-tileouter-pad0 --opt single_pad=0
$K --size 1024 --prec f64 --iters 5 --label zxy-rows-pad0 --opt single_layout=0 --opt single_pad=0
$K --size 1024 --prec f64 --iters 5 --label zxy-tileouter-pad384 --opt single_pad=384
$K --size 1024 --prec f64 --iters 5 --label zxy-tileouter-pad1152 --opt single_pad=1152
for o in 0 1 2; do
  $K --size 1024 --prec f64 --iters 5 --label zxy-order_sx$o --opt order_fx=$o
  $K --size 1024 --prec f64 --iters 5 --label zxy-order_sy$o --opt order_fy=$o
  $K --size 1024 --prec f64 --iters 5 --label zxy-order_sz$o --opt order_fz=$o
done
$K --size 1024 --prec f64 --iters 5 --label zxy-nt-x --opt variant_fx=3
$K --size 1024 --prec f64 --iters 5 --label zxy-nt-y --opt variant_fy=3
$K --size 1024 --prec f64 --iters 5 --label zxy-nt-xy --opt variant_fx=3 --opt variant_fy=3
$K --size 1024 --prec f64 --iters 5 --label zxy-z-plain --opt variant_fz=0
$K --size 1024 --prec f64 --iters 5 --label zxy-roof --opt debug_skip=1
echo "=== fp32"
for rep in 1 2; do
  $K --size 1024 --prec f32 --iters 5 --check --label rep$rep-zyx --opt single_order=0
  $K --size 1024 --prec f32 --iters 5 --check --label rep$rep-zxy
  $K --size 1024 --prec f32 --iters 5 --label rep$rep-zxy-rows --opt single_layout=0
done
$K --size 2048 --prec f32 --iters 3 --check --label zyx --opt single_order=0
$K --size 2048 --prec f32 --iters 3 --check --label zxy
$K --size 2048 --prec f32 --iters 3 --label zxy-rows --opt single_layout=0
echo "=== other shapes"
$K --size 1024x1024x2048 --prec f64 --iters 3 --check --label zxy
$K --size 2048x1024x1024 --prec f64 --iters 3 --check --label zxy
$K --size 1024x2048x1024 --prec f64 --iters 3 --check --label zxy
$K --size 1000 --prec f64 --iters 3 --check --label zxy-bluestein
$K --size 512 --prec f64 --iters 10 --check --label zxy
$K --size 512 --prec f64 --iters 10 --check --label zyx --opt single_order=0
$K --size 256 --prec f64 --iters 20 --check --label zxy
$K --size 256 --prec f64 --iters 20 --check --label zyx --opt single_order=0
} > $OUT/kbench.txt 2>&1
grep -A7 "^PLAN" $OUT/kbench.txt | grep -E "PLAN|FFT" | awk '/PLAN/{printf "\n%-28s %-22s %s: ", $2, $3" "$4, $(NF-1)} !/PLAN/{printf "%s %s  ", $1, $3}'; echo
