#!/bin/bash
# round 2, GPU batch 6: parity with the reworked real-transform kernels and the reference-caller tests, the new bench.py,
# R2C point-fastest forms, workgroup-order sweeps of the tiled-store passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b6
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt )
( time python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err; cat $OUT/bench.time
{
echo "=== R2C/C2R real kernels: new point-fastest fp32 forms (0) vs round-1 final (3)"
for rv in 0 3; do
$K --size 1024 --prec f32 --mode r2c --iters 5 --check --label rv$rv --opt real_variant=$rv
$K --size 2048x512x2048 --prec f32 --mode r2c --iters 3 --check --label rv$rv --opt real_variant=$rv
done
$K --size 1024 --prec f64 --mode r2c --iters 5 --check --label rv0
$K --size 1024 --prec f64 --mode r2c --iters 5 --check --label rv1 --opt real_variant=1
$K --size 1024x1024x512 --prec f64 --mode c2c --iters 5 --label c2c-z512
$K --size 1024x1024x512 --prec f64 --mode c2c --iters 5 --label c2c-z512-roof --opt debug_skip=1
echo "=== workgroup order of the y pass (tiled load -> tiled-same store)"
for o in 0 1 2 3; do
$K --size 1024 --prec f64 --iters 4 --label order_fy$o --opt order_fy=$o
done
for o in 0 1 2 3; do
$K --size 2048 --prec f32 --iters 2 --label order_fy$o --opt variant_fz=7 --opt order_fy=$o
done
for o in 0 1 3; do
$K --size 2048 --prec f32 --iters 2 --label order_fx$o --opt variant_fz=7 --opt order_fx=$o
done
} > $OUT/kbench.txt 2>&1
grep -c PLAN $OUT/kbench.txt
