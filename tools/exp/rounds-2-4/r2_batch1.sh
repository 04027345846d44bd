#!/bin/bash
# round 2, GPU batch 1: parity after the kernel refactor, baselines, access-pattern roofs (debug_skip), sub-tile and
# 2048-point configurations.  Everything through tools/kbench (C ABI, no Python) except the parity suite.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b1
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt ) 
{
echo "=== A. baselines with self-checks"
$K --size 1024 --prec f64 --mode c2c --iters 5 --check --label base
$K --size 1024 --prec f64 --mode c2c --iters 5 --check --label multirank-path --opt mirror_inverse=1 --opt pipeline_chunks=8
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label base
$K --size 1024 --prec f64 --mode r2c --iters 5 --check --label base
$K --size 1024 --prec f32 --mode r2c --iters 5 --check --label base
echo "=== B. pattern roofs: every pass as a copy with its own access pattern"
$K --size 1024 --prec f64 --mode c2c --iters 5 --label roof --opt debug_skip=1
$K --size 1024 --prec f64 --mode c2c --iters 5 --label roof-multirank --opt debug_skip=1 --opt mirror_inverse=1 --opt pipeline_chunks=8
$K --size 1024 --prec f32 --mode c2c --iters 5 --label roof --opt debug_skip=1
$K --size 1024 --prec f64 --mode r2c --iters 5 --label roof --opt debug_skip=1
$K --size 1024 --prec f32 --mode r2c --iters 5 --label roof --opt debug_skip=1
echo "=== C. 1024 fp64 variants"
$K --size 1024 --prec f64 --mode c2c --iters 5 --check --label sub-tiles-all --opt variant_fz=6 --opt variant_fy=5 --opt variant_fx=5
$K --size 1024 --prec f64 --mode c2c --iters 5 --label sub-tiles-all-roof --opt variant_fz=6 --opt variant_fy=5 --opt variant_fx=5 --opt debug_skip=1
$K --size 1024 --prec f64 --mode c2c --iters 5 --label nt-everywhere --opt variant_fz=3 --opt variant_fy=3 --opt variant_fx=3
$K --size 1024 --prec f64 --mode c2c --iters 5 --label no-nt --opt variant_fz=0 --opt variant_fy=0 --opt variant_fx=0
for v in 0 1 4 5 7; do
$K --size 1024 --prec f64 --mode c2c --iters 5 --check --label mirror-ix$v --opt mirror_inverse=1 --opt variant_ix=$v
done
for o in 0 1 2 3; do
$K --size 1024 --prec f64 --mode c2c --iters 4 --label mirror-ix-order$o --opt mirror_inverse=1 --opt order_ix=$o
$K --size 1024 --prec f64 --mode c2c --iters 4 --label mirror-ix-order$o-roof --opt mirror_inverse=1 --opt order_ix=$o --opt debug_skip=1
done
echo "=== D. 1024 fp32 variants (all rebuilt without the SLP vectorizer)"
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label lf-3pass --opt variant_fz=0 --opt variant_fy=0 --opt variant_fx=0
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label lf-2pass --opt variant_fz=6 --opt variant_fy=6 --opt variant_fx=6
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label pf-2pass --opt variant_fz=7 --opt variant_fy=6 --opt variant_fx=6
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label sub-2pass --opt variant_fz=9 --opt variant_fy=9 --opt variant_fx=9
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label e16 --opt variant_fz=1 --opt variant_fy=1 --opt variant_fx=1
$K --size 1024 --prec f32 --mode c2c --iters 5 --check --label mirror --opt mirror_inverse=1 --opt pipeline_chunks=8
echo "=== E. 2048-point axis, natural lines (one pass, 2 GiB)"
for v in 0 3 5 6 2 4; do
$K --line 2048 --batch 65536 --prec f64 --variant $v --iters 5 --check
done
$K --line 2048 --batch 65536 --prec f64 --variant 0 --iters 5 --debug 1
$K --line 2048 --batch 65536 --prec f64 --variant 5 --iters 5 --debug 1
for v in 0 1 3 4 5 6 7 8; do
$K --line 2048 --batch 131072 --prec f32 --variant $v --iters 5 --check
done
$K --line 2048 --batch 131072 --prec f32 --variant 0 --iters 5 --debug 1
$K --line 2048 --batch 131072 --prec f32 --variant 1 --iters 5 --debug 1
$K --line 2048 --batch 131072 --prec f32 --variant 4 --iters 5 --debug 1
echo "=== F. 2048-point axis inside plans: z / y / x axis of length 2048 (other axes 256)"
for sz in 256x256x2048 256x2048x256 2048x256x256; do
  for p in f64 f32; do
    $K --size $sz --prec $p --mode c2c --iters 5 --check --label default
    $K --size $sz --prec $p --mode c2c --iters 5 --label roof --opt debug_skip=1
    $K --size $sz --prec $p --mode c2c --iters 5 --check --label mirror --opt mirror_inverse=1
  done
done
for sz in 256x256x2048 256x2048x256 2048x256x256; do
  for v in 5 6 2 4; do
    $K --size $sz --prec f64 --mode c2c --iters 5 --check --label all-v$v --opt variant_fz=$v --opt variant_fy=$v --opt variant_fx=$v --opt variant_ix=$v --opt variant_iy=$v --opt variant_iz=$v --opt mirror_inverse=1
  done
  for v in 0 1 3 6; do
    $K --size $sz --prec f32 --mode c2c --iters 5 --check --label all-v$v --opt variant_fz=$v --opt variant_fy=$v --opt variant_fx=$v --opt variant_ix=$v --opt variant_iy=$v --opt variant_iz=$v --opt mirror_inverse=1
  done
done
} > $OUT/kbench.txt 2>&1
grep -c PLAN $OUT/kbench.txt; tail -5 $OUT/kbench.txt
