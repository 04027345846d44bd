#!/bin/bash
# round 3, GPU batch 1 (library built with EXTRA=-DDFFT_EXPERIMENTS):
#  (a) GPU parity suite with the wave-uniform (scalar) table reads on by default
#  (b) multi-rank code path (mirrored inverse, 8 pipeline chunks): uniform_tables = 0 (per-lane 16-byte table loads, round 2) vs 1
#  (c) strided-read inverse x pass at 1024^3 fp64: configurations x workgroup orders, persistent forms included
#  (d) 2048-point fp64 x axis, mirrored inverse
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b1
mkdir -p $OUT
cd $R
K=$R/tools/kbench
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
{
for u in 0 1; do
  echo "== c2c fp64 1024 multi-rank path uniform_tables=$u"; $K --size 1024 --prec f64 --iters 5 --check --opt mirror_inverse=1 --opt pipeline_chunks=8 --opt uniform_tables=$u
  echo "== c2c fp32 1024 multi-rank path uniform_tables=$u"; $K --size 1024 --prec f32 --iters 5 --check --opt mirror_inverse=1 --opt pipeline_chunks=8 --opt uniform_tables=$u
  echo "== c2c fp32 2048 multi-rank path uniform_tables=$u"; $K --size 2048 --prec f32 --iters 3 --check --opt mirror_inverse=1 --opt pipeline_chunks=8 --opt uniform_tables=$u
done
echo "== c2c fp64 1024 unchunked mirrored"; $K --size 1024 --prec f64 --iters 5 --check --opt mirror_inverse=1
echo "== c2c fp32 2048 unchunked mirrored"; $K --size 2048 --prec f32 --iters 3 --opt mirror_inverse=1
} > $OUT/tables_ab.txt 2>&1
grep -E "^==|FFT|PLAN" $OUT/tables_ab.txt | cut -c1-150
{
for v in 0 1 3 12 13; do for o in -1 0 1 2 3; do
  echo "== x^-1 variant_ix=$v order_ix=$o"; $K --size 1024 --prec f64 --iters 5 --opt mirror_inverse=1 --opt variant_ix=$v --opt order_ix=$o | grep -E "x-FFT\^-1|PLAN"
done; done
for v in 8 10; do
  echo "== x^-1 variant_ix=$v (persistent)"; $K --size 1024 --prec f64 --iters 5 --check --opt mirror_inverse=1 --opt variant_ix=$v | grep -E "x-FFT\^-1|PLAN"
done
} > $OUT/xinv_1024.txt 2>&1
grep -E "^==|x-FFT" $OUT/xinv_1024.txt | paste - - | awk '{print $3, $4, $7, $8, $10, $11}'
{
for v in 0 3 9 12 8 10; do
  echo "== 2048x1024x1024 f64 variant_ix=$v"; $K --size 2048x1024x1024 --prec f64 --iters 3 --check --opt mirror_inverse=1 --opt variant_ix=$v | grep -E "FFT|PLAN"
done
echo "== 2048x1024x1024 f64 forward x variant 9 / 12"; for v in 9 12; do $K --size 2048x1024x1024 --prec f64 --iters 3 --check --opt variant_fx=$v | grep -E "FFT|PLAN"; done
} > $OUT/x2048_f64.txt 2>&1
grep -E "^==|x-FFT" $OUT/x2048_f64.txt | cut -c1-120
