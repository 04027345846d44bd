#!/bin/bash
# round 3, GPU batch 3 (library built with EXTRA=-DDFFT_EXPERIMENTS) = batches 1 + 2 in one call (their results were lost
# with the container they ran in):
#  (a) the whole GPU suite incl. tests/test_gpu_round3.py
#  (b) same-process A/B of the wave-uniform table reads on the multi-rank code path (kbench --sweep: shared buffers)
#  (c) strided-read inverse x pass at 1024^3 fp64: configurations x workgroup orders, persistent forms included
#  (d) 2048-point fp64 x axis, mirrored inverse
#  (e) per-GPU kernels of the 8-GPU plans with the exchange stubbed out (kbench --ranks)
#  (f) bench.py line
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b3
mkdir -p $OUT
cd $R
K=$R/tools/kbench
timeout 900 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_gpu.txt 2>&1; tail -25 $OUT/pytest_gpu.txt
{
echo "== c2c fp64 1024 multi-rank path, uniform_tables 0 | 1 | 0 | 1 (one process, shared buffers)"
timeout 120 $K --size 1024 --prec f64 --iters 5 --opt mirror_inverse=1 --opt pipeline_chunks=8 --sweep "uniform_tables=0;uniform_tables=1;uniform_tables=0;uniform_tables=1"
echo "== c2c fp32 1024 multi-rank path, uniform_tables 0 | 1 | 0 | 1"
timeout 120 $K --size 1024 --prec f32 --iters 5 --opt mirror_inverse=1 --opt pipeline_chunks=8 --sweep "uniform_tables=0;uniform_tables=1;uniform_tables=0;uniform_tables=1"
echo "== c2c fp32 2048 multi-rank path, uniform_tables 0 | 1 | chunks=1 | debug_skip (pattern roofs) chunks=1"
timeout 200 $K --size 2048 --prec f32 --iters 3 --opt mirror_inverse=1 --sweep "pipeline_chunks=8,uniform_tables=0;pipeline_chunks=8,uniform_tables=1;pipeline_chunks=1;pipeline_chunks=1,debug_skip=1"
echo "== c2c fp64 1024 multi-rank path, correctness of uniform tables"; timeout 60 $K --size 1024 --prec f64 --iters 2 --check --opt mirror_inverse=1 --opt pipeline_chunks=8 --opt uniform_tables=1
} > $OUT/tables_sweep.txt 2>&1
grep -E "^==|FFT|PLAN|err|check" $OUT/tables_sweep.txt | cut -c1-170
{
echo "== x^-1 at 1024^3 fp64, mirrored inverse, one process: variant_ix x order_ix"
timeout 300 $K --size 1024 --prec f64 --iters 5 --opt mirror_inverse=1 --sweep "variant_ix=1;variant_ix=0;variant_ix=3;variant_ix=12;variant_ix=13;variant_ix=8;variant_ix=10;variant_ix=9;variant_ix=11;variant_ix=1,order_ix=0;variant_ix=1,order_ix=1;variant_ix=1,order_ix=2;variant_ix=1,order_ix=3;variant_ix=12,order_ix=0;variant_ix=12,order_ix=1;variant_ix=12,order_ix=2;variant_ix=12,order_ix=3;variant_ix=1"
for v in 8 10 12; do
  echo "== x^-1 variant_ix=$v --check"; timeout 60 $K --size 1024 --prec f64 --iters 2 --check --opt mirror_inverse=1 --opt variant_ix=$v | grep -E "x-FFT\^-1|PLAN|err|check"
done
} > $OUT/xinv_1024.txt 2>&1
grep -E "^==|x-FFT\^-1|sweep|err|check" $OUT/xinv_1024.txt | cut -c1-170
{
echo "== 2048x1024x1024 f64 mirrored inverse: variant_ix sweep"
timeout 200 $K --size 2048x1024x1024 --prec f64 --iters 3 --opt mirror_inverse=1 --sweep "variant_ix=1;variant_ix=0;variant_ix=3;variant_ix=9;variant_ix=12;variant_ix=8;variant_ix=10;variant_ix=1"
echo "== 2048x1024x1024 f64 forward x variant 9 / 12"; timeout 120 $K --size 2048x1024x1024 --prec f64 --iters 3 --sweep "variant_fx=0;variant_fx=9;variant_fx=12;variant_fx=0"
} > $OUT/x2048_f64.txt 2>&1
grep -E "^==|x-FFT|sweep" $OUT/x2048_f64.txt | cut -c1-140
{
for g in 2x4 8x1; do
  echo "== 1024^3 fp64, rank 0 of $g"; timeout 60 $K --size 1024 --prec f64 --iters 20 --ranks $g
  echo "== 1024^3 fp64 r2c, rank 0 of $g"; timeout 60 $K --size 1024 --prec f64 --mode r2c --iters 20 --ranks $g
  echo "== 2048^3 fp32, rank 0 of $g"; timeout 60 $K --size 2048 --prec f32 --iters 10 --ranks $g
  echo "== 2048^3 fp32, rank 0 of $g, uniform_tables=0"; timeout 60 $K --size 2048 --prec f32 --iters 10 --ranks $g --opt uniform_tables=0
done
echo "== 1024^3 fp64, rank 5 of 2x4"; timeout 60 $K --size 1024 --prec f64 --iters 20 --ranks 2x4 --rank 5
} > $OUT/per_gpu_kernels.txt 2>&1
grep -E "^==|FFT|PLAN|total" $OUT/per_gpu_kernels.txt | cut -c1-170
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; echo; tail -3 $OUT/bench.err
