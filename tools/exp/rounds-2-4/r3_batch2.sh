#!/bin/bash
# round 3, GPU batch 2 (library built with EXTRA=-DDFFT_EXPERIMENTS):
#  (a) the whole GPU suite incl. tests/test_gpu_round3.py (every-point C4, full-size C5, forced z,x,y order, uniform tables)
#  (b) same-process A/B (kbench --sweep: shared buffers) of the wave-uniform table reads on the multi-rank code path
#  (c) pattern roofs (debug_skip) of the mirrored order at 2048^3 fp32
#  (d) per-GPU kernels of the 8-GPU plans with the exchange stubbed out (kbench --ranks)
#  (e) bench.py line
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b2
mkdir -p $OUT
cd $R
K=$R/tools/kbench
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_gpu.txt 2>&1; tail -25 $OUT/pytest_gpu.txt
{
echo "== c2c fp64 1024 multi-rank path, uniform_tables 0 | 1 | 0 | 1 (one process, shared buffers)"
$K --size 1024 --prec f64 --iters 5 --opt mirror_inverse=1 --opt pipeline_chunks=8 --sweep "uniform_tables=0;uniform_tables=1;uniform_tables=0;uniform_tables=1"
echo "== c2c fp32 1024 multi-rank path, uniform_tables 0 | 1 | 0 | 1"
$K --size 1024 --prec f32 --iters 5 --opt mirror_inverse=1 --opt pipeline_chunks=8 --sweep "uniform_tables=0;uniform_tables=1;uniform_tables=0;uniform_tables=1"
echo "== c2c fp32 2048 multi-rank path, uniform_tables 0 | 1 | chunks=1 | debug_skip (pattern roofs) chunks=1"
$K --size 2048 --prec f32 --iters 3 --opt mirror_inverse=1 --sweep "pipeline_chunks=8,uniform_tables=0;pipeline_chunks=8,uniform_tables=1;pipeline_chunks=1;pipeline_chunks=1,debug_skip=1"
} > $OUT/tables_sweep.txt 2>&1
grep -E "^==|FFT|PLAN" $OUT/tables_sweep.txt | cut -c1-170
{
for g in 2x4 8x1; do
  echo "== 1024^3 fp64, rank 0 of $g"; $K --size 1024 --prec f64 --iters 20 --ranks $g
  echo "== 1024^3 fp64 r2c, rank 0 of $g"; $K --size 1024 --prec f64 --mode r2c --iters 20 --ranks $g
  echo "== 2048^3 fp32, rank 0 of $g"; $K --size 2048 --prec f32 --iters 10 --ranks $g
  echo "== 2048^3 fp32, rank 0 of $g, uniform_tables=0"; $K --size 2048 --prec f32 --iters 10 --ranks $g --opt uniform_tables=0
done
echo "== 1024^3 fp64, rank 5 of 2x4"; $K --size 1024 --prec f64 --iters 20 --ranks 2x4 --rank 5
} > $OUT/per_gpu_kernels.txt 2>&1
grep -E "^==|FFT|PLAN|total" $OUT/per_gpu_kernels.txt | cut -c1-170
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json; echo; tail -3 $OUT/bench.err
