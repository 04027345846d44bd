#!/bin/bash
# round 2, GPU batch 22: hipGraph replay of single-rank execs -- parity, then host-side latency of small grids with and without
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b22
mkdir -p $OUT
cd $R
K=$R/tools/kbench
( timeout 900 python -m pytest tests -x -q -m gpu -k "graph_replay or reinitialised or single_rank or partial_dimension or golden or cpp or cli" > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt )
{
for n in 64 128 256 512; do
for mode in r2c c2c; do
$K --size $n --prec f64 --mode $mode --iters 200 --latency --label graph
$K --size $n --prec f64 --mode $mode --iters 200 --latency --label plain --opt graph=0
done
done
$K --size 128 --prec f32 --mode r2c --iters 200 --latency --label graph
$K --size 128 --prec f32 --mode r2c --iters 200 --latency --label plain --opt graph=0
$K --size 1024 --prec f64 --mode c2c --iters 10 --latency --label graph
$K --size 1024 --prec f64 --mode c2c --iters 10 --latency --label plain --opt graph=0
} > $OUT/latency.txt 2>&1
cat $OUT/latency.txt
