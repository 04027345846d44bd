#!/bin/bash
# round 3, GPU batch 14: two-level lines, per-pass times (a long axis in each position, against the nearest one-launch lengths)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b14
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
for sz in 16384x128x128 128x16384x128 128x128x16384 8192x128x256 128x8192x256 128x256x8192 10000x128x128 128x128x10000 65536x64x64 64x64x65536; do
  for p in f64 f32; do
    echo "== c2c $p $sz"; timeout 120 $K --size $sz --prec $p --mode c2c --iters 5 --check
  done
done
for sz in 128x128x16384 128x128x10000; do
  echo "== r2c f64 $sz"; timeout 120 $K --size $sz --prec f64 --mode r2c --iters 5 --check
done
for n in 16384 65536 10000; do
  echo "== line f64 $n"; timeout 60 $K --line $n --batch $((2097152 * 16 / n)) --prec f64
done
echo "== line f64 8192"; timeout 60 $K --line 8192 --batch 4096 --prec f64
echo "== line f64 1024 forced two-level"; timeout 60 $K --line 1024 --batch 32768 --prec f64 --variant -2
echo "== line f64 1024"; timeout 60 $K --line 1024 --batch 32768 --prec f64
} > $OUT/two_level_times.txt 2>&1
grep -E "^==|PLAN|FFT|line|GB/s" $OUT/two_level_times.txt | cut -c1-160
