#!/bin/bash
# round 3, GPU batch 11: the shipped build (no experiments): mixed-radix real passes with the bounded-loads C2R (scratch removed at 24 points
# per thread) -- parity of every mixed real length, and a few timings; then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b11
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
for sz in 512x512x576 512x512x1152 512x512x1200 512x512x2000; do
  echo "== r2c fp64 $sz"; timeout 100 $K --size $sz --prec f64 --mode r2c --iters 10 --check
done
echo "== r2c fp32 512x512x2000"; timeout 100 $K --size 512x512x2000 --prec f32 --mode r2c --iters 10 --check
echo "== r2c fp64 1000^3"; timeout 100 $K --size 1000 --prec f64 --mode r2c --iters 5 --check
} > $OUT/mixed_real.txt 2>&1
grep -E "^==|PLAN|z-FFT" $OUT/mixed_real.txt | cut -c1-150
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
