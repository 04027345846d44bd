#!/bin/bash
# round 3, GPU batch 12: packed real z passes of the mixed-radix lengths with the Hermitian split / merge in registers (45 of 74 kernels pairs):
# parity of every length, timings against profiles/r3_mixed_real_before_paired.txt, strided-read fp64 variants on 1024 threads
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b12
mkdir -p $OUT
cd $R
K=$R/tools/kbench
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -k "mixed or real or r2c or any_size or randomised" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
{
for sz in 512x512x576 512x512x1152 512x512x1200 512x512x2000 512x512x1000 512x512x800; do
  echo "== r2c fp64 $sz"; timeout 100 $K --size $sz --prec f64 --mode r2c --iters 10 --check
done
for sz in 512x512x2000 512x512x1000 512x512x1152 512x512x500; do
  echo "== r2c fp32 $sz"; timeout 100 $K --size $sz --prec f32 --mode r2c --iters 10 --check
done
echo "== r2c fp64 1000^3"; timeout 100 $K --size 1000 --prec f64 --mode r2c --iters 5 --check
echo "== r2c fp32 1000^3"; timeout 100 $K --size 1000 --prec f32 --mode r2c --iters 5 --check
} > $OUT/mixed_real.txt 2>&1
grep -E "^==|PLAN|z-FFT" $OUT/mixed_real.txt | cut -c1-150
