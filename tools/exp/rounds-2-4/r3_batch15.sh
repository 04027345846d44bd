#!/bin/bash
# round 3, GPU batch 15: two-level lines with lanes over sub-lines on natural-line sides (parity again, times against batch 14),
# and one-launch Bluestein against the forced two-level form for lengths whose Bluestein inner transform has 4096 / 8192 points
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b15
mkdir -p $OUT
cd $R
K=$R/tools/kbench
timeout 1200 python -m pytest tests/test_gpu_two_level.py -x -q > $OUT/pytest_two_level.txt 2>&1; tail -5 $OUT/pytest_two_level.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab_sequences.py -x -q -k "any_length or any_size or bluestein or y_then_zx or long or partial" > $OUT/pytest_generic.txt 2>&1; tail -3 $OUT/pytest_generic.txt
{
for sz in 16384x128x128 128x16384x128 128x128x16384 10000x128x128 128x128x10000 64x64x65536; do
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
for p in f64 f32; do
for n in 1026 1500 2022 3000 3072 4000 4095; do
  echo "== line $p $n one launch (Bluestein)"; timeout 60 $K --line $n --batch $((2097152 * 16 / n)) --prec $p
  echo "== line $p $n two levels"; timeout 60 $K --line $n --batch $((2097152 * 16 / n)) --prec $p --variant -2
done
done
} > $OUT/two_level_times.txt 2>&1
grep -E "^==|PLAN|z-FFT|LINE" $OUT/two_level_times.txt | cut -c1-160
