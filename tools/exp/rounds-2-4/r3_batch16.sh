#!/bin/bash
# round 3, GPU batch 16: workgroup -> tile orders (a_fastest + 2 * xcd_swizzle) of every pass on the per-GPU plans of configs 4 and 5
# (rank 0 of 2x4, exchange stubbed), which dfft_tune_variants does not try yet
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b16
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
for cfg in "2048 f32" "1024 f64"; do
  set -- $cfg
  echo "== $1^3 $2 rank 0 of 2x4, defaults"; timeout 100 $K --size $1 --prec $2 --iters 8 --ranks 2x4 | grep -E "FFT|total"
  for pass in fz fy fx ix iy iz; do
    case $pass in fz) pat="z-FFT  ";; fy) pat="y-FFT  ";; fx) pat="x-FFT  ";; ix) pat="x-FFT\^-1";; iy) pat="y-FFT\^-1";; iz) pat="z-FFT\^-1";; esac
    for o in 0 1 2 3; do
      echo -n "order_$pass=$o: "; timeout 100 $K --size $1 --prec $2 --iters 8 --ranks 2x4 --opt order_$pass=$o | grep -E "$pat" | head -1
    done
  done
done
} > $OUT/orders.txt 2>&1
cat $OUT/orders.txt | cut -c1-120
