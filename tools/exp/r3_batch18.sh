#!/bin/bash
# round 3, GPU batch 18: every kernel configuration (role variant) of every pass on the per-GPU plans of configs 4 and 5 (rank 0 of 2x4,
# exchange stubbed): is a rule of dfft_init wrong for these plans?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b18
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
for cfg in "1024 f64 0,1,2,3" "2048 f32 0,4,5,6,9" "1024 f32 0,4,5,6,9"; do
  set -- $cfg
  echo "== $1^3 $2 rank 0 of 2x4, rule-based"; timeout 100 $K --size $1 --prec $2 --iters 8 --ranks 2x4 | grep -E "FFT|total"
  for pass in fz fy fx ix iy iz; do
    case $pass in fz) pat="z-FFT  ";; fy) pat="y-FFT  ";; fx) pat="x-FFT  ";; ix) pat="x-FFT\^-1";; iy) pat="y-FFT\^-1";; iz) pat="z-FFT\^-1";; esac
    for v in $(echo $3 | tr , ' '); do
      echo -n "variant_$pass=$v: "; timeout 100 $K --size $1 --prec $2 --iters 8 --ranks 2x4 --opt variant_$pass=$v | grep -E "$pat" | head -1
    done
  done
done
} > $OUT/variants.txt 2>&1
cat $OUT/variants.txt | cut -c1-120
