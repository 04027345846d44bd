#!/bin/bash
# round 3, GPU batch 7:
#  (a) real z passes with nontemporal loads / stores (real_variant 5) against the default, 1024^3 R2C both precisions, tuned buffers
#  (b) y / x passes of the single-GPU plan on tuned (virtual-memory) buffers: nontemporal configurations and workgroup orders
#  (c) LDS bank conflicts of the 2048-point fp32 natural-line kernels (pad once per 64 points for the radix-64 scatter)
#  (d) per-GPU kernels of the 8-GPU plans
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b7
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
for p in f64 f32; do
  echo "== R2C + C2R 1024^3 $p, tuned buffers: real_variant 0 | 5 (nontemporal) | 0 | 5"
  timeout 200 $K --size 1024 --prec $p --mode r2c --iters 10 --tune 4 --sweep "real_variant=0;real_variant=5;real_variant=0;real_variant=5"
done
echo "== check real_variant=5"; for p in f64 f32; do timeout 60 $K --size 256x256x1024 --prec $p --mode r2c --iters 2 --check --opt real_variant=5 | grep PLAN;  timeout 60 $K --size 256x256x2048 --prec $p --mode r2c --iters 2 --check --opt real_variant=5 | grep PLAN; done
} > $OUT/real_nt.txt 2>&1
grep -E "^==|PLAN|z-FFT|TUNE" $OUT/real_nt.txt | cut -c1-170
{
echo "== C2C 1024^3 f64, tuned buffers: default | y nontemporal | x nontemporal | both | orders of y (0..3) | orders of x (0..3) | default"
timeout 400 $K --size 1024 --prec f64 --iters 8 --tune 4 --sweep "variant_fy=0;variant_fy=3;variant_fx=3;variant_fy=3,variant_fx=3;order_fy=0;order_fy=1;order_fy=2;order_fy=3;order_fx=0;order_fx=1;order_fx=2;order_fx=3;variant_fy=0"
} > $OUT/yx_sweep.txt 2>&1
grep -E "^==|PLAN|y-FFT |x-FFT |TUNE" $OUT/yx_sweep.txt | cut -c1-170
bash tools/pmc_quick.sh r3_f32_2048 -- $K --size 2048x512x2048 --prec f32 --iters 1 > /dev/null 2>&1
python tools/pmc_summary.py $R/gpurun_out/pmcq_r3_f32_2048 fft_ > $OUT/r3_pmc_f32_2048.txt 2>&1
grep -E "dispatch|LDS_BANK|LDS_IDX" $OUT/r3_pmc_f32_2048.txt | head -40
rm -rf $R/gpurun_out/pmcq_r3_f32_2048
