#!/bin/bash
# round 4, GPU batch 1.  Needs the A/B build beside the shipped one (CPU, beforehand):
#   make -j8 -C distributedfft_amd/csrc exp && make -C tools kbench_exp
# 1. PERSIST = 3 (stores of a tile fused with the loads of the next) on the strided read of the API layout (multi-rank x^-1)
# 2. row-aligned LOAD windows for the inverse x pass of R2C plans (option shift_load), 513- and 129-wide rows
# 3. fixed virtual-memory recipes against the placement search and plain hipMalloc (which recipe to make the default)
# 4. baselines of the fp32 2048-point tiled passes (rank 0 of 2x4 at 2048^3)
# 5. the BASELINE-config parity tests that round 3 never ran (C5 full size, C5-shaped 2x4 every point) + the 1-rank worker tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b1
mkdir -p $OUT
cd $R
K=$R/tools/kbench_exp
KS=$R/tools/kbench
S="variant_ix=1;variant_ix=8;variant_ix=9;variant_ix=1"          # table store (P1 > 1)
S1="variant_ix=1;variant_ix=10;variant_ix=11;variant_ix=1"       # one-block store (P1 = 1: one rank)
{
echo "== host: $(nproc) cores, $(free -g | awk '/Mem:/{print $2" GiB RAM, "$7" GiB available"}')"
echo "== 1024^3 fp64 multi-rank path: x^-1 plain (1) | PERSIST 3 + hints (10) | PERSIST 3 (11) | plain"
timeout 200 $K --size 1024 --prec f64 --iters 5 --check --opt mirror_inverse=1 --opt pipeline_chunks=8 --sweep "$S1" 2>&1 | grep -E "^PLAN|FFT|total"
echo "== rank 0 of 2x4"
timeout 100 $K --size 1024 --prec f64 --iters 10 --ranks 2x4 --sweep "$S" 2>&1 | grep -E "^PLAN|FFT|total"
echo "== rank 0 of 8x1"
timeout 100 $K --size 1024 --prec f64 --iters 10 --ranks 8x1 --sweep "$S" 2>&1 | grep -E "^PLAN|FFT|total"
} > $OUT/r4_persist3.txt 2>&1
cut -c1-170 $OUT/r4_persist3.txt
{
echo "== 1024^3 fp64 R2C + C2R on one rank: shift_load 0 | 1 | 0 | 1 + strided-read configuration (2)"
timeout 200 $K --size 1024 --prec f64 --mode r2c --iters 5 --check --sweep "shift_load=0;shift_load=1;shift_load=0;shift_load=1,variant_ix=2" 2>&1 | grep -E "^PLAN|FFT|total"
echo "== rank 0 of 2x4 (129-wide rows)"
timeout 100 $K --size 1024 --prec f64 --mode r2c --iters 10 --ranks 2x4 --sweep "shift_load=0;shift_load=1;shift_load=0;shift_load=1,variant_ix=2" 2>&1 | grep -E "^PLAN|FFT|total"
} > $OUT/r4_shift_load.txt 2>&1
cut -c1-170 $OUT/r4_shift_load.txt
{
for rep in 1 2; do
echo "== 1024^3 fp64 C2C one rank, fresh processes: plain hipMalloc | vmm 64 | vmm 2 | vmm 1024 | vmm 256 (rep $rep)"
for v in 0 64 2 1024 256; do
  if [ $v = 0 ]; then timeout 100 $KS --size 1024 --prec f64 --iters 10 2>&1 | grep -E "^PLAN|total";
  else timeout 100 $KS --size 1024 --prec f64 --iters 10 --vmm $v 2>&1 | grep -E "^PLAN|total"; fi
done
done
echo "== placement search (6 backings per buffer)"
timeout 200 $KS --size 1024 --prec f64 --iters 10 --tune 6 2>&1 | grep -E "^PLAN|TUNE|FFT|total"
} > $OUT/r4_fixed_recipes.txt 2>&1
cut -c1-170 $OUT/r4_fixed_recipes.txt
{
echo "== 2048^3 fp32, rank 0 of 2x4: rule-based configurations, then dfft_tune_variants"
timeout 200 $KS --size 2048 --prec f32 --iters 5 --ranks 2x4 2>&1 | grep -E "^PLAN|FFT|total"
timeout 300 $KS --size 2048 --prec f32 --iters 5 --ranks 2x4 --tune-variants 2>&1 | grep -E "^PLAN|TUNE|FFT|total"
} > $OUT/r4_f32_2048_baseline.txt 2>&1
cut -c1-170 $OUT/r4_f32_2048_baseline.txt
timeout 1500 python -m pytest tests/test_gpu_round3.py tests/test_gpu_multi_device.py -m gpu -q -x --durations=8 \
  -k "c5 or one_rank" > $OUT/r4_pytest_c5.txt 2>&1
tail -25 $OUT/r4_pytest_c5.txt
