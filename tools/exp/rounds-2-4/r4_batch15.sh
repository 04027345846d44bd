#!/bin/bash
# round 4, GPU batch 15: rocprofv3 kernel summaries of the per-GPU plans (rank 0 of 2x4, exchange stubbed): C5 2048^3 fp32 and C4 1024^3 fp64
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b15
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for cfg in "f32_2048 --size 2048 --prec f32" "f64_1024 --size 1024 --prec f64"; do
  set -- $cfg
  tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o k -- $R/tools/kbench "$@" --ranks 2x4 --iters 20 > $OUT/kbench_$tag.txt 2>&1
  find $OUT/prof_$tag -name "*kernel_stats.csv" -exec cp {} $OUT/r4_rank0_2x4_${tag}_kernel_stats.csv \;
  rm -rf $OUT/prof_$tag
  grep -E "^PLAN|FFT|total" $OUT/kbench_$tag.txt | cut -c1-140
  head -12 $OUT/r4_rank0_2x4_${tag}_kernel_stats.csv | cut -c1-220
done
