#!/bin/bash
# round 4, GPU batch 14: the record lines on buffers from the library's probing allocator: execR2C / execC2R at 1024^3 (fp64, fp32),
# fp32 C2C at 1024^3, mixed radix 1000^3, the multi-rank path
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b14
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
for args in "--size 1024 --prec f64 --mode r2c" "--size 1024 --prec f32 --mode r2c" "--size 1024 --prec f32" "--size 1000 --prec f64" "--size 512 --prec f64" \
            "--size 1024 --prec f64 --opt mirror_inverse=1 --opt pipeline_chunks=8" "--size 1024 --prec f32 --opt mirror_inverse=1 --opt pipeline_chunks=8"; do
  timeout 200 $K $args --iters 8 --check --lib-buffers --tune-variants 2>&1 | grep -E "^PLAN|FFT|total" | cut -c1-170
done
} > $OUT/r4_phase_times.txt 2>&1
cat $OUT/r4_phase_times.txt
