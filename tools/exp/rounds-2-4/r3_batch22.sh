#!/bin/bash
# round 3, GPU batch 22: the scalar-base form of the transposed-tile store (z passes of one-rank and chunked plans) against the old
# forms (debug bit 1 switches ALL scalar-base forms off), pairs in one process on the same buffers
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b22
mkdir -p $OUT
cd $R
K=$R/tools/kbench
S="debug_skip=0;debug_skip=2;debug_skip=0;debug_skip=2"
run() { echo "== $1"; shift; timeout 120 $K "$@" --sweep "$S" 2>&1 | grep -E "^PLAN|FFT|total"; }
{
run "1024^3 fp32 c2c"                 --size 1024 --prec f32 --iters 8 --check
run "1024^3 fp64 c2c"                 --size 1024 --prec f64 --iters 5 --check
run "1024^3 fp32 c2c multi-rank path" --size 1024 --prec f32 --iters 8 --check --opt mirror_inverse=1 --opt pipeline_chunks=8
run "1024^3 fp64 c2c multi-rank path" --size 1024 --prec f64 --iters 5 --check --opt mirror_inverse=1 --opt pipeline_chunks=8
run "512^3 fp64 c2c"                  --size 512 --prec f64 --iters 10 --check
run "2048x1024x1024 fp32 c2c"         --size 2048x1024x1024 --prec f32 --iters 5 --check
} > $OUT/r3_scalar_base_transposed_store.txt 2>&1
cat $OUT/r3_scalar_base_transposed_store.txt | cut -c1-150
