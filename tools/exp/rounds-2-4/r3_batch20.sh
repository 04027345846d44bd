#!/bin/bash
# round 3, GPU batch 20: scalar-base address forms (global_load / global_store v_off, s[base:base+1]: one scalar 64-bit base per
# point, one 32-bit lane offset) against the per-point 64-bit vector addresses they replace (debug bit 1 = the old forms), each
# pair IN ONE PROCESS ON THE SAME BUFFERS (tools/kbench --sweep), twice over to see the noise
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b20
mkdir -p $OUT
cd $R
K=$R/tools/kbench
S="debug_skip=0;debug_skip=2;debug_skip=0;debug_skip=2"
run() { echo "== $1"; shift; timeout 120 $K "$@" --sweep "$S" 2>&1 | grep -E "^PLAN|FFT|total"; }
{
run "1024^3 fp64 c2c"                 --size 1024 --prec f64 --iters 5 --check
run "1024^3 fp64 c2c multi-rank path" --size 1024 --prec f64 --iters 5 --check --opt mirror_inverse=1 --opt pipeline_chunks=8
run "1024^3 fp32 c2c"                 --size 1024 --prec f32 --iters 8 --check
run "1024^3 fp32 c2c multi-rank path" --size 1024 --prec f32 --iters 8 --check --opt mirror_inverse=1 --opt pipeline_chunks=8
run "1024^3 fp64 rank 0 of 2x4"       --size 1024 --prec f64 --iters 10 --ranks 2x4
run "2048^3 fp32 rank 0 of 2x4"       --size 2048 --prec f32 --iters 5 --ranks 2x4
run "1024^3 fp64 r2c"                 --size 1024 --prec f64 --mode r2c --iters 8 --check
run "1024^3 fp32 r2c"                 --size 1024 --prec f32 --mode r2c --iters 8 --check
run "1000^3 fp64 c2c (mixed radix)"   --size 1000 --prec f64 --iters 5 --check
} > $OUT/r3_scalar_base_addresses.txt 2>&1
cat $OUT/r3_scalar_base_addresses.txt | cut -c1-150
