#!/bin/bash
# round 3, GPU batch 21: as batch 20 (scalar-base address forms against the per-point 64-bit vector addresses, debug bit 1, each
# pair in one process on the same buffers) after the wave-uniform table paths got scalar tile coordinates too: the plans with
# segmented sides (multi-rank path, rank 0 of the 8-GPU grids), then dfft_tune_variants with its address-form trial
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b21
mkdir -p $OUT
cd $R
K=$R/tools/kbench
S="debug_skip=0;debug_skip=2;debug_skip=0;debug_skip=2"
run() { echo "== $1"; shift; timeout 120 $K "$@" --sweep "$S" 2>&1 | grep -E "^PLAN|FFT|total"; }
{
run "1024^3 fp64 c2c multi-rank path" --size 1024 --prec f64 --iters 5 --check --opt mirror_inverse=1 --opt pipeline_chunks=8
run "1024^3 fp32 c2c multi-rank path" --size 1024 --prec f32 --iters 8 --check --opt mirror_inverse=1 --opt pipeline_chunks=8
run "1024^3 fp64 rank 0 of 2x4"       --size 1024 --prec f64 --iters 10 --ranks 2x4
run "1024^3 fp64 rank 0 of 8x1"       --size 1024 --prec f64 --iters 10 --ranks 8x1
run "1024^3 fp64 r2c rank 0 of 2x4"   --size 1024 --prec f64 --mode r2c --iters 10 --ranks 2x4
run "2048^3 fp32 rank 0 of 2x4"       --size 2048 --prec f32 --iters 5 --ranks 2x4
run "1024^3 fp32 rank 0 of 2x4"       --size 1024 --prec f32 --iters 10 --ranks 2x4
run "1024^3 fp64 c2c"                 --size 1024 --prec f64 --iters 5 --check
S="point_tables=1;point_tables=2;point_tables=1;point_tables=2"
echo "#### single-segment sides through the (now scalar) wave-uniform table path (point_tables=2) against their closed forms (1)"
run "1024^3 fp32 c2c, tables for every tiled side" --size 1024 --prec f32 --iters 8 --check
run "1024^3 fp64 c2c, tables for every tiled side" --size 1024 --prec f64 --iters 5 --check
echo "== dfft_tune_variants (trial times: as built, 4 orders, chosen orders, configurations, address forms, final)"
for cfg in "1024 f64" "2048 f32"; do set -- $cfg; timeout 200 $K --size $1 --prec $2 --iters 5 --ranks 2x4 --tune-variants 2>&1 | grep -E "^PLAN|TUNE|tune|FFT|total"; done
timeout 100 $K --size 1024 --prec f64 --iters 5 --tune 4 --check 2>&1 | grep -E "^PLAN|TUNE|tune|FFT|total"
} > $OUT/r3_scalar_base_addresses_tables.txt 2>&1
cat $OUT/r3_scalar_base_addresses_tables.txt | cut -c1-170
