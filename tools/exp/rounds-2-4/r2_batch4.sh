#!/bin/bash
# round 2, GPU batch 4: plan timings with buffers from the virtual-memory API in physical chunks (optionally shuffled)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b4
mkdir -p $OUT
cd $R
K=$R/tools/kbench
NT="--opt variant_fz=3 --opt variant_fy=3 --opt variant_fx=3"
{
for rep in 1 2; do
for mode in "--vmm 2" "--vmm 2 --shuffle" "--vmm 64" "--vmm 64 --shuffle" "--vmm 1024" "--vmm 1024 --shuffle" "--vmm 16384"; do
  $K --size 1024 --prec f64 --iters 4 --label rep$rep-default $mode
  $K --size 1024 --prec f64 --iters 4 --label rep$rep-ntall $NT $mode
done
done
$K --size 1024 --prec f64 --iters 4 --check --label mirror --opt mirror_inverse=1 --opt variant_ix=11 --opt pipeline_chunks=8 --vmm 2 --shuffle
$K --size 1024 --prec f64 --iters 4 --label mirror --opt mirror_inverse=1 --opt variant_ix=11 --opt pipeline_chunks=8 --vmm 64 --shuffle
$K --size 1024 --prec f64 --iters 4 --label mirror --opt mirror_inverse=1 --opt variant_ix=1 --opt pipeline_chunks=8 --vmm 2 --shuffle
$K --size 1024 --prec f32 --iters 4 --label f32 --vmm 2 --shuffle
$K --size 1024 --prec f32 --iters 4 --label f32-nt --opt variant_fz=13 --opt variant_fy=10 --opt variant_fx=10 --vmm 2 --shuffle
$K --size 1024 --prec f64 --mode r2c --iters 4 --label r2c --vmm 2 --shuffle
$K --size 1024 --prec f32 --mode r2c --iters 4 --label r2c --vmm 2 --shuffle
$K --size 2048 --prec f32 --iters 3 --label z7-yx6 --opt variant_fz=7 --opt variant_fy=6 --opt variant_fx=6 --vmm 64 --shuffle
} > $OUT/kbench.txt 2>&1
grep -c PLAN $OUT/kbench.txt; tail -3 $OUT/kbench.txt
