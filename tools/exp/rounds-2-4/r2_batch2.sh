#!/bin/bash
# round 2, GPU batch 2: buffer placement x nontemporal policy, strided-read pass, fp32 NT forms, full-size 2048 runs
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b2
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
echo "=== A. placement x nontemporal, 1024^3 fp64 c2c (one slab, buffers at slot*(size+delta))"
NT="--opt variant_fz=3 --opt variant_fy=3 --opt variant_fx=3"
$K --size 1024 --prec f64 --iters 5 --label default-slab --slab
for perm in wiob iowb obwi boiw; do
  $K --size 1024 --prec f64 --iters 5 --label nt-all $NT --perm $perm
done
for d in 4096 65536 1048576 2105344 33562624 1073750016; do
  $K --size 1024 --prec f64 --iters 5 --label nt-all $NT --perm wiob --delta $d
done
$K --size 1024 --prec f64 --iters 5 --label nt-all-noslab $NT
$K --size 1024 --prec f64 --iters 5 --label nt-loads-yx --opt variant_fy=8 --opt variant_fx=8 --slab
$K --size 1024 --prec f64 --iters 5 --label nt-stores-yx --opt variant_fy=9 --opt variant_fx=9 --slab
$K --size 1024 --prec f64 --iters 5 --label nt-loads-yx --opt variant_fy=8 --opt variant_fx=8 --perm obwi
$K --size 1024 --prec f64 --iters 5 --label nt-stores-yx --opt variant_fy=9 --opt variant_fx=9 --perm obwi
$K --size 1024 --prec f64 --iters 5 --label nt-all-roof $NT --slab --opt debug_skip=1
$K --size 1024 --prec f64 --iters 5 --label nt-all-roof $NT --perm obwi --opt debug_skip=1
echo "=== B. multi-rank inverse x pass (strided read)"
for v in 1 7 10 11 8 3; do
  $K --size 1024 --prec f64 --iters 5 --label mirror-ix$v --opt mirror_inverse=1 --opt variant_ix=$v --slab
done
$K --size 1024 --prec f64 --iters 5 --label mirror-ix11 --opt mirror_inverse=1 --opt variant_ix=11 --perm obwi
$K --size 1024 --prec f64 --iters 5 --label mirror-ix11 --opt mirror_inverse=1 --opt variant_ix=11 --perm iowb
$K --size 1024 --prec f64 --iters 5 --label mirror-ix11-roof --opt mirror_inverse=1 --opt variant_ix=11 --slab --opt debug_skip=1
$K --size 1024 --prec f64 --iters 5 --label mirror-allnt --opt mirror_inverse=1 --opt variant_ix=11 --opt variant_iy=3 --opt variant_iz=3 $NT --opt pipeline_chunks=8 --slab
echo "=== C. fp32 1024 nontemporal forms"
$K --size 1024 --prec f32 --iters 5 --label base --slab
$K --size 1024 --prec f32 --iters 5 --label nt-all --opt variant_fz=13 --opt variant_fy=10 --opt variant_fx=10 --slab
$K --size 1024 --prec f32 --iters 5 --label nt-all --opt variant_fz=13 --opt variant_fy=10 --opt variant_fx=10 --perm obwi
$K --size 1024 --prec f32 --iters 5 --label nt-stores --opt variant_fz=7 --opt variant_fy=11 --opt variant_fx=11 --slab
$K --size 1024 --prec f32 --iters 5 --label nt-loads --opt variant_fz=7 --opt variant_fy=12 --opt variant_fx=12 --slab
$K --size 1024 --prec f32 --iters 5 --check --label pf2-z --opt variant_fz=7
echo "=== D. 2048^3 fp32 c2c, one GPU (in = back aliased)"
$K --size 2048 --prec f32 --iters 3 --check --label auto
$K --size 2048 --prec f32 --iters 3 --label z7-yx6 --opt variant_fz=7 --opt variant_fy=6 --opt variant_fx=6
$K --size 2048 --prec f32 --iters 3 --label z7-yx0 --opt variant_fz=7 --opt variant_fy=0 --opt variant_fx=0
$K --size 2048 --prec f32 --iters 3 --label z9-yx10 --opt variant_fz=9 --opt variant_fy=10 --opt variant_fx=10
$K --size 2048 --prec f32 --iters 3 --label z9-yx11 --opt variant_fz=9 --opt variant_fy=11 --opt variant_fx=11
$K --size 2048 --prec f32 --iters 3 --label z7-yx1 --opt variant_fz=7 --opt variant_fy=1 --opt variant_fx=1
$K --size 2048 --prec f32 --iters 3 --label roof-z7-yx0 --opt variant_fz=7 --opt variant_fy=0 --opt variant_fx=0 --opt debug_skip=1
$K --size 2048 --prec f32 --iters 3 --label mirror-z7-yx0 --opt variant_fz=7 --opt variant_fy=0 --opt variant_fx=0 --opt variant_ix=0 --opt variant_iy=0 --opt variant_iz=8 --opt mirror_inverse=1 --opt pipeline_chunks=8
echo "=== E. fp64, one 2048-point axis at 32 GiB per buffer"
for sz in 1024x1024x2048 1024x2048x1024 2048x1024x1024; do
  $K --size $sz --prec f64 --iters 3 --check --label auto
  $K --size $sz --prec f64 --iters 3 --label z6-yx0 --opt variant_fz=6 --opt variant_fy=0 --opt variant_fx=0
  $K --size $sz --prec f64 --iters 3 --label z6-yx3 --opt variant_fz=6 --opt variant_fy=3 --opt variant_fx=3
  $K --size $sz --prec f64 --iters 3 --label z5-yx5 --opt variant_fz=5 --opt variant_fy=5 --opt variant_fx=5
  $K --size $sz --prec f64 --iters 3 --label roof --opt debug_skip=1
done
echo "=== F. R2C placement sensitivity"
$K --size 1024 --prec f64 --mode r2c --iters 5 --label r2c --slab
$K --size 1024 --prec f64 --mode r2c --iters 5 --label r2c --perm obwi
$K --size 1024 --prec f32 --mode r2c --iters 5 --label r2c --slab
$K --size 1024 --prec f32 --mode r2c --iters 5 --label r2c --perm obwi
} > $OUT/kbench.txt 2>&1
grep -c PLAN $OUT/kbench.txt; tail -3 $OUT/kbench.txt
