#!/bin/bash
# round 3, GPU batch 8 (f32 objects built with EXTRA=-DDFFT_EXPERIMENTS):
#  (a) tiled 2048-point fp32 passes: 64.32 (shipped) | 32.64 | 64.32 nontemporal | 32.64 nontemporal, y and x passes, on 2048 x 2048 x 512
#      (one process, shared buffers) and the same on the per-GPU plan of rank 0 of 2 x 4 at 2048^3
#  (b) fp64 R2C inverse x pass (513-wide rows): strided-read configuration with nontemporal loads + stores (1) | stores only (2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b8
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
echo "== 2048 x 2048 x 512 fp32 c2c, y and x passes: variant 6 (64.32) | 10 (32.64) | 11 (64.32 nt) | 12 (32.64 nt) | 6"
timeout 300 $K --size 2048x2048x512 --prec f32 --iters 5 --sweep "variant_fy=6,variant_fx=6;variant_fy=10,variant_fx=10;variant_fy=11,variant_fx=11;variant_fy=12,variant_fx=12;variant_fy=6,variant_fx=6"
for v in 10 12; do echo "== check variant $v"; timeout 100 $K --size 2048x2048x64 --prec f32 --iters 2 --check --opt variant_fy=$v --opt variant_fx=$v --opt variant_ix=$v --opt variant_iy=$v | grep PLAN; done
echo "== same, mirrored inverse (x^-1, y^-1 of the multi-rank order): 6 | 10 | 12"
timeout 300 $K --size 2048x2048x512 --prec f32 --iters 5 --opt mirror_inverse=1 --sweep "variant_ix=6,variant_iy=6;variant_ix=10,variant_iy=10;variant_ix=12,variant_iy=12;variant_ix=6,variant_iy=6"
echo "== 2048^3 fp32, rank 0 of 2x4 (exchange stubbed): 6 | 10 | 12 on all four tiled passes"
for v in 6 10 12; do timeout 100 $K --size 2048 --prec f32 --iters 10 --ranks 2x4 --opt variant_fy=$v --opt variant_fx=$v --opt variant_ix=$v --opt variant_iy=$v; done
} > $OUT/f32_2048_tiled.txt 2>&1
grep -E "^==|PLAN|y-FFT|x-FFT|total" $OUT/f32_2048_tiled.txt | cut -c1-170
{
echo "== fp64 R2C + C2R 1024^3, tuned buffers: variant_ix 1 (nt loads + stores) | 2 (nt stores only) | 0 (8 lines, no hints) | 1 | 2"
timeout 300 $K --size 1024 --prec f64 --mode r2c --iters 10 --tune 4 --sweep "variant_ix=1;variant_ix=2;variant_ix=0;variant_ix=1;variant_ix=2"
echo "== fp64 C2C 1024^3 mirrored inverse: variant_ix 1 | 2 | 1 | 2"
timeout 300 $K --size 1024 --prec f64 --iters 10 --opt mirror_inverse=1 --sweep "variant_ix=1;variant_ix=2;variant_ix=1;variant_ix=2"
} > $OUT/xinv_nt.txt 2>&1
grep -E "^==|PLAN|x-FFT\^-1" $OUT/xinv_nt.txt | cut -c1-170
