#!/bin/bash
# round 3, GPU batch 10: fp32 inverse y pass (tiled load, transposed-tile store) with the point-fastest store mapping (role 10) against the
# line-fastest tiled configuration (6); whole GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b10
mkdir -p $OUT
cd $R
K=$R/tools/kbench
{
echo "== c2c fp32 1024, multi-rank path (mirrored inverse, 8 chunks): variant_iy 10 (rule) | 6 | 10 | 6"
timeout 200 $K --size 1024 --prec f32 --iters 10 --opt mirror_inverse=1 --opt pipeline_chunks=8 --sweep "variant_iy=10;variant_iy=6;variant_iy=10;variant_iy=6"
echo "== check"; timeout 100 $K --size 1024 --prec f32 --iters 2 --check --opt mirror_inverse=1 --opt pipeline_chunks=8 | grep PLAN
echo "== c2c fp32 2048 x 2048 x 512, mirrored inverse: variant_iy 10 | 6 | 11 (10 + nontemporal) | 10"
timeout 300 $K --size 2048x2048x512 --prec f32 --iters 5 --opt mirror_inverse=1 --sweep "variant_iy=10;variant_iy=6;variant_iy=11;variant_iy=10"
echo "== 2048^3 fp32, rank 0 of 2x4: rule | variant_iy=6 | rule + tune-variants"
timeout 100 $K --size 2048 --prec f32 --iters 10 --ranks 2x4
timeout 100 $K --size 2048 --prec f32 --iters 10 --ranks 2x4 --opt variant_iy=6
timeout 100 $K --size 2048 --prec f32 --iters 10 --ranks 2x4 --tune-variants
echo "== 2048^3 fp32, rank 0 of 8x1: rule + tune-variants"
timeout 100 $K --size 2048 --prec f32 --iters 10 --ranks 8x1 --tune-variants
echo "== r2c fp32 1024^3: rule | variant_iy=6"
timeout 100 $K --size 1024 --prec f32 --mode r2c --iters 10 --check
timeout 100 $K --size 1024 --prec f32 --mode r2c --iters 10 --opt variant_iy=6
} > $OUT/pfstore.txt 2>&1
grep -E "^==|PLAN|y-FFT\^-1|TUNE|total" $OUT/pfstore.txt | cut -c1-170
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
