#!/bin/bash
# round 3, GPU batch 9: dfft_tune_variants (streaming siblings of the y / x passes chosen by measurement), strided-read role only on aligned rows
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b9
mkdir -p $OUT
cd $R
K=$R/tools/kbench
timeout 900 python -m pytest tests/test_gpu_placement.py tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q > $OUT/pytest.txt 2>&1; tail -6 $OUT/pytest.txt
{
echo "== R2C + C2R 1024^3 f64 (x^-1 now the default configuration on 513-wide rows), plain | tuned"
timeout 100 $K --size 1024 --prec f64 --mode r2c --iters 10 --check
timeout 100 $K --size 1024 --prec f64 --mode r2c --iters 10 --check --tune 4
echo "== R2C + C2R 1024^3 f32, tuned"
timeout 100 $K --size 1024 --prec f32 --mode r2c --iters 10 --check --tune 4
echo "== C2C 1024^3 f32, tuned (variants 9 tried by the tuner)"
timeout 100 $K --size 1024 --prec f32 --iters 10 --check --tune 4
echo "== 2048^3 fp32, rank 0 of 2x4, rule-based | tune-variants"
timeout 100 $K --size 2048 --prec f32 --iters 10 --ranks 2x4
timeout 100 $K --size 2048 --prec f32 --iters 10 --ranks 2x4 --tune-variants
echo "== 2048^3 fp32, rank 0 of 8x1, rule-based | tune-variants"
timeout 100 $K --size 2048 --prec f32 --iters 10 --ranks 8x1
timeout 100 $K --size 2048 --prec f32 --iters 10 --ranks 8x1 --tune-variants
echo "== 1024^3 fp64, rank 0 of 2x4 | tune-variants;  8x1 | tune-variants"
timeout 100 $K --size 1024 --prec f64 --iters 20 --ranks 2x4
timeout 100 $K --size 1024 --prec f64 --iters 20 --ranks 2x4 --tune-variants
timeout 100 $K --size 1024 --prec f64 --iters 20 --ranks 8x1
timeout 100 $K --size 1024 --prec f64 --iters 20 --ranks 8x1 --tune-variants
echo "== 2048^3 fp32 one GPU, rule-based | tune-variants"
timeout 200 $K --size 2048 --prec f32 --iters 3
timeout 200 $K --size 2048 --prec f32 --iters 3 --tune-variants
} > $OUT/variants.txt 2>&1
grep -E "^==|PLAN|FFT|TUNE|total" $OUT/variants.txt | cut -c1-170
timeout 400 python bench.py --size 2048 --precision float --steps 5 --warmup 2 --no-cpu-baseline --no-multi-rank-path > $OUT/bench_f32_2048.json 2> $OUT/bench_f32_2048.err; tail -c 1200 $OUT/bench_f32_2048.json; tail -2 $OUT/bench_f32_2048.err
