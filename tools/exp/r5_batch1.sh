#!/bin/bash
# round 5, GPU batch 1: the one-shot relay on virtual ranks / MPI ranks / gloo ranks, bench.py launching its own ranks,
# and where the per-GPU passes of the 2x4 plans (R2C fp64, C2C fp64, C5 fp32) stand: rule-based, tuned, by pipeline depth, as copies
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b1
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_relay.py tests/test_gpu_cpp_shim.py tests/test_gpu_round3.py -m gpu -q -x -k "relay or bench_multi or shim" --durations=8 > $OUT/pytest_relay.txt 2>&1; tail -15 $OUT/pytest_relay.txt
K="$R/tools/kbench --size 1024 --prec f64 --iters 10 --lib-buffers --ranks 2x4 --rank 0"
{
echo "== R2C fp64 1024^3 rank 0 of 2x4"
$K --mode r2c
$K --mode r2c --tune-variants
for c in 1 2 8; do $K --mode r2c --opt pipeline_chunks=$c; done
$K --mode r2c --opt debug_skip=1
for v in 0 1 2 3; do $K --mode r2c --opt variant_ix=$v | grep -E "PLAN|x-FFT\^-1"; done
echo "== C2C fp64 1024^3 rank 0 of 2x4"
$K --mode c2c
$K --mode c2c --tune-variants
for c in 1 2 8; do $K --mode c2c --opt pipeline_chunks=$c; done
$K --mode c2c --opt debug_skip=1
echo "== C2C fp32 2048^3 rank 0 of 2x4"
K5="$R/tools/kbench --size 2048 --prec f32 --iters 5 --lib-buffers --ranks 2x4 --rank 0"
$K5 --mode c2c
$K5 --mode c2c --tune-variants
for c in 1 2 8; do $K5 --mode c2c --opt pipeline_chunks=$c; done
$K5 --mode c2c --opt debug_skip=1
} > $OUT/kbench_2x4.txt 2>&1
grep -E "PLAN|FFT|TUNE|==" $OUT/kbench_2x4.txt | cut -c1-200
