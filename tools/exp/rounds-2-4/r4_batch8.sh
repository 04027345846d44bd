#!/bin/bash
# round 4, GPU batch 8: whole-line transposed stores at 1024 fp32 points (A/B), the C5 per-GPU kernels and 2048^3 on one GPU with the new
# defaults, parity of the new configurations, the bench line of N ranks sharing this GPU (relay leg included)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b8
mkdir -p $OUT
cd $R
K=$R/tools/kbench_exp
KS=$R/tools/kbench
S=""; C=""
for v in 6 9 5 7 11 3 2 10; do S="$S${S:+;}variant_iy=$v"; C="$C${C:+;}debug_skip=1,variant_iy=$v"; done
{
echo "== 1024^3 fp32, rank 0 of 2x4: inverse y pass on configuration 6 9 5 7 11 3 2 10 (set order), transforms"
timeout 200 $K --size 1024 --prec f32 --iters 10 --ranks 2x4 --sweep "$S" 2>&1 | grep -E "y-FFT\^-1"
echo "== as copies"
timeout 200 $K --size 1024 --prec f32 --iters 10 --ranks 2x4 --sweep "$C" 2>&1 | grep -E "y-FFT\^-1"
echo "== 1024^3 fp32 on one rank, mirrored inverse, 8 chunks: same configurations"
timeout 300 $K --size 1024 --prec f32 --iters 5 --opt mirror_inverse=1 --opt pipeline_chunks=8 --sweep "$S" 2>&1 | grep -E "y-FFT\^-1"
} > $OUT/r4_f32_1024_inverse_y_whole_lines.txt 2>&1
cat $OUT/r4_f32_1024_inverse_y_whole_lines.txt | cut -c1-100
{
echo "== 2048^3 fp32, rank 0 of 2x4 (shipped library): rule-based configurations, then dfft_tune_variants"
timeout 200 $KS --size 2048 --prec f32 --iters 5 --ranks 2x4 2>&1 | grep -E "^PLAN|FFT|total"
timeout 300 $KS --size 2048 --prec f32 --iters 5 --ranks 2x4 --tune-variants 2>&1 | grep -E "^PLAN|TUNE|FFT|total"
echo "== slab 8"
timeout 300 $KS --size 2048 --prec f32 --iters 5 --ranks 8x1 --tune-variants 2>&1 | grep -E "^PLAN|TUNE|FFT|total"
echo "== 1024^3 fp64, rank 0 of 2x4 and 8x1, tuned"
timeout 300 $KS --size 1024 --prec f64 --iters 10 --ranks 2x4 --tune-variants 2>&1 | grep -E "^PLAN|FFT|total"
timeout 300 $KS --size 1024 --prec f64 --iters 10 --ranks 8x1 --tune-variants 2>&1 | grep -E "^PLAN|FFT|total"
} > $OUT/r4_per_gpu_kernels_8gpu_plans.txt 2>&1
cut -c1-150 $OUT/r4_per_gpu_kernels_8gpu_plans.txt
timeout 600 python bench.py --size 2048 --precision float --no-cpu-baseline > $OUT/bench_r4_f32_2048.json 2> $OUT/bench_2048.err; tail -c 300 $OUT/bench_r4_f32_2048.json; tail -2 $OUT/bench_2048.err
timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_round3.py -m gpu -q -x --durations=6 -k "configuration or bench" > $OUT/r4_pytest_b8.txt 2>&1
tail -14 $OUT/r4_pytest_b8.txt
python - <<'PY'
import json, os
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4b8")
try:
    j = json.loads([l for l in open(os.path.join(d, "bench_r4_f32_2048.json")) if l.startswith("{")][-1])
    print("2048^3 fp32:", j["ms_per_step"], "ms", j["roofline"]["frac"], {k: v["ms"] for k, v in j["config"]["per_pass"].items() if "FFT" in k})
except Exception as e:
    print("unreadable", e)
PY
