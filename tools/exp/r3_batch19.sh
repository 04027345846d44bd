#!/bin/bash
# round 3, GPU batch 19: every configuration on every pass (tests/test_gpu_variants.py), the tuner with per-pass configuration trials
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b19
mkdir -p $OUT
cd $R
K=$R/tools/kbench
timeout 1200 python -m pytest tests/test_gpu_variants.py tests/test_gpu_placement.py -x -q --durations=8 > $OUT/pytest_variants.txt 2>&1; tail -25 $OUT/pytest_variants.txt
{
for cfg in "1024 f64 2x4 c2c" "1024 f64 8x1 c2c" "2048 f32 2x4 c2c" "2048 f32 8x1 c2c" "1024 f64 2x2 c2c" "1024 f64 2x1 c2c" "1024 f64 2x4 r2c" "1024 f32 2x4 c2c"; do
  set -- $cfg
  echo "== $1^3 $2 $4 rank 0 of $3: tune-variants"
  timeout 200 $K --size $1 --prec $2 --mode $4 --iters 10 --ranks $3 --tune-variants
done
} > $OUT/tune_variants.txt 2>&1
grep -E "^==|TUNE|FFT|total" $OUT/tune_variants.txt | grep -v exchange | cut -c1-200
for i in 1 2; do timeout 600 python bench.py > $OUT/bench_$i.json 2> $OUT/bench_$i.err; python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["roofline"]["frac"], d["config"].get("placement",{}).get("trial_fft_ms_fwd_plus_inv"))
PY
done
