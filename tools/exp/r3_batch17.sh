#!/bin/bash
# round 3, GPU batch 17: dfft_tune_variants with the workgroup-order trials: tuner tests, per-GPU plans of configs 4 / 5 before and after, the N = 1 bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b17
mkdir -p $OUT
cd $R
K=$R/tools/kbench
timeout 900 python -m pytest tests/test_gpu_placement.py -x -q > $OUT/pytest_placement.txt 2>&1; tail -5 $OUT/pytest_placement.txt
{
for cfg in "1024 f64 2x4" "1024 f64 8x1" "2048 f32 2x4" "2048 f32 8x1" "1024 f64 2x2" "1024 f64 2x1"; do
  set -- $cfg
  echo "== $1^3 $2 rank 0 of $3: rule-based | tune-variants"
  timeout 200 $K --size $1 --prec $2 --iters 10 --ranks $3 --tune-variants
done
echo "== 1024^3 f64 r2c rank 0 of 2x4"; timeout 200 $K --size 1024 --prec f64 --mode r2c --iters 10 --ranks 2x4 --tune-variants
} > $OUT/tune_variants.txt 2>&1
grep -E "^==|PLAN|TUNE|total" $OUT/tune_variants.txt | cut -c1-200
for i in 1 2; do timeout 600 python bench.py > $OUT/bench_$i.json 2> $OUT/bench_$i.err; python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
print("bench", d["ms_per_step"], d["roofline"]["frac"], d["config"].get("placement",{}).get("trial_fft_ms_fwd_plus_inv"))
PY
done
