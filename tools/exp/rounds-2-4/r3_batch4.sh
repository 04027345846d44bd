#!/bin/bash
# round 3, GPU batch 4: placement tuner (dfft_tune_placement) -- tests, then bench.py lines from fresh processes with the
# tuner off / 4 / 8 backings per buffer
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b4
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_placement.py -x -q > $OUT/pytest_placement.txt 2>&1; tail -15 $OUT/pytest_placement.txt
for rep in 1 2 3; do
  for k in 0 4 8; do
    timeout 200 python bench.py --no-cpu-baseline --no-multi-rank-path --tune-placement $k > $OUT/bench_k${k}_r${rep}.json 2> $OUT/bench_k${k}_r${rep}.err
    python - <<P
import json
try:
    d = json.loads(open("$OUT/bench_k${k}_r${rep}.json").read().strip().splitlines()[-1])
    pp = d["config"]["per_pass"]
    print("tune=$k rep=$rep ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], {k: v["ms"] for k, v in pp.items() if "FFT" in k}, (d["config"].get("placement") or {}))
except Exception as e:
    print("tune=$k rep=$rep FAILED", e); print(open("$OUT/bench_k${k}_r${rep}.err").read()[-1500:])
P
  done
done
