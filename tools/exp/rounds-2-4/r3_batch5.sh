#!/bin/bash
# round 3, GPU batch 5: (a) the C++ `pencil` / `slab` executables + shim timer CSV, placement tests
#                       (b) placement tuner with all candidates of a buffer held at once: bench.py from fresh processes, 0 / 4 / 8 / 12 backings
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b5
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_cpp_drivers.py tests/test_gpu_cpp_shim.py tests/test_gpu_placement.py -x -q > $OUT/pytest_drivers.txt 2>&1; tail -25 $OUT/pytest_drivers.txt
for rep in 1 2 3; do
  for k in 0 4 8 12; do
    timeout 200 python bench.py --no-cpu-baseline --no-multi-rank-path --tune-placement $k > $OUT/bench_k${k}_r${rep}.json 2> $OUT/bench_k${k}_r${rep}.err
    python - <<P
import json
try:
    d = json.loads(open("$OUT/bench_k${k}_r${rep}.json").read().strip().splitlines()[-1])
    pp = d["config"]["per_pass"]
    pl = d["config"].get("placement") or {}
    print("tune=$k rep=$rep ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], {k: v["ms"] for k, v in pp.items() if "FFT" in k}, pl.get("trial_fft_ms_fwd_plus_inv"), pl.get("seconds"))
except Exception as e:
    print("tune=$k rep=$rep FAILED", e); print(open("$OUT/bench_k${k}_r${rep}.err").read()[-1500:])
P
  done
done
