#!/bin/bash
# round 5, GPU batch 6: smoke(), bench.py --pmc 1 (HBM traffic measured in the run), an N = 8 line from 8 gloo ranks sharing the GPU (what the
# line looks like: direct + relayed run, transport counters; the times mean nothing there), the changed bench tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b6
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --pmc 1 --no-cpu-baseline --no-multi-rank-path --no-plain-leg > $OUT/bench_r5b_pmc_live.json 2> $OUT/pmc_live.err; tail -2 $OUT/pmc_live.err
python - <<'PY'
import json, os
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r5b6")
try:
    j = json.loads([l for l in open(os.path.join(d, "bench_r5b_pmc_live.json")) if l.startswith("{")][-1])
    r = j["roofline"]
    print("pmc live", j["ms_per_step"], r["frac"], r.get("traffic"), r.get("traffic_static"), r.get("traffic_source"), r.get("traffic_live"))
except Exception as e:
    print("unreadable", e)
PY
timeout 600 python bench.py --gpus 8 --backend gloo --size 256 --steps 3 --warmup 1 > $OUT/bench_r5_8ranks_one_gpu_gloo.json 2> $OUT/gloo8.err; tail -2 $OUT/gloo8.err
python - <<'PY'
import json, os
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r5b6")
try:
    j = json.loads([l for l in open(os.path.join(d, "bench_r5_8ranks_one_gpu_gloo.json")) if l.startswith("{")][-1]); c = j["config"]
    print("gloo8", j["n_gpus"], c["decomposition"], j["ms_per_step"], j.get("headline_is_best_of"), j.get("headline_candidates"), c["relay"].get("transport_counters"), c["relay"]["round_trip_rel_linf"], j["round_trip_rel_linf"])
except Exception as e:
    print("unreadable", e)
PY
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -k "bench" --durations=5 2>&1 | tail -8
