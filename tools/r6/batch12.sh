#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 400 python bench.py --no-cpu-baseline > $O/bench_r6b_$i.json 2> $O/bench_r6b_$i.err
  python - $O/bench_r6b_$i.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], j["ms_per_step"], j["roofline"]["frac"], j["config"].get("plain_buffers_ms_per_step"), j["roofline"].get("traffic_static"))
PY
done
timeout 900 python bench.py --no-cpu-baseline --no-multi-rank-path --no-plain-leg --pmc 1 > $O/bench_r6b_pmc_live.json 2> $O/bench_r6b_pmc_live.err
python - $O/bench_r6b_pmc_live.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = j["roofline"]
print("live PMC:", j["ms_per_step"], r["frac"], r.get("traffic"), r.get("traffic_static"), r.get("traffic_source"))
PY
