#!/bin/bash
# round 4, GPU batch 13: the default backing with its placement probe (dev_alloc_default): bench lines without the plan-level search from
# fresh processes, a 64 GiB allocation coming and going before each
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b13
mkdir -p $OUT
cd $R
for i in 1 2 3 4; do
timeout 200 tools/kbench --size 2048 --prec f32 --iters 1 > /dev/null 2>&1
timeout 300 python bench.py --tune-placement 0 --no-cpu-baseline --no-multi-rank-path --no-plain-leg > $OUT/bench_r4_c_probe_$i.json 2> $OUT/bench_p$i.err; tail -1 $OUT/bench_p$i.err | cut -c1-100
done
DFFT_PLACEMENT_TRIES=1 timeout 300 python bench.py --tune-placement 0 --no-cpu-baseline --no-multi-rank-path --no-plain-leg > $OUT/bench_r4_c_noprobe.json 2> $OUT/bench_np.err
python - <<'PY'
import json, os, glob
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4b13")
for f in sorted(glob.glob(os.path.join(d, "bench_r4_c*.json"))):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        c = j["config"]
        print(os.path.basename(f), "ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"], "alloc s", (c.get("placement") or {}).get("alloc_seconds_out_and_back"),
              {k: v["ms"] for k, v in c["per_pass"].items() if "FFT" in k}, "variants first/last", (c.get("variants") or {}).get("trial_fft_ms", [None])[0], (c.get("variants") or {}).get("trial_fft_ms", [None])[-1])
    except Exception as e:
        print(f, "unreadable:", e)
PY
