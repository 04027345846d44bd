#!/bin/bash
# round 5, GPU batch 5: the allocator with a learned yardstick (first large buffer built, later ones judged against it): bench lines from
# three fresh processes, the placement tests, and the part of the GPU suite the checkpoint run did not reach
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b5
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-multi-rank-path --no-plain-leg > $OUT/bench_r5b_$i.json 2>> $OUT/bench.err; done
python - <<'PY'
import json, os, glob
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r5b5")
for f in sorted(glob.glob(os.path.join(d, "bench_r5b_*.json"))):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1]); c = j["config"]; pl = c["placement"]
        print(os.path.basename(f), j["ms_per_step"], j["roofline"]["frac"], pl.get("alloc_seconds_out_and_back"),
              {k: (pl[k]["kept"][:28], pl[k]["probe_TBps"], pl[k]["good_threshold_TBps"], pl[k]["seconds"]) for k in ("out", "back") if k in pl},
              {k: v["ms"] for k, v in c["per_pass"].items() if "FFT" in k})
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout 1500 python -m pytest tests/test_gpu_placement.py tests/test_gpu_rccl_one_rank.py tests/test_gpu_relay.py tests/test_gpu_round3.py tests/test_gpu_slab_sequences.py tests/test_gpu_spectral.py tests/test_gpu_transport_probe.py tests/test_gpu_two_level.py tests/test_gpu_variants.py -x -q -m gpu --durations=8 > $OUT/pytest_rest.txt 2>&1; tail -16 $OUT/pytest_rest.txt
