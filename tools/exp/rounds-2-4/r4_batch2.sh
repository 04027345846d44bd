#!/bin/bash
# round 4, GPU batch 2 (A/B build: make -j8 -C distributedfft_amd/csrc exp && make -C tools kbench_exp)
# 1. fp32 2048-point tiled passes of C5 (rank 0 of 2x4 at 2048^3): every configuration number on the four tiled passes, as transforms and
#    as copies with the same access pattern (debug_skip = 1: the pattern's own floor), incl. the new two-workgroups-per-CU candidates 12-15
# 2. parity: relay on virtual ranks, every configuration incl. the fused strided read (8), placement / drivers / shim / cli after the
#    allocator changes
# 3. bench.py: default line (tuner + plain-buffer leg + MPI CPU baseline) and --tune-placement 0 (default backing, no search)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b2
mkdir -p $OUT
cd $R
K=$R/tools/kbench_exp
sets() { local pre=$1; local s=""; for v in 6 9 5 4 0 12 13 14 15; do s="$s${s:+;}${pre}variant_fy=$v,variant_fx=$v,variant_ix=$v,variant_iy=$v"; done; echo "$s"; }
{
echo "== 2048^3 fp32, rank 0 of 2x4: all four tiled passes on configuration v = 6 9 5 4 0 12 13 14 15 (set order)"
timeout 300 $K --size 2048 --prec f32 --iters 5 --ranks 2x4 --sweep "$(sets '')" 2>&1 | grep -E "^PLAN|FFT|total"
echo "== the same as copies (debug_skip = 1): the access pattern's own time per configuration"
timeout 300 $K --size 2048 --prec f32 --iters 5 --ranks 2x4 --sweep "$(sets 'debug_skip=1,')" 2>&1 | grep -E "^PLAN|FFT|total"
} > $OUT/r4_f32_2048_tiled_candidates.txt 2>&1
grep -E "^PLAN|y-FFT|x-FFT" $OUT/r4_f32_2048_tiled_candidates.txt | cut -c1-150
timeout 1500 python -m pytest tests/test_gpu_relay.py tests/test_gpu_variants.py tests/test_gpu_placement.py tests/test_gpu_cpp_drivers.py \
  tests/test_gpu_cpp_shim.py tests/test_gpu_cli.py -m gpu -q -x --durations=8 > $OUT/r4_pytest_b2.txt 2>&1
tail -22 $OUT/r4_pytest_b2.txt
timeout 400 python bench.py > $OUT/bench_r4_a.json 2> $OUT/bench_a.err; tail -c 600 $OUT/bench_r4_a.json; echo; tail -3 $OUT/bench_a.err
timeout 300 python bench.py --tune-placement 0 --no-cpu-baseline --no-multi-rank-path > $OUT/bench_r4_a_notuner.json 2> $OUT/bench_b.err; tail -c 300 $OUT/bench_r4_a_notuner.json; echo; tail -3 $OUT/bench_b.err
python - <<'PY'
import json, os
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4b2")
for f in ("bench_r4_a.json", "bench_r4_a_notuner.json"):
    try:
        j = json.loads([l for l in open(os.path.join(d, f)) if l.startswith("{")][-1])
        c = j["config"]
        print(f, "ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"], "plain", c.get("plain_buffers_ms_per_step"),
              "placement s", (c.get("placement") or {}).get("seconds"), "cpu", (j.get("cpu_baseline") or {}).get("value"),
              (j.get("cpu_baseline") or {}).get("cores"), "traffic", j["roofline"].get("traffic"), j["roofline"].get("traffic_stale"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
