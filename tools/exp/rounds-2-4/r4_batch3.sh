#!/bin/bash
# round 4, GPU batch 3
# 1. natural-line passes of 2048 fp32 points with 32 points per thread (A/B configurations 1 / 3 load, 2 / 7 store) against the shipped 4 / 5:
#    rank 0 of 2x4 at 2048^3, transforms and copies
# 2. 2048^3 fp32 on ONE GPU (pass order z, x, y): configuration sweeps per pass
# 3. parity after the allocator / tuner changes; 4. bench: no-search line three times, default line once (MPI CPU baseline on the usable cores)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b3
mkdir -p $OUT
cd $R
K=$R/tools/kbench_exp
{
echo "== 2048^3 fp32, rank 0 of 2x4: z pass configuration 4 | 1 | 3, z^-1 5 | 2 | 7 (set order), transforms"
timeout 200 $K --size 2048 --prec f32 --iters 5 --ranks 2x4 --sweep "variant_fz=4,variant_iz=5;variant_fz=1,variant_iz=2;variant_fz=3,variant_iz=7" 2>&1 | grep -E "^PLAN|z-FFT|total"
echo "== as copies"
timeout 200 $K --size 2048 --prec f32 --iters 5 --ranks 2x4 --sweep "debug_skip=1,variant_fz=4,variant_iz=5;debug_skip=1,variant_fz=1,variant_iz=2;debug_skip=1,variant_fz=3,variant_iz=7" 2>&1 | grep -E "^PLAN|z-FFT|total"
} > $OUT/r4_f32_2048_natural_candidates.txt 2>&1
cut -c1-150 $OUT/r4_f32_2048_natural_candidates.txt
{
echo "== 2048^3 fp32 on one GPU (z, x, y order; in = back aliased): as built | z on 1 | z on 3 | x,y on 14 | 15 | 12 | 0 | 9"
timeout 600 $K --size 2048 --prec f32 --iters 3 --sweep "variant_fz=-1;variant_fz=1;variant_fz=3;variant_fx=14,variant_fy=14;variant_fx=15,variant_fy=15;variant_fx=12,variant_fy=12;variant_fx=0,variant_fy=0;variant_fx=9,variant_fy=9" 2>&1 | grep -E "^PLAN|FFT|total"
} > $OUT/r4_f32_2048_single_gpu_candidates.txt 2>&1
grep -E "^PLAN|FFT " $OUT/r4_f32_2048_single_gpu_candidates.txt | cut -c1-150
timeout 1500 python -m pytest tests/test_gpu_placement.py tests/test_gpu_cpp_drivers.py tests/test_gpu_cpp_shim.py tests/test_gpu_cli.py tests/test_gpu_variants.py \
  -m gpu -q -x --durations=5 > $OUT/r4_pytest_b3.txt 2>&1
tail -12 $OUT/r4_pytest_b3.txt
for i in 1 2 3; do
timeout 300 python bench.py --tune-placement 0 --no-cpu-baseline --no-multi-rank-path --no-plain-leg > $OUT/bench_r4_b_notuner_$i.json 2> $OUT/bench_n$i.err; tail -2 $OUT/bench_n$i.err
done
timeout 500 python bench.py > $OUT/bench_r4_b.json 2> $OUT/bench_b.err; tail -3 $OUT/bench_b.err
python - <<'PY'
import json, os, glob
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4b3")
for f in sorted(glob.glob(os.path.join(d, "bench_r4_b*.json"))):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        c = j["config"]
        cb = j.get("cpu_baseline") or {}
        print(os.path.basename(f), "ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"], "plain", c.get("plain_buffers_ms_per_step"),
              "placement s", (c.get("placement") or {}).get("seconds"), "cpu", cb.get("value"), cb.get("cores"), (cb.get("mpi") or {}).get("error", "")[:80],
              "variants", (c.get("variants") or {}).get("trial_fft_ms", [None])[-1])
    except Exception as e:
        print(f, "unreadable:", e)
PY
