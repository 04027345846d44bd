#!/bin/bash
# round 5, GPU batch 2: new tests (x-contiguous spectrum, hardened allocator, RCCL world of one, rccl_smoke), the per-GPU kernels of the
# 2x4 plans with spectral_layout = 1, and a bench line (K = 5 allocator: seconds, rates)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b2
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spectral.py tests/test_gpu_placement.py tests/test_gpu_rccl_one_rank.py tests/test_gpu_multi_device.py -m gpu -q --durations=8 > $OUT/pytest_new.txt 2>&1; tail -25 $OUT/pytest_new.txt
python tools/rccl_smoke.py --gpus 1 > $OUT/rccl_smoke_1gpu.json 2> $OUT/rccl_smoke.err; cat $OUT/rccl_smoke_1gpu.json
K="$R/tools/kbench --size 1024 --prec f64 --iters 10 --lib-buffers --ranks 2x4 --rank 0"
K5="$R/tools/kbench --size 2048 --prec f32 --iters 5 --lib-buffers --ranks 2x4 --rank 0"
{
echo "== spectral_layout=1: fp64 1024^3 rank 0 of 2x4"
$K --mode c2c --opt spectral_layout=1
$K --mode c2c --opt spectral_layout=1 --tune-variants
$K --mode r2c --opt spectral_layout=1
$K --mode r2c --opt spectral_layout=1 --tune-variants
echo "== spectral_layout=1: fp32 2048^3 rank 0 of 2x4"
$K5 --mode c2c --opt spectral_layout=1
$K5 --mode c2c --opt spectral_layout=1 --tune-variants
echo "== spectral_layout=1: slab 8, fp64"
$R/tools/kbench --size 1024 --prec f64 --iters 10 --lib-buffers --ranks 8x1 --rank 0 --mode c2c --opt spectral_layout=1 --tune-variants
$R/tools/kbench --size 1024 --prec f64 --iters 10 --lib-buffers --ranks 8x1 --rank 0 --mode c2c --tune-variants
echo "== one rank, 1024^3 fp64, mirrored inverse: API layout vs spectral_layout=1"
$R/tools/kbench --size 1024 --prec f64 --iters 5 --lib-buffers --mode c2c --opt mirror_inverse=1
$R/tools/kbench --size 1024 --prec f64 --iters 5 --lib-buffers --mode c2c --opt spectral_layout=1
} > $OUT/kbench_spectral.txt 2>&1
grep -E "PLAN|FFT|TUNE|==" $OUT/kbench_spectral.txt | cut -c1-220
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r5a.json 2> $OUT/bench_r5a.err; tail -3 $OUT/bench_r5a.err
python - <<'PY'
import json, os
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r5b2")
try:
    j = json.loads([l for l in open(os.path.join(d, "bench_r5a.json")) if l.startswith("{")][-1]); c = j["config"]
    print("bench", j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("frac_of_measured_copy_peak"), j["roofline"].get("traffic"), c["placement"])
    print({k: v["ms"] for k, v in c["per_pass"].items() if "FFT" in k}, c.get("plain_buffers_ms_per_step"))
    for key in ("plans", "spectral_layout_plans"):
        for p in c["per_gpu_kernels_8gpu"][key]:
            print(key, p["decomposition"], p["kernels_ms_per_step"], {k: v["ms"] for k, v in p["per_pass"].items()})
except Exception as e:
    print("bench unreadable", e)
PY
