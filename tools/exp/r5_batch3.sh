#!/bin/bash
# round 5, GPU batch 3: the C2R load side with wave-uniform table entries, fp32 variant 1 (point-fastest load, line-fastest store), the
# allocator's seconds by phase
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b3
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_variants.py tests/test_gpu_spectral.py tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_fullsize.py -m gpu -q -x -k "not bench and not single_order" --durations=6 > $OUT/pytest_kernels.txt 2>&1; tail -12 $OUT/pytest_kernels.txt
K="$R/tools/kbench --size 1024 --prec f64 --iters 10 --lib-buffers --ranks 2x4 --rank 0"
K5="$R/tools/kbench --size 2048 --prec f32 --iters 5 --lib-buffers --ranks 2x4 --rank 0"
{
echo "== R2C fp64 1024^3 rank 0 of 2x4 (C2R load side: wave-uniform table entries)"
$K --mode r2c
$K --mode r2c --tune-variants
$K --mode r2c --opt pipeline_chunks=1
$K --mode r2c --opt debug_skip=1
echo "== R2C fp32 1024^3 rank 0 of 2x4"
$R/tools/kbench --size 1024 --prec f32 --iters 10 --lib-buffers --ranks 2x4 --rank 0 --mode r2c
echo "== one rank R2C 1024^3 fp64"
$R/tools/kbench --size 1024 --prec f64 --iters 5 --lib-buffers --mode r2c
echo "== spectral_layout=1 fp32 2048^3 rank 0 of 2x4 (inverse x: variant 1 by rule)"
$K5 --mode c2c --opt spectral_layout=1
$K5 --mode c2c --opt spectral_layout=1 --tune-variants
$K5 --mode c2c --opt spectral_layout=1 --opt debug_skip=1
echo "== spectral_layout=1 fp32 1024^3 rank 0 of 2x4"
$R/tools/kbench --size 1024 --prec f32 --iters 10 --lib-buffers --ranks 2x4 --rank 0 --mode c2c --opt spectral_layout=1
$R/tools/kbench --size 1024 --prec f32 --iters 10 --lib-buffers --ranks 2x4 --rank 0 --mode c2c
} > $OUT/kbench_r5b3.txt 2>&1
grep -E "PLAN|FFT|TUNE|==" $OUT/kbench_r5b3.txt | cut -c1-200
python - <<'PY' > $OUT/alloc_times.txt 2>&1
import json, time
import distributedfft_amd as d
for nb in (16 << 30, 16 << 30, 32 << 30, 2 << 30):
    t0 = time.perf_counter()
    b = d.DeviceBuffer.alloc(nb)
    dt = time.perf_counter() - t0
    print(nb >> 30, "GiB", round(dt, 3), "s", json.dumps(d.last_placement_info()))
    b.free()
PY
cat $OUT/alloc_times.txt
