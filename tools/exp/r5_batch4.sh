#!/bin/bash
# round 5, GPU batch 4: what the tuner chooses on the 2x4 plans (dfft_get_pass_choices), and where the seconds of the allocator's
# chunk pool go (tools/vmm_cycle patterns; K = 3 against K = 5 in fresh processes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5b4
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for p in a b c d e; do timeout 120 $R/tools/vmm_cycle $p; done > $OUT/vmm_cycle.txt 2>&1
cat $OUT/vmm_cycle.txt
for K in 5 3 2; do
DFFT_PLACEMENT_SPREAD=$K python - <<'PY'
import json, os, time
import distributedfft_amd as d
t00 = time.perf_counter()
out = []
for nb in (32 << 30, 16 << 30, 16 << 30):
    t0 = time.perf_counter()
    b = d.DeviceBuffer.alloc(nb)
    i = d.last_placement_info()
    out.append((nb >> 30, round(time.perf_counter() - t0, 2), i["spread_K"], i["candidates_drawn"], i["probe_TBps"], i["seconds_pool_create"], i["seconds_pool_release"], i["kept"][:12]))
print("SPREAD", os.environ["DFFT_PLACEMENT_SPREAD"], "total", round(time.perf_counter() - t00, 2), out)
PY
done > $OUT/alloc_spread.txt 2>&1
cat $OUT/alloc_spread.txt
K="$R/tools/kbench --size 1024 --prec f64 --iters 10 --lib-buffers --rank 0 --tune-variants"
{
$K --ranks 2x4 --mode c2c
$K --ranks 2x4 --mode r2c
$K --ranks 8x1 --mode c2c
$K --ranks 2x2 --mode c2c
$K --ranks 2x1 --mode c2c
$R/tools/kbench --size 2048 --prec f32 --iters 5 --lib-buffers --ranks 2x4 --rank 0 --mode c2c --tune-variants
$R/tools/kbench --size 2048 --prec f32 --iters 5 --lib-buffers --ranks 8x1 --rank 0 --mode c2c --tune-variants
} > $OUT/kbench_choices.txt 2>&1
grep -E "PLAN|FFT|TUNE|CHOICES" $OUT/kbench_choices.txt | cut -c1-200
