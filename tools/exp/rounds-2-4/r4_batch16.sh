#!/bin/bash
# round 4, GPU batch 16 (final library): bench lines of the other grids, the per-GPU plans on library buffers, an N = 8 line of ranks sharing
# this GPU over gloo (what the line of a real 8-GPU run looks like: direct and relayed run, xGMI model)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b16
mkdir -p $OUT
cd $R
export PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python bench.py --size 2048 --precision float --no-cpu-baseline > $OUT/bench_r4c_f32_2048.json 2> $OUT/e1.err
timeout 300 python bench.py --precision float --no-cpu-baseline --no-multi-rank-path > $OUT/bench_r4c_f32_1024.json 2> $OUT/e2.err
{
for cfg in "--size 2048 --prec f32" "--size 1024 --prec f64"; do
  for grid in 2x4 8x1; do
    timeout 300 tools/kbench $cfg --iters 10 --ranks $grid --lib-buffers --tune-variants 2>&1 | grep -E "^PLAN|FFT|total" | cut -c1-150
  done
done
} > $OUT/r4c_per_gpu_kernels_8gpu_plans_library_buffers.txt 2>&1
cat $OUT/r4c_per_gpu_kernels_8gpu_plans_library_buffers.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29683 bench.py --gpus 8 --backend gloo --size 256 --steps 3 --warmup 1 > $OUT/bench_r4c_8ranks_one_gpu_gloo.out 2> $OUT/e3.err
grep "^{" $OUT/bench_r4c_8ranks_one_gpu_gloo.out > $OUT/bench_r4c_8ranks_one_gpu_gloo.json
python - <<'PY'
import json, os
d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4b16")
for f in ("bench_r4c_f32_2048.json", "bench_r4c_f32_1024.json", "bench_r4c_8ranks_one_gpu_gloo.json"):
    try:
        j = json.loads([l for l in open(os.path.join(d, f)) if l.startswith("{")][-1])
        c = j["config"]
        print(f, j["ms_per_step"], j["roofline"]["frac"], c.get("transport"), {k: v["ms"] for k, v in c["per_pass"].items()},
              "relay", (c.get("relay") or {}).get("ms_per_step"), (c.get("relay") or {}).get("error"), "direct" in c)
    except Exception as e:
        print(f, "unreadable", e)
PY
