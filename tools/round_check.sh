#!/bin/bash
# tools/round_check.sh [TAG] -- the standard measurement set of a round in ONE gpurun call:
#   gpurun --timeout 1800 -- 'bash tools/round_check.sh r3'
# GPU parity suite, the bench line (1024^3 fp64) and its rocprofv3 kernel stats, the 2048^3 fp32 bench line and
# stats, per-pass times of every configuration DESIGN.md quotes (kbench, C ABI, no Python), PMC counters of the
# headline kernels.  Everything lands in gpurun_out/round_check/; copy what DESIGN.md cites into profiles/.
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_check
mkdir -p $OUT
cd $R
K=$R/tools/kbench
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu --durations=10 > $OUT/pytest_gpu.txt 2>&1; tail -14 $OUT/pytest_gpu.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json; echo
# five more bench lines from fresh processes (placement tuner on, its default) and two with plain allocations
for i in 1 2 3 4 5; do python bench.py --no-cpu-baseline --no-multi-rank-path > $OUT/bench_fresh_$i.json 2>> $OUT/bench.err; done
for i in 1 2; do python bench.py --no-cpu-baseline --no-multi-rank-path --tune-placement 0 > $OUT/bench_plain_$i.json 2>> $OUT/bench.err; done
python - <<P > $OUT/${TAG}_placement.txt
import json
print("bench.py from fresh processes, 1024^3 fp64 complex forward+inverse, 10 warm-up + 20 timed steps (ms per step, roofline.frac, per-pass ms)")
for name in ["bench"] + [f"bench_fresh_{i}" for i in range(1, 6)] + ["bench_plain_1", "bench_plain_2"]:
    try:
        d = json.loads(open("$OUT/" + name + ".json").read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "FAILED", e); continue
    pl = d["config"].get("placement")
    print(f"{name:14s} {'tuned placement' if pl else 'plain hipMalloc / torch buffers'}: {d['ms_per_step']:.3f} ms  frac {d['roofline']['frac']:.4f}  ",
          {k: v["ms"] for k, v in d["config"]["per_pass"].items() if "FFT" in k}, "trials", (pl or {}).get("trial_fft_ms_fwd_plus_inv"))
P
cat $OUT/${TAG}_placement.txt
prof() {  # name, command...: rocprofv3 kernel trace + stats of a command, stats csv copied next to the logs
  local name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/prof_$name.log 2>&1 )
  find $OUT/prof_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_${name}_kernel_stats.csv \;
}
# every axis-pass launch of this command is one full pass (the multi-rank leg, whose launches are 1/8 passes, is profiled separately below)
prof bench python $R/bench.py --no-cpu-baseline --no-multi-rank-path
python bench.py --size 2048 --precision float --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_f32_2048.json 2> $OUT/bench_f32_2048.err; tail -c 300 $OUT/bench_f32_2048.json; echo
prof f32_2048 $K --size 2048 --prec f32 --iters 3
prof f32_2048_multirank $K --size 2048 --prec f32 --iters 3 --opt mirror_inverse=1 --opt pipeline_chunks=8
prof f64_2048z $K --size 1024x1024x2048 --prec f64 --iters 3
prof f64_2048y $K --size 1024x2048x1024 --prec f64 --iters 3
prof f64_2048x $K --size 2048x1024x1024 --prec f64 --iters 3
prof f64_multirank $K --size 1024 --prec f64 --iters 5 --opt mirror_inverse=1 --opt pipeline_chunks=8
prof f64_r2c $K --size 1024 --prec f64 --mode r2c --iters 5
prof f32_r2c $K --size 1024 --prec f32 --mode r2c --iters 5
prof f32_c2c $K --size 1024 --prec f32 --iters 5
prof f64_bluestein1000 $K --size 1000 --prec f64 --iters 3 --opt native_mixed=0
prof f64_mixed1000 $K --size 1000 --prec f64 --iters 3
prof f64_mixed1000_r2c $K --size 1000 --prec f64 --mode r2c --iters 3
{
  echo "== c2c fp64 1024";                 $K --size 1024 --prec f64 --iters 5 --check
  echo "== c2c fp64 1024, tuned placement"; $K --size 1024 --prec f64 --iters 5 --check --tune 4
  echo "== c2c fp64 1024 pattern roofs, tuned placement"; $K --size 1024 --prec f64 --iters 5 --opt debug_skip=1 --tune 4
  echo "== c2c fp64 1024 multi-rank path"; $K --size 1024 --prec f64 --iters 5 --check --opt mirror_inverse=1 --opt pipeline_chunks=8
  echo "== c2c fp64 1024 pattern roofs";   $K --size 1024 --prec f64 --iters 5 --opt debug_skip=1
  echo "== c2c fp32 1024";                 $K --size 1024 --prec f32 --iters 5 --check
  echo "== c2c fp32 1024 multi-rank path"; $K --size 1024 --prec f32 --iters 5 --check --opt mirror_inverse=1 --opt pipeline_chunks=8
  echo "== r2c fp64 1024";                 $K --size 1024 --prec f64 --mode r2c --iters 5 --check
  echo "== r2c fp64 1024, tuned placement"; $K --size 1024 --prec f64 --mode r2c --iters 10 --check --tune 4
  echo "== r2c fp32 1024";                 $K --size 1024 --prec f32 --mode r2c --iters 5 --check
  echo "== r2c fp32 1024, tuned placement"; $K --size 1024 --prec f32 --mode r2c --iters 10 --check --tune 4
  echo "== c2c fp32 1024, tuned placement"; $K --size 1024 --prec f32 --iters 10 --check --tune 4
  echo "== c2c fp32 2048";                 $K --size 2048 --prec f32 --iters 3 --check
  echo "== c2c fp32 2048 pattern roofs";   $K --size 2048 --prec f32 --iters 3 --opt debug_skip=1
  echo "== c2c fp32 2048 multi-rank path"; $K --size 2048 --prec f32 --iters 3 --opt mirror_inverse=1 --opt pipeline_chunks=8
  for sz in 1024x1024x2048 1024x2048x1024 2048x1024x1024; do
    echo "== c2c fp64 $sz";                $K --size $sz --prec f64 --iters 3 --check
    echo "== c2c fp64 $sz multi-rank path"; $K --size $sz --prec f64 --iters 3 --opt mirror_inverse=1
  done
  echo "== Bluestein 1000^3 fp64";         $K --size 1000 --prec f64 --iters 3 --check --opt native_mixed=0
  echo "== mixed radix 1000^3 fp64";       $K --size 1000 --prec f64 --iters 5 --check
  echo "== mixed radix 1000^3 fp64 r2c";   $K --size 1000 --prec f64 --mode r2c --iters 5 --check
  echo "== mixed radix 1000^3 fp32";       $K --size 1000 --prec f32 --iters 5 --check
  echo "== mixed radix 1536^3 fp32";       $K --size 1536 --prec f32 --iters 3 --check
  echo "== 4096-point lines fp64";         for sz in 256x256x4096 256x4096x256 4096x256x256; do $K --size $sz --prec f64 --iters 5 --check; done
  echo "== Bluestein 1500-point lines";    $K --size 256x1500x256 --prec f64 --iters 5 --check
  for n in 128 256 512; do echo "== r2c fp64 $n^3 (wall = latency)"; $K --size $n --prec f64 --mode r2c --iters 20; done
  echo "== c2c fp64 256^3 (C2), 512^3";    $K --size 256 --prec f64 --iters 20 --check; $K --size 512 --prec f64 --iters 10 --check
} > $OUT/${TAG}_phase_times.txt 2>&1
grep -E "^==|^PLAN|FFT" $OUT/${TAG}_phase_times.txt | head -220
# PMC: LDS conflicts / activity of the 2048-point fp32 passes and the fp64 R2C passes (separate passes, kernel-trace only)
bash tools/pmc_quick.sh ${TAG}_f32_2048 -- $K --size 2048 --prec f32 --iters 1 > /dev/null 2>&1
bash tools/pmc_quick.sh ${TAG}_f64_r2c -- $K --size 1024 --prec f64 --mode r2c --iters 1 > /dev/null 2>&1
python tools/pmc_summary.py $R/gpurun_out/pmcq_${TAG}_f32_2048 fft_ > $OUT/${TAG}_pmc_f32_2048.txt 2>&1
python tools/pmc_summary.py $R/gpurun_out/pmcq_${TAG}_f64_r2c fft_ > $OUT/${TAG}_pmc_f64_r2c.txt 2>&1
# HBM traffic per launch of the headline kernel (FETCH_SIZE / WRITE_SIZE, separate passes) -> the json bench.py quotes
bash tools/pmc_traffic.sh ${TAG}_f64_1024 -- $K --size 1024 --prec f64 --iters 2 > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_${TAG}_f64_1024 34359738368 "1024^3 fp64 complex, one axis pass per launch (tools/kbench --size 1024 --prec f64)" > $OUT/${TAG}_pmc_traffic.json 2>&1
cat $OUT/${TAG}_pmc_traffic.json
bash tools/pmc_traffic.sh ${TAG}_f64_1000 -- $K --size 1000 --prec f64 --iters 2 > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_${TAG}_f64_1000 32000000000 "1000^3 fp64 complex (mixed radix), one axis pass per launch" > $OUT/${TAG}_pmc_traffic_mixed1000.json 2>&1
cat $OUT/${TAG}_pmc_traffic_mixed1000.json
rm -rf $R/gpurun_out/pmcq_* $R/gpurun_out/pmct_* 2>/dev/null; du -sh $R/gpurun_out 2>/dev/null
