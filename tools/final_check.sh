#!/bin/bash
# tools/final_check.sh [TAG] -- the end-of-session check in ONE gpurun call (about 15 GPU-minutes):
#   gpurun --timeout 2400 -- 'bash tools/final_check.sh r4'
# full GPU parity suite, the default bench line, the same line under rocprofv3 --kernel-trace --stats, and the PMC traffic of the
# library as it is (FETCH_SIZE / WRITE_SIZE in separate passes; the json records the library's sha256, bench.py refuses another's).
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_check
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
K=$R/tools/kbench
sha256sum distributedfft_amd/libdfft_amd.so distributedfft_amd/libdfft_amd_any.so > $OUT/${TAG}_library_sha256.txt
# PMC first: bench.py then finds a profile of THIS library (profiles/ is where it looks: the files are copied there by hand afterwards;
# for this call they are also put where bench.py reads them)
bash tools/pmc_traffic.sh ${TAG}_f64_1024 -- $K --size 1024 --prec f64 --iters 2 > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_${TAG}_f64_1024 34359738368 "1024^3 fp64 complex, one axis pass per launch (tools/kbench --size 1024 --prec f64)" > $OUT/${TAG}_pmc_traffic.json 2>&1
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
bash tools/pmc_traffic.sh ${TAG}_f32_2048 -- $K --size 2048 --prec f32 --iters 1 > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_${TAG}_f32_2048 137438953472 "2048^3 fp32 complex on one GPU, one axis pass per launch" > $OUT/${TAG}_pmc_traffic_f32_2048.json 2>&1
bash tools/pmc_traffic.sh ${TAG}_f64_r2c -- $K --size 1024 --prec f64 --mode r2c --iters 2 > /dev/null 2>&1
python tools/pmc_traffic.py $R/gpurun_out/pmct_${TAG}_f64_r2c 17213423616 "1024^3 fp64 R2C + C2R, spectrum 1024 x 1024 x 513: 2 x 8.61 GB per y / x pass (the real z passes move 8.59 + 8.61 GB)" "dfft::fft_" > $OUT/${TAG}_pmc_traffic_f64_r2c.json 2>&1
rm -rf $R/gpurun_out/pmct_* 2>/dev/null
python - <<PY
import json
for f in ("${TAG}_pmc_traffic.json", "${TAG}_pmc_traffic_f32_2048.json", "${TAG}_pmc_traffic_f64_r2c.json"):
    try:
        j = json.load(open("$OUT/" + f))
        print(f, "traffic / algorithmic =", round(j["hbm_bytes_per_launch"] / j["algorithmic_bytes_per_launch"], 4), "launches", j["dispatches"], "sha", (j.get("library_sha256") or "")[:12])
    except Exception as e:
        print(f, "unreadable", e)
PY
# SUITE=full (default): the whole GPU suite; SUITE=<pytest -k expression>: only the tests that expression selects (a late change of one subsystem)
if [ "${SUITE:-full}" = "full" ]; then
  rm -f $OUT/${TAG}_parity_table.txt
  ( export DFFT_PARITY_TABLE=$OUT/${TAG}_parity_table.txt; timeout 1500 python -m pytest tests -x -q -m gpu --durations=40 ) > $OUT/${TAG}_pytest_gpu.txt 2>&1; tail -22 $OUT/${TAG}_pytest_gpu.txt
  # the part of the thinned parametrisation that the default run leaves out (tests/conftest.py)
  timeout 600 python -m pytest tests -q -m "gpu and slow" --durations=5 > $OUT/${TAG}_pytest_gpu_slow.txt 2>&1; tail -4 $OUT/${TAG}_pytest_gpu_slow.txt
else
  timeout 900 python -m pytest tests -x -q -m gpu -k "$SUITE" --durations=10 > $OUT/${TAG}_pytest_gpu_subset.txt 2>&1; tail -14 $OUT/${TAG}_pytest_gpu_subset.txt
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s); timeout 500 python bench.py > $OUT/bench_${TAG}.json 2> $OUT/bench.err; echo "bench.py (default flags) took $(( $(date +%s) - T0 )) s"; tail -c 400 $OUT/bench_${TAG}.json; echo; tail -3 $OUT/bench.err
timeout 300 python bench.py --size 2048 --precision float --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_${TAG}_f32_2048.json 2> $OUT/bench_f32_2048.err; tail -c 300 $OUT/bench_${TAG}_f32_2048.json; echo
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-multi-rank-path --no-plain-leg > $OUT/bench_${TAG}_profiled.json 2> $OUT/prof_bench.log )
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench_kernel_stats.csv \;
find $OUT/prof_bench -name "*kernel_trace.csv" -exec cp {} $OUT/${TAG}_bench_kernel_trace.csv \;
python tools/trace_timed_region.py $OUT/${TAG}_bench_kernel_trace.csv 120 > $OUT/${TAG}_bench_kernel_trace_timed_region.txt 2>&1
tail -8 $OUT/${TAG}_bench_kernel_trace_timed_region.txt
rm -rf $OUT/prof_bench; du -sh $R/gpurun_out
