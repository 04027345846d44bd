#!/bin/bash
# tools/final_check.sh [TAG] -- the short end-of-session check in ONE gpurun call (about 9 GPU-minutes):
#   gpurun --timeout 1100 -- 'bash tools/final_check.sh r3b'
# full GPU parity suite, the default bench line, the same line under rocprofv3 --kernel-trace --stats.
TAG=${1:-r3b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_check
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 700 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/${TAG}_pytest_gpu.txt 2>&1; tail -14 $OUT/${TAG}_pytest_gpu.txt
timeout 300 python bench.py > $OUT/bench_${TAG}.json 2> $OUT/bench.err; tail -c 400 $OUT/bench_${TAG}.json; echo; tail -3 $OUT/bench.err
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-multi-rank-path > $OUT/bench_${TAG}_profiled.json 2> $OUT/prof_bench.log )
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench_kernel_stats.csv \;
find $OUT/prof_bench -name "*kernel_trace.csv" -exec cp {} $OUT/${TAG}_bench_kernel_trace.csv \;
python tools/trace_timed_region.py $OUT/${TAG}_bench_kernel_trace.csv 120 > $OUT/${TAG}_bench_kernel_trace_timed_region.txt 2>&1
cat $OUT/${TAG}_bench_kernel_trace_timed_region.txt | tail -8
rm -rf $OUT/prof_bench; du -sh $R/gpurun_out
