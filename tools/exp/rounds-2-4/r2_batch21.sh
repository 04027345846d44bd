#!/bin/bash
# round 2, GPU batch 21: slab-sequence tests after the Y_Then_ZX length limits moved; bench kernel stats with full-pass launches only
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/b21
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_slab_sequences.py -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-multi-rank-path > $OUT/prof_bench.log 2>&1 )
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/r2_bench_kernel_stats.csv \;
tail -1 $OUT/prof_bench.log | cut -c1-600
head -4 $OUT/r2_bench_kernel_stats.csv | cut -c1-250
